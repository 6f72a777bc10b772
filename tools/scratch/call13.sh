cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "resize_bilinear or stats_finalize or affine_act" 2>&1 | tail -4) > gpurun_out/c13_t1.log; tail -2 gpurun_out/c13_t1.log
(timeout 900 python -m pytest tests/test_parity_timed_gpu.py tests/test_networks_gpu.py -x -q 2>&1 | tail -6) > gpurun_out/c13_t2.log; tail -3 gpurun_out/c13_t2.log
timeout 300 python tools/scratch/aten_ops.py 16 2>&1 | grep -v "amdgpu\|Warn\|warn" | head -30 | tee gpurun_out/c13_aten16.txt
STEPS=8 bash tools/knob_bench.sh "" 2>&1 | tee gpurun_out/c13_knob.log
