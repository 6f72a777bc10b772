cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_timed_gpu.py tests/test_adversarial_gpu.py -x -q -k "detector or score_mlp" 2>&1 | tail -3) > gpurun_out/c30_t.log; tail -2 gpurun_out/c30_t.log
STEPS=10 bash tools/knob_bench.sh "" "" 2>&1 | tee gpurun_out/c30_knob.log
BATCH=8 python tools/layer_table.py 2>&1 | grep "## detector" 
