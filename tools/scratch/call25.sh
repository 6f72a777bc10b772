cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or selector_tail" 2>&1 | tail -4) > gpurun_out/c25_t1.log; tail -2 gpurun_out/c25_t1.log
(timeout 900 python -m pytest tests/test_networks_gpu.py tests/test_parity_timed_gpu.py tests/test_adversarial_gpu.py -x -q -k "detector or selector" 2>&1 | tail -4) > gpurun_out/c25_t2.log; tail -2 gpurun_out/c25_t2.log
STEPS=10 bash tools/knob_bench.sh "" "" 2>&1 | tee gpurun_out/c25_knob.log
