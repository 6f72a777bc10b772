cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_igemm" 2>&1 | tail -8) > gpurun_out/c6_tests.log; tail -5 gpurun_out/c6_tests.log
STEPS=8 bash tools/knob_bench.sh "conv_pm=0" "" "conv_pm=0" "" 2>&1 | tee gpurun_out/c6_knob.log
