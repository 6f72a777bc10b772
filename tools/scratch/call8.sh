cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "winograd or conv_igemm" 2>&1 | tail -6) > gpurun_out/c8_tests.log; tail -4 gpurun_out/c8_tests.log
(timeout 900 python -m pytest tests/test_networks_gpu.py -x -q -k "selector or refiner" 2>&1 | tail -6) > gpurun_out/c8_tests2.log; tail -4 gpurun_out/c8_tests2.log
BATCH=8 python tools/layer_table.py 2>&1 | grep "wino3x3 conv\|whole stage" > gpurun_out/c8_layers.txt; cat gpurun_out/c8_layers.txt
STEPS=8 bash tools/knob_bench.sh "" "" 2>&1 | tee gpurun_out/c8_knob.log
