cd $GRAFT_REPO_ROOT
(timeout 400 python -m pytest tests/test_wino43_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/c3_w43_tests.log; tail -2 gpurun_out/c3_w43_tests.log
G6D_LIB_PATH=$PWD/gen6d_amd/csrc/_abl/libgen6d_t.so timeout 200 python tools/w43_timing.py 2>&1 | grep -v amdgpu > gpurun_out/c3_w43_timing.md; cat gpurun_out/c3_w43_timing.md
STEPS=8 bash tools/knob_bench.sh "" "w43_map=1" 2>&1 | tee gpurun_out/c3_knob.log
