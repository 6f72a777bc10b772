cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_wino43_gpu.py -m gpu -q 2>&1 | tail -3
G6D_LIB_PATH=$PWD/gen6d_amd/csrc/_abl/libgen6d_t.so python tools/w43_timing.py 2>&1 | grep "^| p\|^| c"
python tools/w43_probe.py 2>&1 | grep -v amdgpu
bash tools/knob_bench.sh "" ""
