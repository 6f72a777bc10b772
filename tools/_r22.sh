cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wino or conv" 2>&1 | tail -5) > gpurun_out/r22_t1.log; tail -2 gpurun_out/r22_t1.log
cd tools
SIZES=big REPS=10 python wino_split_probe.py 2>&1 | grep -v amdgpu
python wino_split_probe.py 2>&1 | grep -v amdgpu | tail -1
