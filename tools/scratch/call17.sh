cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_rccl_world1_gpu.py -x -q 2>&1 | tail -12) > gpurun_out/c17_t.log; tail -6 gpurun_out/c17_t.log
