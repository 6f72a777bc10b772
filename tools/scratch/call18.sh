cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_rccl_world1_gpu.py -x -q 2>&1 | grep -v "^frame\|^E   *frame" | tail -60) > gpurun_out/c18_t.log; grep -n "Error\|assert\|error\|Traceback" gpurun_out/c18_t.log | head -20
