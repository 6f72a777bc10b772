cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_wino43_gpu.py -q -x 2>&1 | tail -30) > gpurun_out/w43_tests.log 2>&1
tail -5 gpurun_out/w43_tests.log
(timeout 600 python tools/w43_bench.py 5 2>&1 | grep -v amdgpu.ids) > gpurun_out/w43_bench.md 2>&1
cat gpurun_out/w43_bench.md
ABL="1 2 4 7" bash tools/w43_ablate_run.sh > gpurun_out/w43_ablate.log 2>&1; cat gpurun_out/w43_ablate.log
