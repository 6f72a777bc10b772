cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_lowp_gpu.py -x -q -k "winograd_kernel or wino16_family" 2>&1 | tail -3) > gpurun_out/c33_t.log; tail -2 gpurun_out/c33_t.log
for k in "wino_perm=0" "" "wino_perm=0" ""; do echo "== [$k]"; KNOBS="$k" BATCH=8 python tools/layer_table.py 2>&1 | grep "mul aff stats\|## selector"; done | tee gpurun_out/c33_perm.log
