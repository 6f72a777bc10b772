cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line); print('%.1f images/s' % r['value'], r['parity_vs_reference'])"
(timeout 900 python -m pytest tests/test_rccl_world1_gpu.py -x -q 2>&1 | grep -v "^frame\|^E   *frame" | tail -30) > gpurun_out/c19_t.log; tail -5 gpurun_out/c19_t.log
