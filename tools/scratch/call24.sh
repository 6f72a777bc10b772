cd $GRAFT_REPO_ROOT
for cfg in "--lanes 3" "--lanes 3 --fork" "--lanes 2 --fork" "--lanes 3 --batch 12" "--lanes 4 --batch 8"; do
  timeout 300 python bench.py $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line); print('$cfg: %.1f images/s, %.2f ms/step, parity %.1e' % (r['value'], r['ms_per_step'], r['parity_vs_reference']['max_rel_diff_row']))"
done | tee gpurun_out/c24_fork.log
