cd $GRAFT_REPO_ROOT
for cfg in "--batch 16 --lanes 2" "--batch 16 --lanes 3" "--batch 32 --lanes 2" "--batch 24 --lanes 2" "--batch 16 --lanes 4"; do
  timeout 300 python bench.py $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line); print('$cfg: %.1f images/s, %.2f ms/step' % (r['value'], r['ms_per_step']))"
done | tee gpurun_out/c16_lanes.log
