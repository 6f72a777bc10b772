# Throughput of the hipGraph launch mode over (queries per batch) x (batches in flight): python bench.py --batch B --lanes L.
# Usage (GPU box, repo root): bash tools/batch_sweep.sh "1x4 2x2 4x1 4x2 8x1 8x2" > gpurun_out/batch_sweep.txt
for c in ${1:-"1x4 2x2 4x1 4x2 8x1 8x2"}; do
  B=${c%x*}; L=${c#*x}
  timeout 300 python bench.py --batch $B --lanes $L --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" 2> /tmp/sweep_err.log | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        r = json.loads(line)
        print('batch $B lanes $L: %.1f images/s, %.2f ms/step, single %.2f ms (%s), wino frac %.3f (%.2f ms/query), conv frac %.3f (%.2f ms/query), launches/query %.0f, parity %s' % (
            r['value'], r['ms_per_step'], r.get('single_query_ms') or 0, {k: round(v, 2) for k, v in (r.get('single_query_ms_detail') or {}).items()}, r['roofline']['frac'], r['roofline']['ms_per_query'], r['roofline_conv']['frac'], r['roofline_conv']['ms_per_query'],
            r['roofline']['launches_per_query'] + r['roofline_conv']['launches_per_query'], r.get('parity_vs_reference')))
" || tail -5 /tmp/sweep_err.log
done
