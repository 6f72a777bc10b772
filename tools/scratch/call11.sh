cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_lowp_gpu.py -x -q -k "selector_headline or detector_headline" 2>&1 | tail -6) > gpurun_out/c11_tests.log; tail -4 gpurun_out/c11_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cached --no-chained --no-sweep > gpurun_out/c11_bench.json 2> gpurun_out/c11_bench.err; echo rc $?
python - <<'PY'
import json
for line in open('gpurun_out/c11_bench.json'):
    if line.startswith('{'):
        d=json.loads(line); print(d['value'])
        for k,v in d['lowp'].items(): print(k, round(v['value'],1), v.get('selector_logits'), v['parity_vs_reference'], round(v['roofline']['frac'],4))
PY
