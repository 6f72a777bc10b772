cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear_gemv or resize_bilinear" 2>&1 | tail -4) > gpurun_out/c14_t1.log; tail -2 gpurun_out/c14_t1.log
(timeout 600 python -m pytest tests/test_networks_gpu.py -x -q -k "refiner" 2>&1 | tail -4) > gpurun_out/c14_t2.log; tail -2 gpurun_out/c14_t2.log
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-cached --no-chained --no-sweep --lowp "" > gpurun_out/c14_bench.json 2>/dev/null
python - <<'PY'
import json
for line in open('gpurun_out/c14_bench.json'):
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['single_query_ms']); print(json.dumps(d['hbm_kernels']))
PY
