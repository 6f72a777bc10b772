cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_networks_gpu.py -m gpu -x -q -k "selector or Selector" 2>&1 | tail -12) > gpurun_out/r19_t1.log; tail -3 gpurun_out/r19_t1.log
timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --lowp "" > gpurun_out/r19_bench.json 2> gpurun_out/r19_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r19_bench.json").read().strip().splitlines()[-1])
print(round(d["value"], 2), {k: round(v, 3) for k, v in d.get("stages_ms").items()}, d["parity_vs_reference"]["ref_idx_equal"], d["parity_vs_reference"]["max_rel_diff_row"])
print(json.dumps(d["hbm_kernels"], indent=0))
PY
