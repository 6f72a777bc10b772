cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_networks_gpu.py tests/test_parity_timed_gpu.py -m gpu -x -q 2>&1 | tail -12) > gpurun_out/r20_t1.log; tail -3 gpurun_out/r20_t1.log
timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --lowp "" > gpurun_out/r20_bench.json 2> gpurun_out/r20_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r20_bench.json").read().strip().splitlines()[-1])
print(round(d["value"], 2), {k: round(v, 3) for k, v in d.get("stages_ms").items()}, d["parity_vs_reference"]["ref_idx_equal"], d["parity_vs_reference"]["max_rel_diff_row"])
print({k: (round(v["avg_launch_us"], 1), round(v["frac_of_8TBps"], 3)) for k, v in d["hbm_kernels"].items()})
print({k: d[k] for k in d if k.startswith("roofline")})
PY
