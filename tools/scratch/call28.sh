cd $GRAFT_REPO_ROOT
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc $? in ${SECONDS}s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print(round(d["value"], 2), d["ms_per_step"], d["single_query_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline_conv"]["frac"], d["roofline_conv"].get("traffic_over_algorithmic"), {k: round(v["value"], 1) for k, v in d["lowp"].items()}, d["chained"]["value"], d["parity_vs_reference"], d.get("side_leg_errors"))
PY
