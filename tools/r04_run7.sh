cd $GRAFT_REPO_ROOT
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc $? in ${SECONDS}s"
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 2), "batch", d["batch"], "frac", round(d["roofline"]["frac"], 3), "single", d["single_query_ms"], "errors", d.get("side_leg_errors"))
print("lowp", {k: (round(v["value"], 1), round(v["roofline"]["frac"], 3), round(v["selector_logits"]["err_over_margin"], 3), v["selector_logits"]["argmax_equal"], v["parity_vs_reference"]["ref_idx_equal"]) for k, v in d["lowp"].items()})
print("chained", d["chained"]["value"], d["chained"]["vs_host_driven_predict"]["each_step_on_the_host_paths_input_pose_maxabs"])
print("sweep", {k: (round(v["value"], 1), (v.get("parity_vs_reference") or {}).get("logits_max_abs_diff")) for k, v in d["sweep"].items()})
print("cached", d["cached"]["value"], d["cached"]["refiner_step_ms_single_query"], "parity", d["parity_vs_reference"], "cpu", d["cpu_baseline"]["value"])
PY
