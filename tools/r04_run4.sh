# Round 4, fourth GPU call: which parts of the selector must stay on fp32 operands in the reduced-precision mode; 16 x 3 lanes; the bench
# line with the new defaults (batches of 16, refiner F43 only from 4 queries per launch on)
cd $GRAFT_REPO_ROOT
timeout 300 python tools/lowp_selector_schemes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_lowp_selector_schemes.md; cat gpurun_out/r04_lowp_selector_schemes.md
STEPS=8 bash tools/batch_sweep.sh "16x3" > gpurun_out/batch_sweep_16x3.txt 2>&1; cut -c1-200 gpurun_out/batch_sweep_16x3.txt
timeout 900 python bench.py --no-sweep > gpurun_out/r04_bench4.json 2> gpurun_out/r04_bench4.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("gpurun_out/r04_bench4.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "batch", d["batch"], "frac", round(d["roofline"]["frac"], 3), "single", d["single_query_ms"], d["stages_ms"])
print("chained", d["chained"]["value"], d["chained"]["vs_host_driven_predict"])
print("lowp", {k: (round(v["value"], 1), v["selector_logits"]["err_over_margin"]) for k, v in d["lowp"].items()})
print("hbm", {k: round(v["avg_launch_us"], 1) for k, v in d["hbm_kernels"].items()}, "parity", d["parity_vs_reference"])
PY
