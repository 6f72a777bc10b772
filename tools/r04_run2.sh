# Round 4, second GPU call: the full GPU suite on the current tree, smoke, and bench.py exactly as the driver runs it (no flags)
cd $GRAFT_REPO_ROOT
export G6D_PARITY_LOG=$PWD/gpurun_out/parity_r04.jsonl; rm -f $G6D_PARITY_LOG
(timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | grep -v "^  \|Warning\|warnings" | tail -60) > gpurun_out/r04_tests_full.log 2>&1
tail -4 gpurun_out/r04_tests_full.log
unset G6D_PARITY_LOG
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SECONDS=0
timeout 900 python bench.py > gpurun_out/r04_bench2.json 2> gpurun_out/r04_bench2.err; echo "bench rc $? in ${SECONDS}s"; tail -3 gpurun_out/r04_bench2.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_bench2.json").read().strip().splitlines()[-1])
    print("value", round(d["value"], 1), "frac", round(d["roofline"]["frac"], 3), "parity", d.get("parity_vs_reference"))
    print("lowp", {k: (round(v["value"], 1), v.get("roofline", {}).get("frac"), v.get("selector_logits"), (v.get("parity_vs_reference") or {}).get("ref_idx_equal")) for k, v in (d.get("lowp") or {}).items()})
    print("chained", d.get("chained"))
    print("sweep", {k: (round(v["value"], 1), v.get("parity_vs_reference")) for k, v in (d.get("sweep") or {}).items()})
    print("hbm", {k: (round(v["avg_launch_us"], 1), round(v["frac_of_8TBps"], 3)) for k, v in d["hbm_kernels"].items()})
    print("single", d.get("single_query_ms"), "cached", (d.get("cached") or {}).get("value"), (d.get("cached") or {}).get("refiner_step_ms_single_query"))
except Exception as e:
    print("bench parse failed", e)
PY
