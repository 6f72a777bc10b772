# Round 4, first GPU call after wiring F(4x4,3x3) into the detector / refiner: parity tests of the touched networks, layer table, bench
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_parity_timed_gpu.py tests/test_networks_gpu.py tests/test_estimator_gpu.py -q -x 2>&1 | tail -15) > gpurun_out/r04_run1_tests.log 2>&1
tail -4 gpurun_out/r04_run1_tests.log
BATCH=8 timeout 300 python tools/layer_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_layer_table_b8.md
grep -c . gpurun_out/r04_layer_table_b8.md
timeout 600 python bench.py --steps 12 --warmup 4 > gpurun_out/r04_bench1.json 2> gpurun_out/r04_bench1.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04_bench1.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"].get("by_transform"), "parity", d.get("parity_vs_reference"))
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("lowp") or {}).items()})
PY
