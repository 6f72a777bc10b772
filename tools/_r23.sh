cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wino or conv" 2>&1 | tail -5) > gpurun_out/r23_t1.log; tail -2 gpurun_out/r23_t1.log
cd tools
SIZES=big REPS=10 python wino_split_probe.py 2>&1 | grep -v amdgpu
python wino_split_probe.py 2>&1 | grep -v amdgpu | tail -1
cd ..
timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --lowp "" > gpurun_out/r23_bench.json 2> gpurun_out/r23_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r23_bench.json").read().strip().splitlines()[-1])
print(round(d["value"], 2), {k: round(v, 3) for k, v in d.get("stages_ms").items()}, d["parity_vs_reference"]["ref_idx_equal"], d["parity_vs_reference"]["max_rel_diff_row"])
print({k: (round(d[k]["achieved"],1), round(d[k]["ms_per_step"],2)) for k in d if k.startswith("roofline")})
PY
