cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_networks_gpu.py -m gpu -x -q 2>&1 | tail -12) > gpurun_out/r18_t1.log; tail -3 gpurun_out/r18_t1.log
run() { # name, env
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --lowp "" > gpurun_out/r18_$name.json 2> gpurun_out/r18_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r18_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"], 2), {k: round(v, 3) for k, v in d.get("stages_ms").items()}, d["parity_vs_reference"]["ref_idx_equal"], d["parity_vs_reference"]["max_rel_diff_row"])
except Exception as e:
    print("$name failed", e)
PY
}
run multi A=1
run single G6D_TRUNK_MULTI=0
run multi_gain1 G6D_WINO_SPLIT_GAIN=1.0
run multi2 A=1
