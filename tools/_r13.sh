cd $GRAFT_REPO_ROOT
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --lowp "" > gpurun_out/r13_$name.json 2> gpurun_out/r13_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r13_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"], 2), {k: round(v, 3) for k, v in d.get("stages_ms").items()}, d["parity_vs_reference"]["ref_idx_equal"], d["parity_vs_reference"]["max_rel_diff_row"])
except Exception as e:
    print("$name failed", e)
PY
}
run fm8 G6D_SPLIT_FINISH_MAX=8
run fm32 G6D_SPLIT_FINISH_MAX=32
run fm64 G6D_SPLIT_FINISH_MAX=64
run fm16 G6D_SPLIT_FINISH_MAX=16

