cd $GRAFT_REPO_ROOT
run() { # name, args
  name=$1; shift
  timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --lowp "" "$@" > gpurun_out/r17_$name.json 2> gpurun_out/r17_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r17_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"], 2), {k: round(v, 3) for k, v in d.get("stages_ms").items()}, d["parity_vs_reference"]["ref_idx_equal"], d["parity_vs_reference"]["max_rel_diff_row"])
except Exception as e:
    print("$name failed", e)
PY
}
run lanes2 --lanes 2
run lanes3 --lanes 3
run lanes4 --lanes 4
run lanes6 --lanes 6
run lanes3fork --lanes 3 --fork
