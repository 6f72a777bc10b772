cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r14_t1.log; tail -1 gpurun_out/r14_t1.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --lowp "" > gpurun_out/r14_$name.json 2> gpurun_out/r14_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r14_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"], 2), {k: round(v, 3) for k, v in d.get("stages_ms").items()}, d["parity_vs_reference"]["ref_idx_equal"], d["parity_vs_reference"]["max_rel_diff_row"])
except Exception as e:
    print("$name failed", e)
PY
}
run base A=1
run gain92 G6D_WINO_SPLIT_GAIN=0.92
run gain100 G6D_WINO_SPLIT_GAIN=1.0
run gain75 G6D_WINO_SPLIT_GAIN=0.75
run fix4 G6D_WINO_SPLIT_FIX=4
run per03 G6D_WINO_SPLIT_PER=0.3
run per12 G6D_WINO_SPLIT_PER=1.2
run st256 G6D_SPLIT_TARGET=256
run st768 G6D_SPLIT_TARGET=768
run base2 A=1
