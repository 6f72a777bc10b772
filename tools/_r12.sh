cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r12_t1.log
for fm in 0 16; do
  G6D_SPLIT_FINISH_MAX=$fm timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --lowp "" > gpurun_out/r12_bench_fm$fm.json 2> gpurun_out/r12_bench_fm$fm.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r12_bench_fm$fm.json").read().strip().splitlines()[-1])
    print("fm$fm", round(d["value"], 2), d.get("stages_ms"), d["parity_vs_reference"] if "parity_vs_reference" in d else None)
except Exception as e:
    print("fm$fm failed", e)
PY
done
