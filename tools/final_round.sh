# Round-end measurements on the GPU box (gpurun): full GPU test suite, the bench variants kept under profiles/, then the
# rocprofv3 round profile (tools/profile_round.sh).  Raw outputs -> gpurun_out/; tools/assemble_profiles.py rNN commits them.
cd $GRAFT_REPO_ROOT
export G6D_PARITY_LOG=$PWD/gpurun_out/parity_r03.jsonl; rm -f $G6D_PARITY_LOG
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/final_tests.log; tail -3 gpurun_out/final_tests.log
unset G6D_PARITY_LOG
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 400 gpurun_out/bench_final.json | head -c 300; echo
timeout 300 python bench.py --gpus 2 --steps 12 --warmup 3 --no-cpu-baseline --lowp "" > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err
timeout 300 python bench.py --gpus 2 --shard-refs --steps 12 --warmup 3 --no-cpu-baseline --lowp "" > gpurun_out/bench_gpus2_shard.json 2> gpurun_out/bench_gpus2_shard.err
timeout 600 python bench.py --chained --steps 16 --warmup 4 --no-cpu-baseline --lowp "" > gpurun_out/bench_chained.json 2> gpurun_out/bench_chained.err
python - <<PY
import json
for f in ("bench_final", "bench_gpus2", "bench_gpus2_shard", "bench_chained"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 2), d.get("ranks_seen"), d.get("backend"), (d.get("chained") or {}).get("value"), {k: round(v["value"], 1) for k, v in (d.get("lowp") or {}).items() if isinstance(v, dict) and "value" in v})
    except Exception as e:
        print(f, "failed", e)
PY
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -5 gpurun_out/profile_round.log
