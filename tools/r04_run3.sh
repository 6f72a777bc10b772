# Round 4, third GPU call: the tests that failed / changed since run 2, the driver's bench line, sharded + replica runs on 2 gloo ranks,
# launch-batch sweep beyond 8, then the rocprofv3 round profile (tools/profile_round.sh)
cd $GRAFT_REPO_ROOT
export G6D_PARITY_LOG=$PWD/gpurun_out/parity_r04.jsonl
(timeout 900 python -m pytest tests/test_adversarial_gpu.py tests/test_kernels_gpu.py tests/test_networks_gpu.py -m gpu -q -k "adversarial or linear_gemv or refiner or alternative or test_detector" 2>&1 | tail -30) > gpurun_out/r04_tests_run3.log 2>&1
tail -5 gpurun_out/r04_tests_run3.log
unset G6D_PARITY_LOG
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc $?"
timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --no-chained --no-sweep --no-cached --lowp "" > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err
timeout 400 python bench.py --gpus 2 --shard-refs --steps 8 --warmup 2 --no-cpu-baseline --no-chained --no-sweep --no-cached --lowp "" > gpurun_out/bench_gpus2_shard.json 2> gpurun_out/bench_gpus2_shard.err
python - <<PY
import json
for f in ("bench_final", "bench_gpus2", "bench_gpus2_shard"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 2), "batch", d.get("batch"), d.get("ranks_seen"), d.get("backend"), "coll", (d.get("collectives_per_query") or {}).get("total"), (d.get("collectives_per_query") or {}).get("per_batch"),
              "hbm", {k: round(v["avg_launch_us"], 1) for k, v in d["hbm_kernels"].items()}, "single", d.get("single_query_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
STEPS=8 bash tools/batch_sweep.sh "16x1 16x2 32x1" > gpurun_out/batch_sweep_all.txt 2>&1; cat gpurun_out/batch_sweep_all.txt | cut -c1-220
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -8 gpurun_out/profile_round.log | cut -c1-250
