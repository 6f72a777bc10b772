# Round-end measurements on the GPU box (gpurun): full GPU test suite, smoke, the bench line as the driver runs it, 2-rank gloo runs
rm -rf gen6d_amd/csrc/_abl   # ablation builds of the round (28 MB of profiling binaries once rode on every driver push: VERDICT r05 weak #11)
# (replicas / sharded references), then the rocprofv3 round profile (tools/profile_round.sh).  Raw outputs -> gpurun_out/;
# tools/assemble_profiles.py rNN + tools/parity_table.py + tools/sweep_table.py commit them under profiles/.
cd $GRAFT_REPO_ROOT
export G6D_PARITY_LOG=$PWD/gpurun_out/parity_r06.jsonl; rm -f $G6D_PARITY_LOG
(timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=25 2>&1 | tail -70) > gpurun_out/final_tests.log; tail -3 gpurun_out/final_tests.log
unset G6D_PARITY_LOG
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc $? in ${SECONDS}s"
timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --no-chained --no-sweep --no-cached --lowp "" > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err
timeout 400 python bench.py --gpus 2 --shard-refs --steps 8 --warmup 2 --no-cpu-baseline --no-chained --no-sweep --no-cached --lowp "" > gpurun_out/bench_gpus2_shard.json 2> gpurun_out/bench_gpus2_shard.err
# RCCL on the one GPU of the lease: the reference-sharded path (9 + 1 collectives per batch, captured into the batch's hipGraph) and the
# query-replica plumbing on a one-rank nccl group
timeout 400 python bench.py --gpus 1 --shard-refs --steps 8 --warmup 2 | grep '^{"metric' > gpurun_out/bench_shard_rccl_world1.json 2> gpurun_out/bench_shard_rccl_world1.err
timeout 400 python bench.py --gpus 1 --force-dist --steps 8 --warmup 2 --no-cpu-baseline --no-chained --no-sweep --no-cached --lowp "" | grep '^{"metric' > gpurun_out/bench_force_dist.json 2> gpurun_out/bench_force_dist.err
python - <<PY
import json
for f in ("bench_final", "bench_gpus2", "bench_gpus2_shard", "bench_shard_rccl_world1", "bench_force_dist"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 2), "batch", d.get("batch"), d.get("ranks_seen"), d.get("backend"), "coll/query", (d.get("collectives_per_query") or {}).get("total"),
              "single", d.get("single_query_ms"), {k: round(v["value"], 1) for k, v in (d.get("lowp") or {}).items() if isinstance(v, dict)}, (d.get("chained") or {}).get("value"), d.get("parity_vs_reference"))
    except Exception as e:
        print(f, "failed", e)
PY
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -4 gpurun_out/profile_round.log | cut -c1-200
bash tools/single_query_profile.sh | tail -2
