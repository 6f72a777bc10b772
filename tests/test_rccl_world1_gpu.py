"""The reference-sharded selector / detector path on REAL RCCL, on the one GPU a lease has.

A one-rank `nccl` process group accepts every collective, so with `set_shard(0, 1, force_collectives=True)` the sharded code path
(reference selector.py:27-90,201,207-209 forces the exchanges; detector.py:246-247 the MAX over references) issues its 9 + 1
collectives per batch through ncclAllReduce / ncclAllGather on device tensors on the launch stream.  The worker
  * counts them (10 per batch of queries, 1 at build time),
  * compares the eager rows with the plain (collective-free) path on the same pipeline,
  * captures the batch — kernels AND collectives — into ONE hipGraph per lane, three lanes on three communicators (round 6), replays
    them concurrently on other images and compares again,
  * holds the rows to the reference's golden rows (tests/golden/pipeline_rows.npz).
It runs in a subprocess under a timeout: a communicator that wedges must not take the test session (or the box) with it.
"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch
    sys.path.insert(0, %r)
    from gen6d_amd import ops, parallel, synth
    from gen6d_amd.pipeline import TensorPipeline
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rank, world, _ = parallel.init_from_env(backend="nccl", force=True)
    assert (rank, world) == (0, 1) and parallel.backend_name() == "nccl"
    ops.SERIAL = True
    parallel.COLLECTIVE_LOG = []
    pipe = TensorPipeline(dev, shard=(0, 1), force_collectives=True)
    pipe.build()
    n_build = len(parallel.COLLECTIVE_LOG)
    fulls = synth.imgs_to_tensor(synth.synth_images(4, 480, 640, seed=100)).to(dev)
    crops = synth.imgs_to_tensor(synth.synth_images(4, 128, 128, seed=200)).to(dev)
    B = 2
    rows_sh = pipe.query(fulls[0:B], crops[0:B])
    torch.cuda.synchronize()
    log, parallel.COLLECTIVE_LOG = parallel.COLLECTIVE_LOG, None
    n_query = len(log) - n_build
    kinds = sorted(k for k, *_ in log[n_build:])
    assert n_build == 1 and n_query == 10, (n_build, n_query, kinds)       # R1/R2 at build; 9 selector + 1 detector per batch
    assert kinds.count("all_reduce_max") == 1 and kinds.count("all_reduce_sum") == 6 and kinds.count("all_gather_rows") == 3, kinds
    # the same pipeline without collectives
    for net in (pipe.selector, pipe.detector):
        net.set_shard(0, 1, force_collectives=False)
    rows_un = torch.cat([pipe.query(fulls[i:i + B], crops[i:i + B]) for i in (0, 2)], 0)
    for net in (pipe.selector, pipe.detector):
        net.set_shard(0, 1, force_collectives=True)
    torch.cuda.synchronize()
    rel = lambda a, b: float(((a - b).abs() / b.abs().clamp(min=1.0)).max())
    e_eager = rel(rows_sh, rows_un[0:B])
    assert e_eager <= 1e-4 and bool((rows_sh[:, 3] == rows_un[0:B, 3]).all()), e_eager
    # kernels + collectives of a batch in ONE hipGraph per lane, every lane on its OWN communicator (parallel.lane_groups; round 6:
    # three batches in flight in the sharded mode too); the three replays are enqueued back to back and overlap
    LANES = 3
    pipe.capture(lanes=LANES, batch=B)
    assert len(parallel._LANE_GROUPS) == LANES - 1 and pipe.selector.group is None
    pend = [pipe.query_graph(fulls[i:i + B], crops[i:i + B], lane) for lane, i in enumerate((0, 2, 0))]
    got = []
    for out, stream in pend:
        stream.synchronize()
        got.append(out.clone())
    e_graph = max(rel(got[0], rows_un[0:B]), rel(got[1], rows_un[2:4]), rel(got[2], rows_un[0:B]))
    assert e_graph <= 1e-4, e_graph
    assert float((got[0] - got[1]).abs().max()) > 1e-3                      # the replay really processed the other images
    gold = torch.from_numpy(np.load(os.path.join(%r, "tests", "golden", "pipeline_rows.npz"))["rows"]).float().to(dev)
    e_gold = max(rel(got[0], gold[0:B]), rel(got[1], gold[2:4]))
    assert e_gold <= 1e-4 and bool((got[1][:, 3] == gold[2:4, 3]).all()), e_gold
    torch.distributed.destroy_process_group()
    print("rccl world-1 ok: collectives per batch", n_query, "eager vs plain %%.2e graph vs plain %%.2e graph vs golden %%.2e" %% (e_eager, e_graph, e_gold))
""")


def test_sharded_path_on_rccl_world1_eager_and_captured(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=420)
    sys.stdout.write(out.stdout[-2000:])
    assert out.returncode == 0, out.stderr[-3000:]
    assert "rccl world-1 ok" in out.stdout
    try:
        from parity_log import record
        tail = [l for l in out.stdout.splitlines() if "rccl world-1 ok" in l][-1]      # (RCCL prints its version banner at exit)
        record("test_sharded_path_on_rccl_world1_eager_and_captured", "captured graph rows vs golden rows (rel)", float(tail.split()[-1]), 1e-4)
    except Exception:
        pass


def test_replica_bench_plumbing_on_rccl_world1():
    """bench.py's query-replica mode with the process group forced at world size 1: barrier, MAX-reduce of the time, all-gather of the
    result rows and the hipGraph capture (three lanes) all run next to RCCL and the process group's watchdog thread, as they will on N
    GPUs; the gathered rows keep the reference's golden rows."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1", "--batch", "4",
                          "--no-cpu-baseline", "--no-cached", "--no-chained", "--no-sweep", "--lowp", ""], env=env, capture_output=True, text=True,
                         timeout=420)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["backend"] == "nccl" and d["ranks_seen"] == 1 and d["n_gpus"] == 1
    pv = d["parity_vs_reference"]
    assert pv["ref_idx_equal"] and pv["max_rel_diff_row"] <= 1e-4 and pv["rows_checked"] == 12, pv
