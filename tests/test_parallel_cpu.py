"""N>1 path on CPU: world_size-2 gloo processes exercise the query sharding, the MAX-over-ranks timing reduction and
the all-gather of per-query result rows used by bench.py (query replicas, no data-path collective)."""
import os
import subprocess
import sys
import textwrap

import pytest

from gen6d_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(10, 2), (7, 4), (3, 8), (64, 8), (0, 2)])
def test_shard_range_partitions(n, world):
    spans = [parallel.shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
        assert e0 == b1
    sizes = [e - b for b, e in spans]
    assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import sys, torch
    sys.path.insert(0, %r)
    from gen6d_amd import parallel
    rank, world, local = parallel.init_from_env(backend="gloo")
    assert world == 2
    n_items = 5
    b, e = parallel.shard_range(n_items, rank, world)
    rows = torch.stack([torch.full((3,), float(i)) for i in range(b, e)]) if e > b else torch.zeros((0, 3))
    parallel.barrier()
    t = parallel.max_over_ranks(1.0 + rank)
    assert t == 2.0, t
    allrows = parallel.gather_rows(rows, n_items)
    assert allrows.shape == (5, 3) and torch.equal(allrows[:, 0], torch.arange(5.0)), allrows
    print("rank", rank, "ok")
""") % ROOT


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2
