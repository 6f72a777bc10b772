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


SHARD_WORKER = textwrap.dedent("""
    import sys, numpy as np, torch
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import ref_ops
    from gen6d_amd import ops, parallel, synth
    from gen6d_amd.network import name2network
    for name in dir(ops):                       # CPU emulation of the HIP ops (host-logic test)
        if not name.startswith("_") and callable(getattr(ops, name)) and hasattr(ref_ops, name):
            setattr(ops, name, getattr(ref_ops, name))
    rank, world, local = parallel.init_from_env(backend="gloo")
    g = dict(np.load(%r))
    rfn, an = int(g["rfn"]), int(g["an"])
    net = name2network["selector"]({"name": "t", "selector_angle_num": an}).eval()
    net.load_state_dict(synth.synth_state_dict("selector", an=an))
    net.set_shard(rank, world)
    case = synth.selector_case(rfn, an)
    with torch.no_grad():
        out = net({"ref_imgs": case["ref_imgs"], "ref_imgs_info": {"poses": case["ref_poses"]},
                   "object_center": case["object_center"], "object_vert": case["object_vert"],
                   "que_imgs_info": {"imgs": case["que_imgs"]}, "eval": True})
    b, e = parallel.shard_range(rfn, rank, world)
    assert net.ref_feats_cache[0].shape[0] == (e - b) * an          # only the local slice of the cache is resident
    np.testing.assert_allclose(out["ref_vp_logits"].numpy(), g["logits"], atol=2e-3)
    np.testing.assert_allclose(out["angles_pr"].numpy(), g["angles"], atol=2e-3)
    assert np.array_equal(out["ref_vp_logits"].argmax(1).numpy(), g["logits"].argmax(1))
    print("rank", rank, "sharded selector ok")
""")


@pytest.mark.parametrize("world,port", [(2, 29631), (3, 29633)])
def test_reference_sharded_selector_matches_golden(tmp_path, world, port):
    """8 references x 5 rotations sharded over 2 / 3 (ragged) gloo ranks reproduce the reference's logits."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", "sel_small.npz")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("sharded selector ok") == world


LANES_WORKER = textwrap.dedent("""
    import sys, threading, time, numpy as np, torch
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import ref_ops
    from gen6d_amd import ops, parallel, synth
    from gen6d_amd.network import name2network
    for name in dir(ops):
        if not name.startswith("_") and callable(getattr(ops, name)) and hasattr(ref_ops, name):
            setattr(ops, name, getattr(ref_ops, name))
    rank, world, local = parallel.init_from_env(backend="gloo")
    g = dict(np.load(%r))
    rfn, an = int(g["rfn"]), int(g["an"])
    groups = parallel.lane_groups(2)                     # lane 0: the default group, lane 1: its own communicator (created collectively)
    assert groups[0] is None and groups[1] is not None
    case = synth.selector_case(rfn, an)
    nets, outs = [], [None, None]
    for lane in range(2):
        net = name2network["selector"]({"name": "t", "selector_angle_num": an}).eval()
        net.load_state_dict(synth.synth_state_dict("selector", an=an))
        net.set_shard(rank, world, group=groups[lane])
        nets.append(net)

    def run(lane):
        with torch.no_grad():
            outs[lane] = nets[lane]({"ref_imgs": case["ref_imgs"], "ref_imgs_info": {"poses": case["ref_poses"]},
                                     "object_center": case["object_center"], "object_vert": case["object_vert"],
                                     "que_imgs_info": {"imgs": case["que_imgs"]}, "eval": True})
    # the two lanes run CONCURRENTLY, started in opposite orders on the two ranks: their collectives interleave differently on every
    # rank — legal only because each lane's collectives live in their own communicator (inside one they would pair up wrongly or hang)
    order = (0, 1) if rank %% 2 == 0 else (1, 0)
    threads = [threading.Thread(target=run, args=(lane,)) for lane in order]
    threads[0].start(); time.sleep(0.3); threads[1].start()
    for t in threads:
        t.join(timeout=500)
        assert not t.is_alive(), "a lane hung"
    for lane in range(2):
        np.testing.assert_allclose(outs[lane]["ref_vp_logits"].numpy(), g["logits"], atol=2e-3)
        assert np.array_equal(outs[lane]["ref_vp_logits"].argmax(1).numpy(), g["logits"].argmax(1))
    assert torch.equal(outs[0]["ref_vp_logits"], outs[1]["ref_vp_logits"])
    print("rank", rank, "two lanes ok")
""")


def test_two_lanes_on_two_communicators(tmp_path):
    """Round 6: the reference-sharded mode keeps several batches in flight because every hipGraph lane enqueues its
    collectives on its OWN process group (parallel.lane_groups).  Two gloo ranks, two lanes in threads started in opposite orders on the two
    ranks: both lanes reproduce the reference's logits, bit-identically to each other."""
    script = tmp_path / "lanes_worker.py"
    script.write_text(LANES_WORKER % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", "sel_small.npz")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("two lanes ok") == 2


DET_WORKER = textwrap.dedent("""
    import sys, numpy as np, torch
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import ref_ops
    from gen6d_amd import ops, parallel, synth
    from gen6d_amd.network import name2network
    for name in dir(ops):
        if not name.startswith("_") and callable(getattr(ops, name)) and hasattr(ref_ops, name):
            setattr(ops, name, getattr(ref_ops, name))
    rank, world, local = parallel.init_from_env(backend="gloo")
    g = dict(np.load(%r))
    net = name2network["detector"]({"name": "t"}).eval()
    net.load_state_dict(synth.synth_state_dict("detector"))
    net.set_shard(rank, world)
    case = synth.detector_case(int(g["rfn"]), int(g["hq"]), int(g["wq"]))
    with torch.no_grad():
        out = net({"ref_imgs_info": {"imgs": case["ref_imgs"]}, "que_imgs_info": {"imgs": case["que_imgs"]}})
    assert net.ref_center_feats[0].shape[0] < int(g["rfn"])            # only the local references are resident
    for k in ("scores", "select_pr_offset", "select_pr_scale"):
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-3, atol=1e-3 * np.abs(g[k]).max())
    assert np.array_equal(out["que_select_id"].numpy(), g["que_select_id"])
    print("rank", rank, "sharded detector ok")
""")


def test_reference_sharded_detector_matches_golden(tmp_path):
    """8 references sharded over 3 gloo ranks (3+3+2): all-reduce(MAX) of the score features reproduces the reference."""
    script = tmp_path / "det_worker.py"
    script.write_text(DET_WORKER % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", "det_small.npz")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                          "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("sharded detector ok") == 3


BATCH_SHARD_WORKER = textwrap.dedent("""
    import sys, numpy as np, torch
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import ref_ops
    from gen6d_amd import ops, parallel, synth
    from gen6d_amd.network import name2network
    for name in dir(ops):                       # CPU emulation of the HIP ops (host-logic test)
        if not name.startswith("_") and callable(getattr(ops, name)) and hasattr(ref_ops, name):
            setattr(ops, name, getattr(ref_ops, name))
    rank, world, local = parallel.init_from_env(backend="gloo")
    g = dict(np.load(%r))
    rfn, an = int(g["rfn"]), int(g["an"])
    net = name2network["selector"]({"name": "t", "selector_angle_num": an}).eval()
    net.load_state_dict(synth.synth_state_dict("selector", an=an))
    net.set_shard(rank, world)
    case = synth.selector_case(rfn, an)
    # a BATCH of three queries through one set of launches and one set of collectives: the golden query first and last, another
    # image in between (its row must differ: the per-query statistics tables must not mix)
    other = synth.imgs_to_tensor(synth.synth_images(1, 128, 128, 977))
    ques = torch.cat([case["que_imgs"], other, case["que_imgs"]], 0)
    parallel.COLLECTIVE_LOG = []
    with torch.no_grad():
        net.extract_ref_feats(case["ref_imgs"], case["ref_poses"], case["object_center"], case["object_vert"])
        n_build = len(parallel.COLLECTIVE_LOG)
        logits, angles = net.compute_view_point_feats(ques)
    n_query = len(parallel.COLLECTIVE_LOG) - n_build
    parallel.COLLECTIVE_LOG = None
    b, e = parallel.shard_range(rfn, rank, world)
    assert net.ref_feats_cache[0].shape[0] == (e - b) * an          # only the local slice of the cache is resident
    assert logits.shape == (3, rfn) and angles.shape == (3, rfn)
    for row in (0, 2):
        np.testing.assert_allclose(logits[row:row + 1].numpy(), g["logits"], atol=2e-3)
        np.testing.assert_allclose(angles[row:row + 1].numpy(), g["angles"], atol=2e-3)
        assert int(logits[row].argmax()) == int(g["logits"].argmax(1)[0])
    assert float((logits[1] - logits[0]).abs().max()) > 1e-2
    assert n_build == 1 and n_query == 9, (n_build, n_query)        # 9 collectives for the whole batch: 3 per query here, 9/8 at 8 queries
    print("rank", rank, "batched sharded selector ok", n_query)
""")


@pytest.mark.parametrize("tag,world,port", [("sel_128x5", 8, 29651), ("sel_small", 3, 29653)])
def test_reference_sharded_selector_batch_of_queries(tmp_path, tag, world, port):
    """BASELINE configs[3]'s shape — 128 references x 5 rotations sharded 8-way (16 per rank) — and a ragged 3-rank split of 8
    references, each with a batch of three queries per call: every collective is shared by the batch (9 per batch), the rows of
    the golden query reproduce the reference's logits (tests/golden/sel_128x5.npz / sel_small.npz)."""
    script = tmp_path / "batch_shard_worker.py"
    script.write_text(BATCH_SHARD_WORKER % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", tag + ".npz")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("batched sharded selector ok") == world


@pytest.mark.parametrize("tag,world,port", [("det_mid", 4, 29661), ("det_mid", 5, 29663)])
def test_reference_sharded_detector_32_refs(tmp_path, tag, world, port):
    """32 references over 4 ranks (8 each) and, ragged, over 5 (7+7+6+6+6) against the reference's golden detection (det_mid: 160x192
    query): the local 15x15 levels fall back from the Winograd route (needs 32 local references) to the direct kernels, the
    all-reduce(MAX) of the score features restores the global max over references."""
    script = tmp_path / "det_worker.py"
    script.write_text(DET_WORKER % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", tag + ".npz")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("sharded detector ok") == world


FORCED_WORKER = textwrap.dedent("""
    import sys, numpy as np, torch
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import ref_ops
    from gen6d_amd import ops, parallel, synth
    from gen6d_amd.network import name2network
    for name in dir(ops):                       # CPU emulation of the HIP ops (host-logic test)
        if not name.startswith("_") and callable(getattr(ops, name)) and hasattr(ref_ops, name):
            setattr(ops, name, getattr(ref_ops, name))
    rank, world, local = parallel.init_from_env(backend="gloo", force=True)      # no launcher: a one-rank group on a free port
    assert (rank, world) == (0, 1) and parallel.backend_name() == "gloo"
    g = dict(np.load(%r))
    rfn, an = int(g["rfn"]), int(g["an"])
    case = synth.selector_case(rfn, an)
    other = synth.imgs_to_tensor(synth.synth_images(1, 128, 128, 977))
    ques = torch.cat([case["que_imgs"], other], 0)
    outs = {}
    for forced in (False, True):
        net = name2network["selector"]({"name": "t", "selector_angle_num": an}).eval()
        net.load_state_dict(synth.synth_state_dict("selector", an=an))
        net.set_shard(0, 1, force_collectives=forced)
        assert net.sharded == forced
        parallel.COLLECTIVE_LOG = []
        with torch.no_grad():
            net.extract_ref_feats(case["ref_imgs"], case["ref_poses"], case["object_center"], case["object_vert"])
            n_build = len(parallel.COLLECTIVE_LOG)
            outs[forced] = net.compute_view_point_feats(ques)
        n_query = len(parallel.COLLECTIVE_LOG) - n_build
        parallel.COLLECTIVE_LOG = None
        assert (n_build, n_query) == ((1, 9) if forced else (0, 0)), (forced, n_build, n_query)
    for a, b in zip(outs[True], outs[False]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), atol=1e-5)
    np.testing.assert_allclose(outs[True][0][0:1].numpy(), g["logits"], atol=2e-3)
    det = name2network["detector"]({"name": "t"}).eval()
    det.load_state_dict(synth.synth_state_dict("detector"))
    dc = synth.detector_case(8, 128, 128)
    res = {}
    for forced in (False, True):
        det.set_shard(0, 1, force_collectives=forced)
        parallel.COLLECTIVE_LOG = []
        with torch.no_grad():
            det.load_impl(dc["ref_imgs"])
            res[forced] = det.detect_impl(dc["que_imgs"])
        n = len(parallel.COLLECTIVE_LOG); parallel.COLLECTIVE_LOG = None
        assert n == (1 if forced else 0), (forced, n)
    assert torch.equal(res[True]["scores"], res[False]["scores"]) and torch.equal(res[True]["que_select_id"], res[False]["que_select_id"])
    print("forced collectives at world size 1 ok")
""")


def test_forced_collectives_on_a_one_rank_group(tmp_path):
    """`set_shard(0, 1, force_collectives=True)` (round 5: how the sharded path meets RCCL on a 1-GPU box, tests/test_rccl_world1_gpu.py)
    on a one-rank gloo group: 1 collective at build time and 9 per batch of queries in the selector, 1 in the detector, none without the
    flag, and the same logits / angles / scores either way."""
    script = tmp_path / "forced_worker.py"
    script.write_text(FORCED_WORKER % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", "sel_small.npz")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "forced collectives at world size 1 ok" in out.stdout
