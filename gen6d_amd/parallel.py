"""Multi-GPU plumbing: queries are independent given the per-object reference state, so the path
shards by QUERY — every rank holds a replica of the reference state (feature caches 220 MB + weights 0.3 GB, trivial
next to 288 GB of HBM), takes a contiguous slice of the query stream, and no collective sits on the data path.
The only exchanges are a barrier / MAX-reduce for timing and an all-gather of the KB-sized per-query results.
One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run). Returns (rank, world, local_rank).
    force=True creates the process group at world size 1 as well (a one-rank RCCL communicator accepts every collective: the
    reference-sharded path can be exercised, and captured in a hipGraph, on a 1-GPU box — `bench.py --shard-refs --gpus 1`)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("G6D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        if world == 1 and "MASTER_PORT" not in os.environ:
            # no launcher: a one-rank group rendezvous through a private file (picking "a free port" and re-binding it later is a race
            # between processes that do this at the same time, e.g. parallel pytest workers — ADVICE r05)
            import tempfile
            fd, path = tempfile.mkstemp(prefix="g6d_pg_")
            os.close(fd)
            os.unlink(path)                                 # FileStore wants to create the file itself
            dist.init_process_group(backend=backend, rank=0, world_size=1, init_method="file://" + path)
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


_LANE_GROUPS = []


def lane_groups(n):
    """One process group (under RCCL: one communicator) per hipGraph lane.  Collectives are ordered WITHIN a communicator only, so two
    batches in flight on different streams may run their collectives in either order as long as every lane's graph enqueues on its own
    communicator: lane i's replays are issued in the same order on every rank, which is all RCCL asks for.  Lane 0 uses the default
    group (None); the others are created once, collectively (every rank must call this with the same n), and reused by later captures."""
    if not dist.is_initialized():
        return [None] * n
    while len(_LANE_GROUPS) < n - 1:
        _LANE_GROUPS.append(dist.new_group(backend=dist.get_backend()))
    return [None] + _LANE_GROUPS[:n - 1]


def quiesce_watchdogs():
    """Call after torch.cuda.synchronize() and before a hipGraph capture.  Every RCCL process group has a watchdog thread that polls the
    end events of its eager collectives (every 100 ms) until it has seen them complete.  Once a capture has pulled the group's internal stream
    into capture mode, HIP refuses the query of an event last recorded on that stream — also of one recorded BEFORE the capture — and the
    watchdog takes the process down (hipErrorCapturedEvent; about half of the runs of tests/test_rccl_world1_gpu.py with three lanes on three
    communicators).  After a device synchronise all eager collectives have completed, so two poll intervals later no watchdog holds an event."""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        import time
        time.sleep(0.3)


def shard_range(n_items, rank, world):
    """Contiguous, balanced [begin, end) slice of n_items for this rank (first n_items % world ranks get one more)."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def _coll_device(device):
    """gloo moves data through host memory; RCCL ("nccl") works on device tensors."""
    return "cpu" if (dist.is_initialized() and dist.get_backend() == "gloo") else device


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    """All-reduce(SUM) of a scalar; bench.py uses it with 1 to report how many ranks really took part."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def backend_name():
    return dist.get_backend() if dist.is_initialized() else "none"


def gather_rows(local_rows, n_items):
    """All-gather per-query result rows [n_local, F] into [n_items, F] in global query order (ragged shards padded)."""
    if not dist.is_initialized():
        return local_rows
    world, rank = dist.get_world_size(), dist.get_rank()
    out_device = local_rows.device
    local_rows = local_rows.to(_coll_device(local_rows.device))
    width = local_rows.shape[1]
    cap = (n_items + world - 1) // world
    pad = torch.zeros((cap, width), dtype=local_rows.dtype, device=local_rows.device)
    pad[:local_rows.shape[0]] = local_rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = []
    for r in range(world):
        b, e = shard_range(n_items, r, world)
        out.append(bufs[r][:e - b])
    return torch.cat(out, 0).to(out_device)


# When set to a list, every data-path collective appends (kind, bytes, seconds, seconds until the call returned): `bench.py --shard-refs` reports the count per query
# and the mean time per collective.  Host wall time around the call: with RCCL the call only ENQUEUES on the stream (no host staging,
# no synchronisation), so a stream synchronisation brackets the call while logging; under gloo the tensor goes through the host.
COLLECTIVE_LOG = None


def _timed(kind, t, fn):
    if COLLECTIVE_LOG is None:
        return fn()
    import time
    if t.is_cuda:
        torch.cuda.synchronize(t.device)
    t0 = time.perf_counter()
    out = fn()
    t1 = time.perf_counter()                                  # the call has returned: enqueued (RCCL) or done (gloo)
    if t.is_cuda:
        torch.cuda.synchronize(t.device)
    COLLECTIVE_LOG.append((kind, t.numel() * t.element_size(), time.perf_counter() - t0, t1 - t0))
    return out


def all_reduce_(t, op="sum", group=None):
    """In-place all-reduce of a tensor that may live on the GPU (RCCL: issued on the device tensor, stream-ordered, no host staging)
    or must be staged through the host (gloo)."""
    if not dist.is_initialized():
        return t
    rop = dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX
    dev = _coll_device(t.device)

    def run():
        if str(dev) == "cpu" and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, op=rop, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=rop, group=group)
        return t
    return _timed("all_reduce_" + op, t, run)


def all_gather_ragged_rows(rows, n_total, world, group=None):
    """[n_local, F] per rank (contiguous shard_range slices) -> [n_total, F] in global order."""
    return _timed("all_gather_rows", rows, lambda: _all_gather_ragged_rows(rows, n_total, world, group))


def _all_gather_ragged_rows(rows, n_total, world, group=None):
    out_device = rows.device
    rows = rows.to(_coll_device(rows.device))
    cap = (n_total + world - 1) // world
    if n_total % world == 0 and rows.is_cuda:
        # even shards under RCCL: one all_gather_into_tensor straight into the result (fixed shapes, device tensors, issued on the
        # current stream: capturable in a hipGraph together with the kernels around it)
        out = torch.empty((n_total, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(out, rows.contiguous(), group=group)
        return out
    pad = torch.zeros((cap, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = []
    for r in range(world):
        b, e = shard_range(n_total, r, world)
        out.append(bufs[r][:e - b])
    return torch.cat(out, 0).to(out_device)
