"""Tensor-level Gen6D pipeline: detect -> select -> refine x N on device tensors (the per-query hot path of
reference estimator.py:173-216 without the host-side cv2 warps, which are SURVEY.md §8(f) "next" rows).
Used by bench.py, __graft_entry__.smoke() and the GPU tests; mirrors `Gen6DEstimator`'s stage order and defaults
(det 32 refs, sel 64 refs x 5 rotations, 6 refiner views, refine_iter 3)."""
import torch

from . import ops, parallel, synth
from .network import name2network
from .network import refiner as refiner_mod


class TensorPipeline:
    def __init__(self, device, sel_rfn=64, det_rfn=32, an=5, refine_iter=3, seed=1234, shard=(0, 1), force_collectives=False):
        self.device = torch.device(device)
        self.refine_iter = refine_iter
        self.cfg = dict(sel_rfn=sel_rfn, det_rfn=det_rfn, an=an, refine_iter=refine_iter)
        self.detector = name2network["detector"]({"name": "detector_synth"})
        self.selector = name2network["selector"]({"name": "selector_synth", "selector_angle_num": an})
        self.refiner = name2network["refiner"]({"name": "refiner_synth"})
        self.state_dicts = {k: synth.synth_state_dict(k, seed, an) for k in ("detector", "selector", "refiner")}
        for k, net in (("detector", self.detector), ("selector", self.selector), ("refiner", self.refiner)):
            net.load_state_dict(self.state_dicts[k])
            net.to(self.device).eval()
        # (rank, world): references of selector and detector sharded over the ranks; force_collectives: the sharded code path (9 + 1
        # collectives per batch) at world size 1 as well
        self.selector.set_shard(*shard, force_collectives=force_collectives)
        self.detector.set_shard(*shard, force_collectives=force_collectives)

    def build(self, seed=1):
        """Reference state from synthetic views (Gen6DEstimator.build, reference estimator.py:139-171)."""
        c = self.cfg
        self.sel_case = synth.selector_case(c["sel_rfn"], c["an"], seed)
        det_refs = self.sel_case["ref_imgs"][c["an"] // 2, :min(c["det_rfn"], c["sel_rfn"])]      # un-rotated copies
        self.det_refs = det_refs.contiguous()
        d = self.device
        with torch.no_grad():
            self.detector.load_impl(self.det_refs.to(d))
            self.selector.extract_ref_feats(self.sel_case["ref_imgs"].to(d), self.sel_case["ref_poses"].to(d),
                                            self.sel_case["object_center"].to(d), self.sel_case["object_vert"].to(d))
        self.ref_feats = None                        # features of the canned reference crops (query(..., cached_refs=True))
        self._canned_cache = {}
        self.ref_case = synth.refiner_case()
        self.ref_dev = {k: v.to(d) for k, v in self.ref_case.items()}
        # a slightly different input pose per refinement iteration (no cross-iteration caching possible)
        self.iter_poses = [torch.from_numpy(synth.perturb_pose(self.ref_case["poses_in"][0].numpy(), 2.0 * i, 0.01 * i))[None].to(d)
                           for i in range(self.refine_iter)]

    def _canned(self, n, it):
        """(Ks_in, pose_in of iteration `it`, ref_Ks, ref_poses) of the canned refinement case replicated for n queries (dense)."""
        key = (n, it)
        if key not in self._canned_cache:
            r = self.ref_dev
            ex = lambda t: t.expand(n, *t.shape[1:]).contiguous()
            self._canned_cache[key] = (ex(r["Ks_in"]), ex(self.iter_poses[it]), ex(r["ref_Ks"]), ex(r["ref_poses"]))
        return self._canned_cache[key]

    def query(self, que_full, que_crop, cached_refs=False):
        """que_full [qn,3,H,W] (detector input), que_crop [qn,3,128,128] (selector/refiner input), device tensors; the qn
        queries of the call share every launch (chunks of <= 16 / 32 / 32 queries inside the detector / selector / refiner).
        Returns [qn, 5 + 7 * refine_iter] rows: position(2), scale, ref_idx, angle, then quaternion(4), offset(2), log2-scale of EVERY
        refinement step in order (26 columns at 3 steps: a wrong first or second step shows in the row)."""
        r = self.ref_dev
        qn = que_crop.shape[0]
        with torch.no_grad():
            if cached_refs and getattr(self, "ref_feats", None) is None:
                # reference-feature caching: the canned reference crops are the same in every step of every
                # query, i.e. every (view, bucket) key hits — their features are computed once
                self.ref_feats = self.refiner.run_feature_net(r["ref_imgs"][0]).clone()
            det = self.detector.detect_impl(que_full)
            logits, angles = self.selector.compute_view_point_feats(que_crop)
            idx = torch.argmax(logits, 1)
            ang = angles.gather(1, idx[:, None])
            steps = []
            for it in range(self.refine_iter):
                if qn == 1:
                    rot, off, scl = self.refiner._step(que_crop, r["Ks_in"][0], self.iter_poses[it][0], r["ref_imgs"][0],
                                                       r["ref_Ks"][0], r["ref_poses"][0], ref_feats=self.ref_feats if cached_refs else None)
                else:                                  # every query of the batch with its own (here: the same canned) views / poses
                    rots, offs, scls = [], [], []
                    rb = refiner_mod.MAX_BATCH
                    for q0 in range(0, qn, rb):
                        n = min(rb, qn - q0)
                        ex = lambda t: t.expand(n, *t.shape[1:])
                        cam = self._canned(n, it)              # the canned cameras / poses of n queries: dense copies made once per n
                        o = self.refiner._step(que_crop[q0:q0 + n], cam[0], cam[1], ex(r["ref_imgs"]), cam[2], cam[3],
                                               ref_feats=self.ref_feats[None].expand(n, *self.ref_feats.shape) if cached_refs else None)
                        rots.append(o[0]); offs.append(o[1]); scls.append(o[2])
                    rot, off, scl = (ops.cat1(t, 0) for t in (rots, offs, scls))
                steps += [rot, off, scl]
        return torch.cat([det["positions"], det["scales"][:, None], idx[:, None].float(), ang] + steps, 1)

    # ------------------------------------------------------------------ hipGraph
    def capture(self, full_shape=(1, 3, 480, 640), crop_shape=(1, 3, 128, 128), warmup=2, lanes=1, batch=None, cached_refs=False):
        """Capture one query — or one batch of `batch` queries that share every launch — into a
        hipGraph with static input / output buffers.  `lanes` > 1 captures that many independent copies (own static
        buffers and intermediates, shared read-only reference state) so that `query_graph(..., lane=i)` can keep
        several queries in flight on different streams: the small grids of one query leave CUs idle that the next
        query fills."""
        d = self.device
        if batch is not None:
            full_shape, crop_shape = (batch,) + tuple(full_shape[1:]), (batch,) + tuple(crop_shape[1:])
        self._lanes = []
        # reference-sharded mode: every lane enqueues its collectives on its OWN communicator (parallel.lane_groups), so that several
        # batches can be in flight — the order of collectives only has to agree between ranks within a communicator
        sharded = self.selector.sharded or self.detector.sharded
        groups = parallel.lane_groups(lanes) if sharded else [None] * lanes
        for li in range(lanes):
            if sharded:
                self.selector.group = self.detector.group = groups[li]
            g_full = torch.zeros(full_shape, dtype=torch.float32, device=d)
            g_crop = torch.zeros(crop_shape, dtype=torch.float32, device=d)
            stream = torch.cuda.Stream(device=d)
            stream.wait_stream(torch.cuda.current_stream(d))
            with torch.cuda.stream(stream):
                for _ in range(warmup):                  # workspaces, statistics arenas and allocator warm-up off-graph
                    self.query(g_full, g_crop, cached_refs)
            torch.cuda.synchronize(d)
            parallel.quiesce_watchdogs()
            graph = torch.cuda.CUDAGraph()
            # thread_local: a process group's watchdog thread polls its work events while this thread captures; under the default
            # (global) capture mode such a query from another thread fails and takes the process down
            with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                g_out = self.query(g_full, g_crop, cached_refs)
            self._lanes.append((graph, stream, g_full, g_crop, g_out))
        if sharded:
            self.selector.group = self.detector.group = groups[0]      # eager calls between replays: the default communicator
        torch.cuda.synchronize(d)
        return self._lanes

    def query_graph(self, que_full, que_crop, lane=0):
        """Enqueue one query on lane `lane` (its own stream) and return its static output row; call
        `torch.cuda.synchronize()` (or sync the lane's stream) before reading, and before reusing the lane."""
        graph, stream, g_full, g_crop, g_out = self._lanes[lane]
        cur = torch.cuda.current_stream(self.device)
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            g_full.copy_(que_full, non_blocking=True)
            g_crop.copy_(que_crop, non_blocking=True)
            for t in (que_full, que_crop):           # allocated on the caller's stream, consumed on the lane's
                if t.is_cuda:
                    t.record_stream(stream)
            graph.replay()
            out = g_out.clone()
        return out, stream
