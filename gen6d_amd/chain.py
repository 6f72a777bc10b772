"""Device-resident `Gen6DEstimator.predict` (BASELINE configs[4]): detect -> crop -> select -> pose ->
refine_iter x (look-at crop, reference selection + alignment, refiner, pose update) as ONE chain of kernel launches with no
host synchronisation between the stages — the pose algebra and warp parameters the reference computes on the host
(estimator.py:173-216, network/refiner.py:275-341) run in the single-thread float64 kernels of csrc/pose_chain.hip.  The chain
is therefore capturable in a hipGraph, and `predict_many` keeps several queries in flight on separate streams, which is
how bench.py fills the chip.

Reference state resident on the GPU (built once per object by `DeviceChain(estimator)` after `estimator.build`):
  * detector / selector reference state inside the networks (as before) + the 64 selected views' poses and intrinsics;
  * for the refiner: the farthest-point subset (<= 128 views) of the reference images as one uint8 stack [n,H,W,3] plus
    their normalised poses and intrinsics — the reference re-reads these from disk on every refinement step.
"""
import itertools

import numpy as np
import torch

from . import estimator as E
from . import geometry as G
from . import ops


class DeviceChain:
    # constants of Gen6DEstimator.predict's refinement call (reference estimator.py:212-213 `refine_que_imgs(..., size=128,
    # ref_num=6, ref_even=True)`, margin 0.05 at network/refiner.py:281); the refiner crop size is NOT cfg['ref_resolution']
    REF_NUM = 6
    MARGIN = 0.05
    REFINE_SIZE = 128

    _TOKENS = itertools.count(1)

    def __init__(self, est, even_num=128):
        self.est = est
        # reference-feature cache keys carry the real view id and a token that is unique per chain (CPython may reuse a freed chain's
        # id(), and the same sub-index can mean another view after a rebuild on the same database: ADVICE r03)
        self.cache_token = next(DeviceChain._TOKENS)
        self.dev = est.device
        self.size = int(est.cfg["ref_resolution"])                      # selector crop (estimator.py:184)
        self.refine_size = int(est.cfg.get("refine_size", self.REFINE_SIZE))
        self.refine_iter = int(est.cfg["refine_iter"]) if est.refiner is not None else 0
        info = est.ref_info
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.dev)
        self.ref_poses = f(info["poses"]).reshape(-1, 12)
        self.ref_Ks = f(info["Ks"]).reshape(-1, 9)
        self.center = f(info["center"])
        if self.refine_iter:
            db = est.refiner.ref_database
            ndb = E.NormalizedDatabase(db)
            ids_all = np.asarray(est.refiner.ref_ids)
            even = min(even_num, len(ids_all))
            poses = np.asarray([ndb.get_pose(i) for i in ids_all])
            cams = np.asarray([G.pose_inverse(p)[:, 3] for p in poses])
            sub = G.sample_fps_points(cams, even + 1, True)          # the `ref_even=True` subset of refine_que_imgs
            self.sub_ids = ids_all[sub]
            self.sub_poses = f(poses[sub]).reshape(-1, 12)
            self.sub_Ks = f(np.asarray([ndb.get_K(i) for i in self.sub_ids])).reshape(-1, 9)
            self.norm = f(np.concatenate([[ndb.scale], ndb.offset]))
            imgs = [np.ascontiguousarray(db.get_image(i)) for i in self.sub_ids]
            if len({im.shape for im in imgs}) != 1:
                raise ValueError("DeviceChain: the reference images of one object must share one size")
            self.stack = torch.from_numpy(np.stack(imgs, 0)).to(self.dev)
        self._lanes = None

    # ------------------------------------------------------------------ one query, no host synchronisation
    def query(self, que_img, que_K, use_feat_cache=False, pose_init=None, refine_iter=None):
        """que_img uint8 [H,W,3] and que_K float32 [3,3], both on the device -> dict of device tensors:
        'pose' [3,4], 'det' [5] (x, y, 2^scale, cell), 'sel' [2] (reference index, in-plane angle), 'logits' [rfn].
        With the refiner's cfg `ref_feat_cache_deg` > 0 the reference alignment angles are snapped to that grid (also inside a
        captured graph); use_feat_cache=True (eager only: one 48-byte read-back per refinement step to form the cache keys) then
        skips the warp + trunk + feature net of every reference crop whose (view, bucket) pair has been seen before."""
        est, size = self.est, self.size
        astep = est.refiner.angle_step() if est.refiner is not None else 0.0
        with torch.no_grad():
            que_K9 = que_K.reshape(9).contiguous()
            if pose_init is None:
                x = que_img.permute(2, 0, 1)[None].float().div_(255)
                det = est.detector.detect_impl(x.contiguous())
                det5 = torch.cat([det["positions"][0], det["scales"][0:1], det["que_select_id"][0].float()]).contiguous()
                crop = ops.warp_batch(None, que_img, None, ops.chain_crop_from_detection(det5, size), size, size)
                logits, angles = est.selector.compute_view_point_feats(crop)
                pose, sel = ops.chain_pose_from_selection(det5, logits[0].contiguous(), angles[0].contiguous(), self.ref_poses, self.ref_Ks,
                                                          que_K9, self.center)
            else:
                # tracking (reference predict.py:56-59, estimator.py:207-208): the previous frame's pose — a DEVICE tensor, no read-back —
                # replaces detection + selection
                pose = pose_init.reshape(3, 4).to(self.dev, torch.float32).contiguous()
                det5 = sel = crop = None
                logits = [None]
            poses = [pose]
            R, rs = self.REF_NUM, self.refine_size
            for _ in range(self.refine_iter if refine_iter is None else int(refine_iter)):
                prep = ops.chain_refine_prepare(pose.reshape(12), que_K9, self.norm, rs, self.MARGIN, self.sub_poses, self.sub_Ks, R,
                                                angle_step=astep)
                geo, idx = prep[0], prep[1]
                hinv = geo[33 + 21 * R:].view(1 + R, 9)
                imgs = torch.empty((1 + R, 3, rs, rs), dtype=torch.float32, device=self.dev)
                ops.warp_batch(None, que_img, None, hinv[0:1], rs, rs, out=imgs[0:1])
                geo_K, geo_P = geo[33:33 + 9 * R].view(R, 3, 3), geo[33 + 9 * R:33 + 21 * R].view(R, 3, 4)
                if use_feat_cache and astep > 0:
                    kb = torch.stack([idx, prep[2]], 0).cpu().numpy()            # the step's only host round trip: 2 x R ints
                    keys = [(self.cache_token, str(self.sub_ids[int(kb[0, k])]), int(kb[1, k]), rs) for k in range(R)]

                    def make(miss):
                        m = torch.tensor(miss, dtype=torch.long, device=self.dev)
                        return ops.warp_batch(self.stack, None, idx[m].contiguous(), hinv[1:][m].contiguous(), rs, rs)
                    ref_feats = est.refiner.cached_ref_feats(keys, make)
                    rot, off, scl = est.refiner._step(imgs[0:1], geo[0:9].view(3, 3), geo[9:21].view(3, 4), None, geo_K, geo_P,
                                                      ref_feats=ref_feats)
                else:
                    ops.warp_batch(self.stack, None, idx, hinv[1:], rs, rs, out=imgs[1:])
                    rot, off, scl = est.refiner._step(imgs[0:1], geo[0:9].view(3, 3), geo[9:21].view(3, 4), imgs[1:], geo_K, geo_P)
                pose = ops.chain_refine_update(rot[0].contiguous(), off[0].contiguous(), scl[0].contiguous(), geo, self.norm)
                poses.append(pose)
        return {"pose": pose, "det": det5, "sel": sel, "logits": logits[0], "refine_poses": poses, "crop": crop}

    # ------------------------------------------------------------------ a batch of queries, no host synchronisation
    def query_batch(self, que_imgs, que_Ks):
        """que_imgs uint8 [B,H,W,3], que_Ks float32 [B,3,3] on the device (B <= 8) -> dict of device tensors: 'pose' [B,3,4], 'det'
        [B,5], 'sel' [B,2], 'logits' [B,rfn], 'refine_poses' (list of [B,3,4]).  The B queries share every launch (round 3): the
        networks' batched paths, one launch of every chain kernel (blockIdx = query), one warp launch per stage for all crops —
        the estimator-level counterpart of `TensorPipeline.query` with the data flow between the stages real."""
        est, size = self.est, self.size
        astep = est.refiner.angle_step() if est.refiner is not None else 0.0
        B = que_imgs.shape[0]
        with torch.no_grad():
            ar = torch.arange(B, dtype=torch.int32, device=self.dev)
            K9 = que_Ks.reshape(B, 9).contiguous()
            x = que_imgs.permute(0, 3, 1, 2).float().div_(255)
            det = est.detector.detect_impl(x.contiguous())
            det5 = torch.cat([det["positions"], det["scales"][:, None], det["que_select_id"].float()], 1).contiguous()
            crop = ops.warp_batch(que_imgs, None, ar, ops.chain_crop_from_detection(det5, size), size, size)
            logits, angles = est.selector.compute_view_point_feats(crop)
            pose, sel = ops.chain_pose_from_selection(det5, logits.contiguous(), angles.contiguous(), self.ref_poses, self.ref_Ks, K9,
                                                      self.center)
            poses = [pose]
            R, rs = self.REF_NUM, self.refine_size
            for _ in range(self.refine_iter):
                prep = ops.chain_refine_prepare(pose.reshape(B, 12), K9, self.norm, rs, self.MARGIN, self.sub_poses, self.sub_Ks, R,
                                                angle_step=astep)
                geo, idx = prep[0], prep[1]
                hinv = geo[:, 33 + 21 * R:].reshape(B, 1 + R, 9)
                que_crops = ops.warp_batch(que_imgs, None, ar, hinv[:, 0].contiguous(), rs, rs)
                ref_crops = ops.warp_batch(self.stack, None, idx.reshape(-1), hinv[:, 1:].reshape(B * R, 9).contiguous(), rs, rs)
                rot, off, scl = est.refiner._step(que_crops, geo[:, 0:9].reshape(B, 3, 3), geo[:, 9:21].reshape(B, 3, 4),
                                                  ref_crops.view(B, R, 3, rs, rs), geo[:, 33:33 + 9 * R].reshape(B, R, 3, 3),
                                                  geo[:, 33 + 9 * R:33 + 21 * R].reshape(B, R, 3, 4))
                pose = ops.chain_refine_update(rot.contiguous(), off.contiguous(), scl.contiguous(), geo, self.norm)
                poses.append(pose)
        return {"pose": pose, "det": det5, "sel": sel, "logits": logits, "refine_poses": poses, "crop": crop}

    # ------------------------------------------------------------------ hipGraph lanes
    def capture(self, img_shape, lanes=3, warmup=2, batch=1):
        """One captured copy of the whole chain per lane (own static input / output buffers, shared read-only reference state); a
        lane's graph processes `batch` queries that share every launch."""
        d = self.dev
        old_serial, ops.SERIAL = ops.SERIAL, True          # whole queries in flight; no intra-query stream forks
        try:
            self._lanes, self._batch = [], int(batch)
            for _ in range(lanes):
                g_img = torch.zeros((batch,) + tuple(img_shape), dtype=torch.uint8, device=d)
                g_K = torch.eye(3, dtype=torch.float32, device=d).repeat(batch, 1, 1)
                g_K[:, 0, 0] = g_K[:, 1, 1] = 500.0; g_K[:, 0, 2] = img_shape[1] / 2; g_K[:, 1, 2] = img_shape[0] / 2
                stream = torch.cuda.Stream(device=d)
                stream.wait_stream(torch.cuda.current_stream(d))
                with torch.cuda.stream(stream):
                    for _ in range(warmup):
                        self.query_batch(g_img, g_K)
                torch.cuda.synchronize(d)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                    out = self.query_batch(g_img, g_K)
                self._lanes.append((graph, stream, g_img, g_K, out, None))
            torch.cuda.synchronize(d)
        finally:
            ops.SERIAL = old_serial
        return self

    def enqueue(self, lane, que_img, que_K):
        """Replay lane `lane` on its stream: que_img uint8 [H,W,3] (batch 1) or [B,H,W,3] with B <= the captured batch (the rest of the
        lane's slots keep their previous content and are ignored), que_K [3,3] / [B,3,3]; returns (rows [B,19] = pose(12) | det(5) |
        sel(2) — [19] for a single image —, stream).  The caller synchronises (event on the stream) before reusing the lane."""
        graph, stream, g_img, g_K, out, _ = self._lanes[lane]
        single = que_img.dim() == 3
        if single:
            que_img, que_K = que_img[None], que_K[None]
        n = que_img.shape[0]
        stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(stream):
            g_img[:n].copy_(que_img, non_blocking=True)
            g_K[:n].copy_(que_K, non_blocking=True)
            # the sources were allocated on the caller's stream and are consumed on the lane's: tell the caching allocator, or it
            # may hand their blocks to the next query's upload while this copy is still pending
            for t in (que_img, que_K):
                if t.is_cuda:
                    t.record_stream(stream)
            graph.replay()
            rows = torch.cat([out["pose"].reshape(-1, 12), out["det"], out["sel"]], 1)[:n]
        return (rows[0] if single else rows), stream

    def predict_many(self, que_imgs, que_Ks, lanes=3, batch=1):
        """Queries [(H,W,3) uint8 numpy or device tensors], intrinsics [3,3] -> list of (pose [3,4] float32 numpy, inter dict).
        `lanes` captured graphs of `batch` queries each are kept in flight; ONE host synchronisation at the end."""
        def as_tensor(q):                                # (decoded JPEG frames can be read-only views: torch wants writable memory)
            if torch.is_tensor(q):
                return q
            q = np.asarray(q)
            return torch.from_numpy(q if (q.flags.writeable and q.flags.c_contiguous) else np.array(q, order="C"))
        imgs = [as_tensor(q) for q in que_imgs]
        batch = max(1, min(8, int(batch)))
        if (self._lanes is None or len(self._lanes) != lanes or getattr(self, "_batch", 1) != batch or
                tuple(self._lanes[0][2].shape[1:]) != tuple(imgs[0].shape)):
            self.capture(tuple(imgs[0].shape), lanes, batch=batch)
        busy, rows = [None] * lanes, []
        for bi, i0 in enumerate(range(0, len(imgs), batch)):
            lane = bi % lanes
            if busy[lane] is not None:
                busy[lane].synchronize()
            ib = torch.stack([im.to(self.dev, non_blocking=True) for im in imgs[i0:i0 + batch]], 0)
            kb = torch.stack([(K if torch.is_tensor(K) else torch.from_numpy(np.ascontiguousarray(K, dtype=np.float32))).to(self.dev, non_blocking=True)
                              for K in que_Ks[i0:i0 + batch]], 0)
            r, stream = self.enqueue(lane, ib, kb)
            ev = torch.cuda.Event(); ev.record(stream)
            busy[lane] = ev
            rows.append(r)
        torch.cuda.synchronize(self.dev)
        out = []
        for r in torch.cat(rows, 0).cpu().numpy():
            out.append((r[:12].reshape(3, 4).astype(np.float32),
                        {"det_position": r[12:14], "det_scale_r2q": float(r[14]), "sel_ref_idx": int(r[17]), "sel_angle_r2q": float(r[18])}))
        return out
