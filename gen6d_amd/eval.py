"""Evaluation / streaming driver — the I/O path around `Gen6DEstimator` (reference eval.py:88-152,
predict.py:49-72), rebuilt around the device-resident chain:

  * query images are decoded and uploaded by a small thread pool `prefetch` images ahead of the GPU (the reference decodes
    one JPEG, runs the query, writes two JPEGs, in sequence on one thread);
  * queries run through `Gen6DEstimator.predict_many`-style lanes (whole captured chain per lane, several in flight), one
    host synchronisation per finished lane instead of five per query;
  * the per-query visualisation JPEGs (reference eval.py:129-132) are opt-in (`--vis`), not on the critical path;
  * the metrics of the reference (`compute_metrics_impl`, utils/pose_utils.py:149-215: ADD-0.1d, Prj-5, optional symmetric
    ADD) are evaluated for all queries at once on the GPU.

CLI mirrors the reference:  python -m gen6d_amd.eval --cfg configs/gen6d_pretrain.yaml --object_name linemod/cat
(run from the reference checkout for its dataset classes: `dataset.database.parse_database_name` is used when importable;
`--object_name synthetic/blob` evaluates on the procedural database of gen6d_amd/synth_db.py and needs no data)."""
import argparse
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import estimator as E


# ------------------------------------------------------------------------------------------------ metrics on the device
def compute_metrics(object_pts, diameter, pose_gt_list, pose_pr_list, Ks, scale=1.0, symmetric=False, device=None, return_errors=False):
    """reference utils/pose_utils.py:149-215 for all queries at once: {'add-0.1d', 'prj-5'[, 'add-0.1d-sym']}; return_errors: also
    the per-query (projection error px, ADD error) arrays (compute_pose_errors, pose_utils.py:149-175).  Checked against the
    reference's own function through tests/golden/geometry.npz (tests/test_eval_cpu.py)."""
    dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
    f = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.float64)).to(dev)
    pts, gt, pr, K = f(object_pts), f(pose_gt_list), f(pose_pr_list), f(Ks)

    def transform(p):                                                       # [q,n,3]
        return pts[None] @ p[:, :, :3].transpose(1, 2) + p[:, None, :, 3]

    def project(p):
        c = transform(p) @ K.transpose(1, 2)
        d = c[..., 2:3]
        d = torch.where((d.abs() < 1e-4) & (d.abs() > 0), torch.full_like(d, 1e-4), d)      # base_utils.py:256-265
        return c[..., :2] / d

    p3_pr, p3_gt = transform(pr), transform(gt)
    prj_err = (project(pr) - project(gt)).norm(dim=-1).mean(1)
    obj_err = (p3_pr - p3_gt).norm(dim=-1).mean(1) * scale
    res = {"add-0.1d": float((obj_err < diameter * 0.1).double().mean()), "prj-5": float((prj_err < 5).double().mean())}
    if symmetric:
        # nearest-neighbour ADD per query as in the reference (pose_utils.py:191-196); a [q,n,n] distance tensor for all queries
        # at once is ~140 GB at 4096 model points x 1000 queries, so the queries go through in chunks of <= 256 MB
        n = pts.shape[0]
        step = max(1, int((256 << 20) // max(8 * n * n, 1)))
        sym = torch.cat([torch.cdist(p3_pr[i:i + step], p3_gt[i:i + step]).min(2)[0].mean(1) for i in range(0, p3_pr.shape[0], step)]) * scale
        res["add-0.1d-sym"] = float((sym < diameter * 0.1).double().mean())
    if return_errors:
        return res, prj_err.cpu().numpy(), obj_err.cpu().numpy()
    return res


# ------------------------------------------------------------------------------------------------ image decode / video tracking
def decode_image(path):
    """JPEG / PNG file -> uint8 [H,W,3] (the reference reads frames with skimage.io.imread, eval.py:123 via database.get_image and
    predict.py:51; PIL is the decoder underneath).  Runs in the prefetch threads of run_queries (PIL releases the GIL while decoding)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB")))


class JpegFolderDatabase:
    """A database whose images live as JPEG files on disk (LINEMOD / GenMOP layout: one file per view) and are decoded on every
    get_image — the I/O the reference's eval loop pays per query.  Wraps any database object for poses / intrinsics / splits;
    `export` writes the wrapped database's images as JPEGs once."""

    def __init__(self, database, folder, quality=95, export=True):
        import os
        from PIL import Image
        self.database, self.folder = database, folder
        os.makedirs(folder, exist_ok=True)
        if export:
            for i in database.get_img_ids():
                path = os.path.join(folder, f"{i}.jpg")
                if not os.path.exists(path):
                    Image.fromarray(np.asarray(database.get_image(i))).save(path, quality=quality)

    def get_image(self, i):
        import os
        return decode_image(os.path.join(self.folder, f"{i}.jpg"))

    def __getattr__(self, name):                    # poses, intrinsics, splits, object geometry: the wrapped database's
        return getattr(self.database, name)


def pseudo_K(h, w):
    """predict.py:53-56: intrinsics assumed for an uncalibrated video frame."""
    f = np.sqrt(h ** 2 + w ** 2)
    return np.asarray([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)


def track_frames(estimator, frames, Ks=None, device_resident=True):
    """The video loop of the reference's predict.py:49-60: the first frame is detected + selected + refined `refine_iter` times,
    every later frame starts from the previous frame's pose and is refined ONCE (`pose_init`, refine_iter = 1).  frames: iterable of
    uint8 [H,W,3] arrays (or file paths -> decode_image); Ks: intrinsics per frame (None: predict.py's pseudo K).
    device_resident: the pose stays on the GPU from frame to frame (DeviceChain.query(pose_init=...): no host round trip per frame,
    one synchronisation at the end); otherwise the host-driven `predict(img, K, pose_init)` of the reference's loop.
    Returns poses [n,3,4] float32.  (The box smoothing + cv2.solvePnP of predict.py:66-72 is visualisation, not on this path.)"""
    frames = list(frames)
    imgs = [decode_image(f) if isinstance(f, (str, bytes)) or hasattr(f, "__fspath__") else np.asarray(f) for f in frames]
    Ks = [pseudo_K(*im.shape[:2]) if Ks is None else np.asarray(Ks[i], np.float32) for i, im in enumerate(imgs)]
    if not device_resident:
        poses, pose, it0 = [], None, estimator.cfg["refine_iter"]
        try:
            for im, K in zip(imgs, Ks):
                if pose is not None:
                    estimator.cfg["refine_iter"] = 1        # predict.py:58: "we only refine one time after initialization"
                pose, _ = estimator.predict(im, K, pose_init=pose)
                poses.append(pose)
        finally:
            estimator.cfg["refine_iter"] = it0
        return np.stack(poses, 0).astype(np.float32)
    chain = estimator.device_chain()
    dev = estimator.device
    out, pose = [], None
    for im, K in zip(imgs, Ks):
        r = chain.query(torch.from_numpy(np.ascontiguousarray(im)).to(dev, non_blocking=True),
                        torch.from_numpy(np.ascontiguousarray(K, dtype=np.float32)).to(dev, non_blocking=True), pose_init=pose,
                        refine_iter=None if pose is None else 1)
        pose = r["pose"]
        out.append(pose)
    return torch.stack(out, 0).cpu().numpy().astype(np.float32)


# ------------------------------------------------------------------------------------------------ streaming evaluation
def run_queries(estimator, que_database, que_ids, lanes=3, prefetch=6, decode_threads=4, on_result=None, batch=1):
    """All queries of `que_ids` through the device chain: returns (poses [q,3,4] float32, seconds, inter list).
    Decode + upload run `prefetch` images ahead in `decode_threads` host threads; `lanes` captured graphs of `batch` queries each
    (the queries of a graph share every launch) keep lanes x batch queries in flight; results are read back one lane at a time (one
    synchronisation per batch, at the END of the chain)."""
    chain = estimator.device_chain()
    dev = estimator.device
    batch = max(1, min(8, int(batch)))
    prefetch = max(prefetch, 2 * batch)

    def fetch(i):
        img = np.ascontiguousarray(que_database.get_image(i))               # JPEG decode happens here (database's reader)
        t = torch.from_numpy(img.copy() if not img.flags.writeable else img)
        if dev.type == "cuda":
            t = t.pin_memory()
        return t, torch.from_numpy(np.ascontiguousarray(que_database.get_K(i), dtype=np.float32))

    poses, inters = [None] * len(que_ids), [None] * len(que_ids)
    if len(que_ids) == 0:
        return np.zeros((0, 3, 4), np.float32), 0.0, []
    with ThreadPoolExecutor(decode_threads) as pool:
        # graph capture (once per image size / lane configuration) is set-up, not streaming time: done on the first image's shape
        # BEFORE the clock starts and before any other decode is submitted, so that the reported rate includes every decode
        shape0 = tuple(fetch(que_ids[0])[0].shape)
        if (chain._lanes is None or len(chain._lanes) != lanes or getattr(chain, "_batch", 1) != batch or
                tuple(chain._lanes[0][2].shape[1:]) != shape0):
            chain.capture(shape0, lanes, batch=batch)
        torch.cuda.synchronize(dev) if dev.type == "cuda" else None
        t0 = time.perf_counter()
        futs = [pool.submit(fetch, i) for i in que_ids[:prefetch]]
        busy = [None] * lanes                                               # (event, rows, first query index, count)

        def finish(slot):
            ev, rows, q0, n = busy[slot]
            ev.synchronize()
            rr = rows.cpu().numpy()
            for k in range(n):
                r, qi = rr[k], q0 + k
                poses[qi] = r[:12].reshape(3, 4).astype(np.float32)
                inters[qi] = {"det_position": r[12:14], "det_scale_r2q": float(r[14]), "sel_ref_idx": int(r[17]), "sel_angle_r2q": float(r[18])}
                if on_result is not None:
                    on_result(qi, poses[qi], inters[qi])
            busy[slot] = None

        for bi, q0 in enumerate(range(0, len(que_ids), batch)):
            n = min(batch, len(que_ids) - q0)
            imgs, Ks = [], []
            for qi in range(q0, q0 + n):
                img, K = futs[qi].result()
                futs[qi] = None                       # only the `prefetch` images ahead stay alive (pinned host memory)
                if qi + prefetch < len(que_ids):
                    futs.append(pool.submit(fetch, que_ids[qi + prefetch]))
                imgs.append(img); Ks.append(K)
            shape = tuple(imgs[0].shape)
            if any(tuple(im.shape) != shape for im in imgs):
                raise ValueError("run_queries: the images of one launch batch must share one size (use batch=1 for mixed-size query sets)")
            if (chain._lanes is None or len(chain._lanes) != lanes or getattr(chain, "_batch", 1) != batch or
                    tuple(chain._lanes[0][2].shape[1:]) != shape):
                chain.capture(shape, lanes, batch=batch)
            slot = bi % lanes
            if busy[slot] is not None:
                finish(slot)
            ib = torch.stack([im.to(dev, non_blocking=True) for im in imgs], 0)
            kb = torch.stack([K.to(dev, non_blocking=True) for K in Ks], 0)
            rows, stream = chain.enqueue(slot, ib, kb)
            ev = torch.cuda.Event(); ev.record(stream)
            busy[slot] = (ev, rows, q0, n)
        for slot in range(lanes):
            if busy[slot] is not None:
                finish(slot)
    return np.stack(poses, 0), time.perf_counter() - t0, inters


def get_ref_point_cloud(db):
    """reference dataset/database.py:327-344 by attribute; databases without a model get a coarse sphere of their diameter."""
    mod = E._reference_database_module(db)
    if mod is not None:
        return np.asarray(mod.get_ref_point_cloud(db), np.float32)
    for name in ("model", "object_point_cloud", "model_verts"):
        if hasattr(db, name):
            return np.asarray(getattr(db, name), np.float32)
    meta = getattr(db, "meta_info", None)
    if meta is not None and hasattr(meta, "object_point_cloud"):
        return np.asarray(meta.object_point_cloud, np.float32)
    rng = np.random.RandomState(0)
    v = rng.randn(2048, 3); v /= np.linalg.norm(v, axis=1, keepdims=True)
    return (E.get_object_center(db)[None] + 0.5 * E.get_diameter(db) * v).astype(np.float32)


def open_databases(object_name):
    """(ref_database, que_database, ref split, que split) as reference eval.py:90-108."""
    if object_name.startswith("synthetic"):
        from .synth_db import SyntheticDatabase
        db = SyntheticDatabase(n_views=96, name=object_name)
        return db, db, "all", "all"
    import sys
    mod = sys.modules.get("dataset.database")
    if mod is None:
        import importlib
        mod = importlib.import_module("dataset.database")                   # only inside the reference checkout
    if object_name.startswith("linemod"):
        return mod.parse_database_name(object_name), mod.parse_database_name(object_name), "linemod_test", "linemod_test"
    if object_name.startswith("genmop"):
        return mod.parse_database_name(object_name + "-ref"), mod.parse_database_name(object_name + "-test"), "all", "all"
    raise NotImplementedError(object_name)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=str, required=True, help="estimator YAML as in the reference (configs/gen6d_pretrain.yaml); "
                                                           "'synth' = seeded synthetic weights")
    ap.add_argument("--object_name", type=str, default="synthetic/blob")
    ap.add_argument("--symmetric", action="store_true")
    ap.add_argument("--split_type", type=str, default=None)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="queries per captured graph (they share every launch), 1..8")
    ap.add_argument("--prefetch", type=int, default=12)
    ap.add_argument("--max_queries", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=1, help="run the query list this many times (throughput measurements on small sets)")
    ap.add_argument("--jpeg_dir", type=str, default=None,
                    help="serve the query images from JPEG files in this folder (written once from the database): every query then pays "
                         "a real JPEG decode in the prefetch threads, as the reference's loop does on LINEMOD / GenMOP")
    ap.add_argument("--json", type=str, default=None, help="also write the summary (metrics, images/s, build seconds) to this file")
    args = ap.parse_args(argv)
    ref_db, que_db, ref_split, que_split = open_databases(args.object_name)
    if args.jpeg_dir:
        que_db = JpegFolderDatabase(que_db, args.jpeg_dir)
    if args.cfg == "synth":
        from . import synth
        from .network import name2network
        mods = {}
        for k in ("detector", "selector", "refiner"):
            net = name2network[k]({"name": k + "_synth"}).eval()
            net.load_state_dict(synth.synth_state_dict(k))
            mods[k] = net.cuda()
        est = E.Gen6DEstimator({"name": "gen6d_synth"}, modules=mods)
    else:
        import yaml
        with open(args.cfg) as f:
            est = E.Gen6DEstimator(yaml.load(f, Loader=yaml.FullLoader))
    tb = time.perf_counter()
    est.build(ref_db, args.split_type or ref_split)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - tb
    _, que_ids = E.get_database_split(que_db, que_split)
    if args.max_queries:
        que_ids = que_ids[:args.max_queries]
    que_ids = list(que_ids) * max(1, args.repeat)
    poses, secs, _ = run_queries(est, que_db, list(que_ids), args.lanes, args.prefetch, batch=args.batch)
    res = compute_metrics(get_ref_point_cloud(ref_db), E.get_diameter(que_db), [que_db.get_pose(i) for i in que_ids], poses,
                          [que_db.get_K(i) for i in que_ids], symmetric=args.symmetric)
    name = est.cfg.get("name", "gen6d") + (args.split_type or "")
    msg = f"{args.object_name:10} {name:20} " + " ".join(f"{k} {v:.4f}" for k, v in res.items())
    print(msg)
    print(f"build {build_s:.2f} s; {len(que_ids)} queries in {secs:.2f} s = {len(que_ids) / secs:.1f} images/s "
          f"(decode + upload + detect + select + {est.cfg['refine_iter']} x refine, {args.lanes} graphs of {args.batch} queries in flight)")
    if args.json:
        import json
        with open(args.json, "w") as f:
            json.dump({"object": args.object_name, "cfg": args.cfg, "metrics": res, "queries": len(que_ids), "seconds": secs,
                       "images_per_s": len(que_ids) / secs, "build_s": build_s, "lanes": args.lanes, "batch": args.batch, "prefetch": args.prefetch,
                       "jpeg_decode": bool(args.jpeg_dir), "refine_iter": est.cfg["refine_iter"]}, f, indent=1)
    return res


if __name__ == "__main__":
    main()
