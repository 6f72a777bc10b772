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
def compute_metrics(object_pts, diameter, pose_gt_list, pose_pr_list, Ks, scale=1.0, symmetric=False, device=None):
    """reference utils/pose_utils.py:149-215 for all queries at once: {'add-0.1d', 'prj-5'[, 'add-0.1d-sym']}."""
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
    return res


# ------------------------------------------------------------------------------------------------ streaming evaluation
def run_queries(estimator, que_database, que_ids, lanes=3, prefetch=6, decode_threads=4, on_result=None):
    """All queries of `que_ids` through the device chain: returns (poses [q,3,4] float32, seconds, inter list).
    Decode + upload run `prefetch` images ahead in `decode_threads` host threads; `lanes` captured graphs keep that many
    queries in flight; results are read back one lane at a time (one synchronisation per query, at the END of the chain)."""
    chain = estimator.device_chain()
    dev = estimator.device

    def fetch(i):
        img = np.ascontiguousarray(que_database.get_image(i))               # JPEG decode happens here (database's reader)
        t = torch.from_numpy(img)
        if dev.type == "cuda":
            t = t.pin_memory()
        return t, torch.from_numpy(np.ascontiguousarray(que_database.get_K(i), dtype=np.float32))

    poses, inters = [None] * len(que_ids), [None] * len(que_ids)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(decode_threads) as pool:
        futs = [pool.submit(fetch, i) for i in que_ids[:prefetch]]
        busy = [None] * lanes                                               # (event, row, query index)

        def finish(slot):
            ev, row, qi = busy[slot]
            ev.synchronize()
            r = row.cpu().numpy()
            poses[qi] = r[:12].reshape(3, 4).astype(np.float32)
            inters[qi] = {"det_position": r[12:14], "det_scale_r2q": float(r[14]), "sel_ref_idx": int(r[17]), "sel_angle_r2q": float(r[18])}
            if on_result is not None:
                on_result(qi, poses[qi], inters[qi])
            busy[slot] = None

        for qi in range(len(que_ids)):
            img, K = futs[qi].result()
            futs[qi] = None                       # only the `prefetch` images ahead stay alive (pinned host memory)
            if qi + prefetch < len(que_ids):
                futs.append(pool.submit(fetch, que_ids[qi + prefetch]))
            if chain._lanes is None or len(chain._lanes) != lanes or tuple(chain._lanes[0][2].shape) != tuple(img.shape):
                chain.capture(tuple(img.shape), lanes)
            slot = qi % lanes
            if busy[slot] is not None:
                finish(slot)
            row, stream = chain.enqueue(slot, img.to(dev, non_blocking=True), K.to(dev, non_blocking=True))
            ev = torch.cuda.Event(); ev.record(stream)
            busy[slot] = (ev, row, qi)
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
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--prefetch", type=int, default=6)
    ap.add_argument("--max_queries", type=int, default=0)
    args = ap.parse_args(argv)
    ref_db, que_db, ref_split, que_split = open_databases(args.object_name)
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
    poses, secs, _ = run_queries(est, que_db, list(que_ids), args.lanes, args.prefetch)
    res = compute_metrics(get_ref_point_cloud(ref_db), E.get_diameter(que_db), [que_db.get_pose(i) for i in que_ids], poses,
                          [que_db.get_K(i) for i in que_ids], symmetric=args.symmetric)
    name = est.cfg.get("name", "gen6d") + (args.split_type or "")
    msg = f"{args.object_name:10} {name:20} " + " ".join(f"{k} {v:.4f}" for k, v in res.items())
    print(msg)
    print(f"build {build_s:.2f} s; {len(que_ids)} queries in {secs:.2f} s = {len(que_ids) / secs:.1f} images/s "
          f"(decode + upload + detect + select + {est.cfg['refine_iter']} x refine, {args.lanes} queries in flight)")
    return res


if __name__ == "__main__":
    main()
