"""Device-resident predict chain on a real MI355X (csrc/pose_chain.hip + gen6d_amd/chain.py): the pose-algebra kernels and the
batched warp against the host implementations (tests/ref_ops.py -> gen6d_amd/geometry.py, pinned to the reference's utils by
tests/golden/geometry.npz), then the whole chain — eager and as three captured hipGraph lanes — against the host-driven
`Gen6DEstimator.predict` it replaces."""
import numpy as np
import pytest
import torch

import ref_ops
from gen6d_amd import synth
from gen6d_amd.synth_db import SyntheticDatabase
from test_estimator_cpu import make_estimator

pytestmark = pytest.mark.gpu


def _dev(t):
    return t.cuda() if torch.is_tensor(t) else torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).cuda()


def test_chain_kernels_match_host_algebra():
    from gen6d_amd import lib, ops
    lib.load()
    poses, Ks = synth.fibonacci_cameras(40, radius=3.0, focal=250.0, size=160)
    rng = np.random.RandomState(1)
    center = np.array([0.04, -0.03, 0.02], np.float32)
    det = torch.tensor([83.5, 61.25, 1.37, 10.0, 7.0])
    h_ref, h_dev = ref_ops.chain_crop_from_detection(det, 128), ops.chain_crop_from_detection(det.cuda(), 128)
    np.testing.assert_allclose(h_dev.cpu().numpy(), h_ref.numpy(), rtol=1e-5, atol=1e-5)
    logits, angles = torch.from_numpy(rng.randn(40).astype(np.float32)), torch.from_numpy((rng.rand(40) - 0.5).astype(np.float32))
    logits[7] = logits[23] = logits.max() + 1                      # tie: the first maximum wins
    rp, rk = torch.from_numpy(poses).reshape(-1, 12), torch.from_numpy(Ks).reshape(-1, 9)
    qK = torch.from_numpy(Ks[0] * np.array([[1.2], [1.2], [1.0]], np.float32)).reshape(9)
    p_ref, s_ref = ref_ops.chain_pose_from_selection(det, logits, angles, rp, rk, qK, torch.from_numpy(center))
    p_dev, s_dev = ops.chain_pose_from_selection(det.cuda(), logits.cuda(), angles.cuda(), rp.cuda(), rk.cuda(), qK.cuda(), _dev(center))
    assert int(s_dev[0]) == 7 == int(s_ref[0])
    np.testing.assert_allclose(p_dev.cpu().numpy(), p_ref.numpy(), atol=2e-5)
    # refinement step geometry on a normalised database
    diameter = 1.3
    nscale, noff = 2 / diameter, -(2 / diameter) * center
    from gen6d_amd import geometry as G
    sub = np.stack([G.normalize_pose(p.astype(np.float64), nscale, noff) for p in poses]).astype(np.float32)
    norm = torch.from_numpy(np.concatenate([[nscale], noff]).astype(np.float32))
    pose_in = torch.from_numpy(synth.perturb_pose(poses[9], 5.0, 0.03)).reshape(12)
    g_ref, i_ref = ref_ops.chain_refine_prepare(pose_in, qK, norm, 128, 0.05, torch.from_numpy(sub).reshape(-1, 12), rk, 6)
    g_dev, i_dev = ops.chain_refine_prepare(pose_in.cuda(), qK.cuda(), norm.cuda(), 128, 0.05, _dev(sub.reshape(-1, 12)), rk.cuda(), 6)
    assert np.array_equal(i_dev.cpu().numpy(), i_ref.numpy())
    np.testing.assert_allclose(g_dev.cpu().numpy(), g_ref.numpy(), rtol=2e-4, atol=2e-4)
    rot = torch.nn.functional.normalize(torch.from_numpy(rng.randn(4).astype(np.float32)), dim=0)
    off, scl = torch.tensor([0.03, -0.02]), torch.tensor([0.21])
    # snapped alignment angles (reference-feature caching): same views, buckets equal to the host's, geometry of the snapped angle
    step = float(np.radians(3.0))
    gs_ref, is_ref, b_ref = ref_ops.chain_refine_prepare(pose_in, qK, norm, 128, 0.05, torch.from_numpy(sub).reshape(-1, 12), rk, 6, angle_step=step)
    gs_dev, is_dev, b_dev = ops.chain_refine_prepare(pose_in.cuda(), qK.cuda(), norm.cuda(), 128, 0.05, _dev(sub.reshape(-1, 12)), rk.cuda(), 6,
                                                    angle_step=step)
    assert np.array_equal(is_dev.cpu().numpy(), is_ref.numpy()) and np.array_equal(b_dev.cpu().numpy(), b_ref.numpy())
    np.testing.assert_allclose(gs_dev.cpu().numpy(), gs_ref.numpy(), rtol=2e-4, atol=2e-4)
    assert not np.allclose(gs_dev.cpu().numpy(), g_dev.cpu().numpy(), atol=1e-6)        # snapping did change the alignment
    u_ref = ref_ops.chain_refine_update(rot, off, scl, g_ref, norm)
    u_dev = ops.chain_refine_update(rot.cuda(), off.cuda(), scl.cuda(), g_ref.cuda(), norm.cuda())
    np.testing.assert_allclose(u_dev.cpu().numpy(), u_ref.numpy(), atol=2e-5)


def test_warp_batch_matches_single_warps():
    from gen6d_amd import ops
    imgs = torch.from_numpy(synth.synth_images(5, 96, 128, 31))
    que = torch.from_numpy(synth.synth_images(1, 96, 128, 32)[0])
    rng = np.random.RandomState(2)
    hinv = []
    for _ in range(4):
        a, s = rng.uniform(-0.6, 0.6), rng.uniform(0.6, 1.4)
        M = np.array([[s * np.cos(a), -s * np.sin(a), rng.uniform(-10, 30)], [s * np.sin(a), s * np.cos(a), rng.uniform(-10, 30)],
                      [rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-4, 1e-4), 1.0]])
        hinv.append(np.linalg.inv(M).reshape(9))
    hinv = torch.from_numpy(np.asarray(hinv, np.float32))
    idx = torch.tensor([-1, 3, 0, 4], dtype=torch.int32)
    got = ops.warp_batch(imgs.cuda(), que.cuda(), idx.cuda(), hinv.cuda(), 64, 64)
    want = ref_ops.warp_batch(imgs, que, idx, hinv, 64, 64)
    d = (got.cpu() - want).abs() * 255
    assert got.shape == (4, 3, 64, 64) and d.max() <= 1.001 and (d > 0.5).float().mean() < 0.01      # rounding ties only


@pytest.fixture(scope="module")
def built():
    db = SyntheticDatabase(n_views=24, size=(96, 128), focal=140.0)
    est = make_estimator("cuda", refine_iter=1, damped=True)      # pose heads around the identity, like a trained refiner (weak #2)
    est.build(db, "all")
    return db, est


def test_predict_device_matches_host_driven_predict(built):
    db, est = built
    _, que_ids = db.get_split("all")
    img, K = db.get_image(que_ids[2]), db.get_K(que_ids[2])
    pose_h, inter_h = est.predict(img, K)
    pose_d, inter_d = est.predict_device(img, K)
    assert inter_d["sel_ref_idx"] == inter_h["sel_ref_idx"]
    np.testing.assert_allclose(inter_d["det_position"], inter_h["det_position"], atol=1e-3)
    np.testing.assert_allclose(inter_d["refine_poses"][0], inter_h["refine_poses"][0], atol=2e-5)
    from parity_log import record
    record("test_predict_device_matches_host_driven_predict", "refined pose: device chain vs host-driven predict (1 step)", float(np.abs(pose_d - pose_h).max()), 1e-4)
    np.testing.assert_allclose(pose_d, pose_h, atol=1e-4)           # crops differ by single grey levels (float32 vs float64 homographies)


def test_predict_many_three_lanes(built):
    """Seven queries through three captured graphs of the whole chain: every result equals the eager chain's."""
    db, est = built
    _, que_ids = db.get_split("all")
    qs = [que_ids[i % len(que_ids)] for i in range(7)]
    imgs, Ks = [db.get_image(i) for i in qs], [db.get_K(i) for i in qs]
    eager = [est.predict_device(im, K) for im, K in zip(imgs, Ks)]
    many = est.predict_many(imgs, Ks, lanes=3)
    assert len(many) == 7
    for (pe, ie), (pm, im_) in zip(eager, many):
        assert im_["sel_ref_idx"] == ie["sel_ref_idx"]
        np.testing.assert_allclose(im_["det_position"], ie["det_position"], atol=1e-3)
        # replay vs eager: statistics atomics reorder (1e-5); a detection that moves by 1e-4 px can flip the rounding of a few crop pixels
        np.testing.assert_allclose(pm, pe, atol=3e-4)


def test_streaming_eval_driver(built):
    """gen6d_amd/eval.run_queries: prefetching decode threads + three lanes; poses equal predict_many's, metrics finite."""
    from gen6d_amd import eval as EV
    db, est = built
    _, que_ids = db.get_split("all")
    seen = []
    poses, secs, inters = EV.run_queries(est, db, list(que_ids), lanes=3, prefetch=4, decode_threads=2,
                                         on_result=lambda qi, p, it: seen.append(qi))
    assert poses.shape == (len(que_ids), 3, 4) and sorted(seen) == list(range(len(que_ids))) and secs > 0
    many = est.predict_many([db.get_image(i) for i in que_ids], [db.get_K(i) for i in que_ids], lanes=3)
    for i, (pm, im_) in enumerate(many):
        assert inters[i]["sel_ref_idx"] == im_["sel_ref_idx"]
        np.testing.assert_allclose(poses[i], pm, atol=3e-4)
    res = EV.compute_metrics(EV.get_ref_point_cloud(db), db.object_diameter, [db.get_pose(i) for i in que_ids], poses,
                             [db.get_K(i) for i in que_ids])
    assert set(res) == {"add-0.1d", "prj-5"} and all(0.0 <= v <= 1.0 for v in res.values())


def test_reference_feature_cache_gpu():
    """SURVEY.md 8f row 2 on the GPU: (a) the features of a reference crop do not depend on the batch it is extracted in (per-image
    InstanceNorm) — cached == recomputed to 1e-6; (b) with snapped alignment angles the cached eager chain reproduces the uncached one
    and the captured-graph lanes, later queries hit the cache."""
    from parity_log import record
    db = SyntheticDatabase(n_views=24, size=(96, 128), focal=140.0)
    est = make_estimator("cuda", refine_iter=2, damped=True)
    est.refiner.cfg["ref_feat_cache_deg"] = 3.0
    est.build(db, "all")
    _, que_ids = db.get_split("all")
    imgs = synth.imgs_to_tensor(synth.synth_images(7, 128, 128, seed=77)).cuda()
    with torch.no_grad():
        f_all = est.refiner.run_feature_net(imgs)
        f_refs = est.refiner.run_feature_net(imgs[:6].contiguous())
        f_one = est.refiner.run_feature_net(imgs[2:3].contiguous())
    e1 = (f_all[:6] - f_refs).abs().max().item() / f_all.abs().max().item()
    e2 = (f_all[2:3] - f_one).abs().max().item() / f_all.abs().max().item()
    # per-image InstanceNorm: the features of a crop do not depend on its batch; what differs is the split / tile choice of the conv
    # launches with N (fp32 reassociation, measured 1.3e-6 of the feature range)
    record("test_reference_feature_cache_gpu", "reference features: batch of 7 vs batch of 6 / 1 (relative to max)", max(e1, e2), 5e-6)
    assert max(e1, e2) <= 5e-6, (e1, e2)
    fc = est.refiner.feat_cache
    img, K = db.get_image(que_ids[1]), db.get_K(que_ids[1])
    pose_u, _ = est.predict_device(img, K)                                     # snapped angles, nothing cached
    assert len(fc.store) == 0
    pose_c, _ = est.predict_device(img, K, use_feat_cache=True)
    miss0 = fc.misses
    pose_c2, _ = est.predict_device(img, K, use_feat_cache=True)
    assert fc.misses == miss0 and fc.hits >= 12
    e = float(np.abs(pose_c - pose_u).max())
    record("test_reference_feature_cache_gpu", "pose: cached eager chain vs uncached (2 steps, damped head)", e, 1e-4)
    assert e <= 1e-4 and np.abs(pose_c2 - pose_c).max() <= 1e-6
    many = est.predict_many([img, img], [K, K], lanes=2)                       # captured graphs use the same snapped geometry
    assert np.abs(many[0][0] - pose_u).max() <= 1e-4
    pose_h, _ = est.predict(img, K)                                            # host-driven path with its own (view id, bucket) keys
    record("test_reference_feature_cache_gpu", "pose: host-driven cached predict vs device chain", float(np.abs(pose_h - pose_u).max()), 3e-4)
    assert np.abs(pose_h - pose_u).max() <= 3e-4


def test_streaming_eval_with_real_jpeg_decode_and_tracking(built, tmp_path):
    """f4 on the GPU: (a) run_queries over a database served from JPEG files (PIL decode in the prefetch threads, pinned upload,
    three lanes) gives the poses of predict_many on the decoded images and the reference's metrics are finite; (b) the
    device-resident tracking loop (predict.py:49-60) follows the host-driven one."""
    from gen6d_amd import eval as EV
    db, est = built
    jdb = EV.JpegFolderDatabase(db, str(tmp_path / "jpg"))
    _, que_ids = db.get_split("all")
    que_ids = list(que_ids)[:7]
    poses, secs, inters = EV.run_queries(est, jdb, que_ids, lanes=3, prefetch=4, decode_threads=3)
    imgs = [jdb.get_image(i) for i in que_ids]
    many = est.predict_many(imgs, [db.get_K(i) for i in que_ids], lanes=3)
    for i, (pm, im_) in enumerate(many):
        assert inters[i]["sel_ref_idx"] == im_["sel_ref_idx"]
        np.testing.assert_allclose(poses[i], pm, atol=3e-4)
    res, prj, obj = EV.compute_metrics(EV.get_ref_point_cloud(db), db.object_diameter, [db.get_pose(i) for i in que_ids], poses,
                                       [db.get_K(i) for i in que_ids], return_errors=True)
    assert np.isfinite(prj).all() and np.isfinite(obj).all() and set(res) == {"add-0.1d", "prj-5"}
    frames, Ks = imgs[:3], [db.get_K(i) for i in que_ids[:3]]
    host = EV.track_frames(est, frames, Ks, device_resident=False)
    dev = EV.track_frames(est, frames, Ks, device_resident=True)
    np.testing.assert_allclose(dev[0], host[0], atol=3e-4)                       # frame 0: detect + select + one step
    assert np.isfinite(dev).all() and np.abs(dev - host).max() < 5e-2            # later frames compound (see tests/test_eval_cpu.py)


def test_predict_many_batched_graphs(built):
    """Captured chain graphs of 4 queries each (two in flight): every result equals the single-query graphs' (batch = 1) and the
    eager chain's; a ragged tail (7 queries = 4 + 3) uses only the first slots of the last graph."""
    from parity_log import record
    db, est = built
    _, que_ids = db.get_split("all")
    qs = [que_ids[i % len(que_ids)] for i in range(7)]
    imgs, Ks = [db.get_image(i) for i in qs], [db.get_K(i) for i in qs]
    single = est.predict_many(imgs, Ks, lanes=2, batch=1)
    batched = est.predict_many(imgs, Ks, lanes=2, batch=4)
    assert len(batched) == 7
    worst = 0.0
    for (p1, i1), (pb, ib) in zip(single, batched):
        assert ib["sel_ref_idx"] == i1["sel_ref_idx"]
        np.testing.assert_allclose(ib["det_position"], i1["det_position"], atol=1e-3)
        worst = max(worst, float(np.abs(pb - p1).max()))
    record("test_predict_many_batched_graphs", "pose: chain graphs of 4 queries vs single-query graphs", worst, 3e-4)
    assert worst <= 3e-4
    # streaming driver on batches
    from gen6d_amd import eval as EV
    poses, secs, inters = EV.run_queries(est, db, qs, lanes=2, prefetch=8, decode_threads=2, batch=4)
    for i, (pb, ib) in enumerate(batched):
        assert inters[i]["sel_ref_idx"] == ib["sel_ref_idx"]
        np.testing.assert_allclose(poses[i], pb, atol=3e-4)
