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
    est = make_estimator("cuda", refine_iter=1)
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
    np.testing.assert_allclose(pose_d, pose_h, atol=2e-2)           # one step of the randomly initialised refiner (see CPU test)


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
        # replay vs eager: statistics atomics reorder (1e-5), a detection that moves by 1e-4 px flips the rounding of a few crop
        # pixels, and the randomly initialised refiner amplifies single grey levels to ~1e-2 (same effect as in the CPU test)
        np.testing.assert_allclose(pm, pe, atol=3e-2)


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
        np.testing.assert_allclose(poses[i], pm, atol=3e-2)
    res = EV.compute_metrics(EV.get_ref_point_cloud(db), db.object_diameter, [db.get_pose(i) for i in que_ids], poses,
                             [db.get_K(i) for i in que_ids])
    assert set(res) == {"add-0.1d", "prj-5"} and all(0.0 <= v <= 1.0 for v in res.values())
