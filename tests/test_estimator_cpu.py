"""Gen6DEstimator.build / predict on the procedural database with the HIP ops emulated on CPU (plumbing, contracts,
determinism).  The same flow runs on the GPU in tests/test_estimator_gpu.py."""
import numpy as np
import pytest
import torch

import ref_ops
from gen6d_amd import synth
from gen6d_amd.estimator import Gen6DEstimator, name2estimator
from gen6d_amd.network import name2network
from gen6d_amd.synth_db import SyntheticDatabase


def make_estimator(device="cpu", refine_iter=1, damped=False):
    """damped: the refiner's pose heads predict small residuals around the identity (synth.damp_refiner_head), as a trained
    refiner does — estimator-level comparisons then hold to 1e-4 instead of being dominated by the random head's amplification."""
    mods = {}
    for k in ("detector", "selector", "refiner"):
        net = name2network[k]({"name": k + "_synth"}).eval()
        sd = synth.synth_state_dict(k)
        net.load_state_dict(synth.damp_refiner_head(sd) if (damped and k == "refiner") else sd)
        mods[k] = net.to(device)
    return Gen6DEstimator({"ref_view_num": 8, "det_ref_view_num": 8, "refine_iter": refine_iter}, modules=mods)


def test_build_and_predict_contract(monkeypatch):
    ref_ops.patch_ops(monkeypatch)
    assert name2estimator["gen6d"] is Gen6DEstimator
    db = SyntheticDatabase(n_views=24, size=(96, 128), focal=140.0)
    est = make_estimator()
    est.build(db, "all")
    info = est.ref_info
    assert info["imgs"].shape == (8, 128, 128, 3) and info["imgs"].dtype == np.uint8
    assert info["ref_imgs"].shape == (5, 8, 128, 128, 3)
    assert info["poses"].shape == (8, 3, 4) and info["Ks"].shape == (8, 3, 3)
    # the normalised reference views look at the object: its centre projects to the crop centre
    from gen6d_amd import geometry as G
    for p, K in zip(info["poses"], info["Ks"]):
        np.testing.assert_allclose(G.project_points(info["center"][None].astype(np.float64), p, K)[0][0], [64, 64], atol=1e-3)
    _, que_ids = db.get_split("all")
    img, K = db.get_image(que_ids[0]), db.get_K(que_ids[0])
    pose, inter = est.predict(img, K)
    assert pose.shape == (3, 4) and np.isfinite(pose).all()
    assert set(inter) >= {"det_position", "det_scale_r2q", "det_que_img", "sel_angle_r2q", "sel_scores", "sel_ref_idx", "refine_poses"}
    assert inter["det_que_img"].shape == (128, 128, 3) and inter["sel_scores"].shape == (8,)
    assert len(inter["refine_poses"]) == 2
    np.testing.assert_allclose(pose[:, :3] @ pose[:, :3].T, np.eye(3), atol=1e-4)       # a rigid pose comes back
    pose2, _ = est.predict(img, K)
    np.testing.assert_allclose(pose, pose2, atol=1e-5)                                   # deterministic
    # pose_init skips detection/selection (predict.py:56-59)
    pose3, inter3 = est.predict(img, K, pose_init=db.get_pose(que_ids[0]))
    assert "det_position" not in inter3 and pose3.shape == (3, 4) and np.isfinite(pose3).all()


def test_device_chain_matches_host_predict(monkeypatch):
    """DeviceChain.query (the launch sequence of the device-resident predict) with every op emulated on the CPU — the chain
    ops by the host pose algebra — reproduces Gen6DEstimator.predict: same detection, same viewpoint, same refined pose."""
    ref_ops.patch_ops(monkeypatch)
    db = SyntheticDatabase(n_views=24, size=(96, 128), focal=140.0)
    est = make_estimator(refine_iter=1)
    est.build(db, "all")
    _, que_ids = db.get_split("all")
    img, K = db.get_image(que_ids[2]), db.get_K(que_ids[2])
    pose_h, inter_h = est.predict(img, K)
    pose_d, inter_d = est.predict_device(img, K)
    assert inter_d["sel_ref_idx"] == inter_h["sel_ref_idx"]
    np.testing.assert_allclose(inter_d["det_position"], inter_h["det_position"], atol=1e-3)
    assert len(inter_d["refine_poses"]) == 2
    np.testing.assert_allclose(inter_d["refine_poses"][0], inter_h["refine_poses"][0], atol=1e-5)     # pose from detection + selection
    # one refinement step from the same pose: the crops differ by single grey levels (float32 vs float64 homographies), which
    # the randomly initialised refiner amplifies; iterating it is chaotic (see test_estimator_gpu.py), so one step is compared
    np.testing.assert_allclose(pose_d, pose_h, atol=2e-2)


def test_reference_feature_cache(monkeypatch):
    """with `ref_feat_cache_deg` > 0 the alignment angle of every reference view is snapped to that grid, so the
    features of its aligned crop depend on (view, bucket) only and are reused between the refinement steps of a query and between
    queries.  Cached results equal the uncached evaluation of the SAME (snapped) geometry; the host-driven predict and the device
    chain agree; the second query hits the cache."""
    ref_ops.patch_ops(monkeypatch)
    db = SyntheticDatabase(n_views=24, size=(96, 128), focal=140.0)
    est = make_estimator(refine_iter=2, damped=True)
    est.refiner.cfg["ref_feat_cache_deg"] = 3.0
    est.build(db, "all")
    _, que_ids = db.get_split("all")
    img, K = db.get_image(que_ids[1]), db.get_K(que_ids[1])
    fc = est.refiner.feat_cache
    pose_a, inter_a = est.predict(img, K)                        # cold cache: every (view, bucket) pair is computed once
    cold_miss, cold_hit = fc.misses, fc.hits
    assert cold_miss >= 6 and len(fc.store) == cold_miss
    pose_b, inter_b = est.predict(img, K)                        # warm: no reference crop goes through the feature net again
    assert fc.misses == cold_miss and fc.hits == cold_hit + 12
    np.testing.assert_allclose(pose_b, pose_a, atol=1e-6)
    # the same snapped geometry evaluated WITHOUT the cache (features recomputed from the crops every step)
    feats_cached = {k: v.clone() for k, v in fc.store.items()}
    fc.clear()
    monkeypatch.setattr(est.refiner, "cached_ref_feats", lambda keys, make: est.refiner.run_feature_net(make(list(range(len(keys))))))
    pose_u, _ = est.predict(img, K)
    np.testing.assert_allclose(pose_u, pose_a, atol=1e-4)
    monkeypatch.undo(); ref_ops.patch_ops(monkeypatch)
    # device chain: snapped angles inside the chain kernels, eager run with the cache == run without it; keys are (view, bucket)
    fc.clear()
    pose_d0, _ = est.predict_device(img, K)
    pose_d1, _ = est.predict_device(img, K, use_feat_cache=True)
    pose_d2, _ = est.predict_device(img, K, use_feat_cache=True)
    assert fc.hits >= 12 and 0.0 < fc.hit_rate < 1.0
    # (two steps: a 1e-6 difference of the first step's pose can flip the uint8 rounding of single pixels of the next query crop)
    np.testing.assert_allclose(pose_d1, pose_d0, atol=1e-4)
    np.testing.assert_allclose(pose_d2, pose_d1, atol=1e-6)
    # snapping is opt-in: the default configuration keeps the reference's exact alignment and caches nothing
    est2 = make_estimator(refine_iter=1, damped=True)
    est2.build(db, "all")
    est2.predict(img, K)
    assert est2.refiner.angle_step() == 0.0 and len(est2.refiner.feat_cache.store) == 0


def test_device_chain_batch_matches_single_queries(monkeypatch):
    """DeviceChain.query_batch: B queries through one launch of every chain kernel / network stage (blockIdx = query, the
    networks' batched paths, one warp launch per stage) give the poses, detections and selections of the single-query chain."""
    ref_ops.patch_ops(monkeypatch)
    db = SyntheticDatabase(n_views=24, size=(96, 128), focal=140.0)
    est = make_estimator(refine_iter=1, damped=True)
    est.build(db, "all")
    _, que_ids = db.get_split("all")
    ids = [que_ids[1], que_ids[4], que_ids[2]]
    imgs = torch.stack([torch.from_numpy(np.ascontiguousarray(db.get_image(i))) for i in ids], 0)
    Ks = torch.stack([torch.from_numpy(np.ascontiguousarray(db.get_K(i), dtype=np.float32)) for i in ids], 0)
    chain = est.device_chain()
    outb = chain.query_batch(imgs, Ks)
    assert outb["pose"].shape == (3, 3, 4) and outb["det"].shape == (3, 5) and outb["sel"].shape == (3, 2) and outb["logits"].shape == (3, 8)
    for b in range(3):
        one = chain.query(imgs[b], Ks[b])
        assert int(outb["sel"][b, 0]) == int(one["sel"][0])
        np.testing.assert_allclose(outb["det"][b].numpy(), one["det"].numpy(), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(outb["refine_poses"][0][b].numpy(), one["refine_poses"][0].numpy(), atol=1e-5)
        np.testing.assert_allclose(outb["pose"][b].numpy(), one["pose"].numpy(), atol=1e-4)
