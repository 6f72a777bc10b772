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


def make_estimator(device="cpu", refine_iter=1):
    mods = {}
    for k in ("detector", "selector", "refiner"):
        net = name2network[k]({"name": k + "_synth"}).eval()
        net.load_state_dict(synth.synth_state_dict(k))
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
