"""Host pose algebra (gen6d_amd/geometry.py, gen6d_amd/estimator.py helpers) against (a) the reference's own utility
functions run by tests/golden/make_golden.py (tests/golden/geometry.npz; transforms3d pieces provided there from
textbook definitions since the package is not installed) and (b) closed-form known answers."""
import numpy as np
import pytest

from gen6d_amd import estimator as E
from gen6d_amd import geometry as G


def test_against_reference_functions(golden):
    g = golden("geometry")
    poses, Ks, c = g["poses"], g["Ks"], g["center"]
    for i in range(4):
        est = G.estimate_pose_from_similarity_transform_compose(np.array([150.0 + 3 * i, 110.0 - 2 * i]), 0.8 + 0.1 * i,
                                                                0.3 * i - 0.5, poses[i], Ks[i],
                                                                Ks[0] * np.array([[1.3], [1.3], [1]]), c)
        np.testing.assert_allclose(est, g["est_pose"][i], atol=1e-5)
        R, f = G.let_me_look_at(poses[i], Ks[i], c)
        np.testing.assert_allclose(R, g["look_R"][i], atol=1e-6)
        np.testing.assert_allclose(f, g["look_f"][i], rtol=1e-7)
    sd, ad = G.scale_rotation_difference_from_cameras(poses[:6], poses[6:12], Ks[:6], Ks[6:12], c)
    np.testing.assert_allclose(sd, g["scale_diff"], rtol=1e-6)
    np.testing.assert_allclose(ad, g["angle_diff"], atol=1e-6)
    sim = G.compose_sim_pose(1.17, g["quat"], g["offset"], poses[3], c)
    np.testing.assert_allclose(sim, g["sim_pose"], atol=1e-6)
    np.testing.assert_allclose(G.pose_sim_to_pose_rigid(sim, poses[3], Ks[3], Ks[3], c), g["rigid_pose"], atol=1e-6)
    assert np.array_equal(G.sample_fps_points(g["fps_pts"], 33, True), g["fps_idx"])
    np.testing.assert_allclose(G.view_correlation(poses[:3], poses[3:], c), g["corr"], atol=1e-7)
    off = np.array([0.2, -0.1, 0.05])
    np.testing.assert_allclose(G.normalize_pose(poses[2], 1.7, off), g["norm_pose"], atol=1e-6)
    np.testing.assert_allclose(G.denormalize_pose(g["norm_pose"], 1.7, off), g["denorm_pose"], atol=1e-6)
    K_new, pose_new, rect, H = G.look_at_crop_params(Ks[5], poses[5], np.array([70.0, 66.0]), 0.4, 1.3, 128, 128)
    np.testing.assert_allclose(K_new, g["lac_K"], rtol=1e-5)
    np.testing.assert_allclose(pose_new, g["lac_pose"], atol=1e-5)
    np.testing.assert_allclose(rect, g["lac_rect"], atol=1e-6)
    np.testing.assert_allclose(H, g["lac_H"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(G.crop_transform(np.array([55.0, 42.0]), 0.7, 0.25, 128), g["crop_M"], atol=1e-4)

    class DB:
        object_center = c
        def get_pose(self, i): return poses[int(i)].astype(np.float32)
    ids = [str(i) for i in range(40)]
    got = E.select_reference_img_ids_refinement(DB(), c, ids, poses[7].astype(np.float32), 6, True, 16)
    assert np.array_equal(got.astype(np.int64), g["refine_ids"])
    cache = E.DeviceImageCache("cpu")                       # memoised FPS subset: same answer, keyed by the id list
    db = DB()
    for _ in range(2):
        got = E.select_reference_img_ids_refinement(db, c, ids, poses[7].astype(np.float32), 6, True, 16, cache)
        assert np.array_equal(got.astype(np.int64), g["refine_ids"])
    got2 = E.select_reference_img_ids_refinement(db, c, ids[::-1], poses[7].astype(np.float32), 6, True, 16, cache)
    assert len(cache._subsets) == 2 and set(got2) <= set(ids)   # another id list of the same length is another entry


def test_closed_forms():
    # look_at_rotation = R_x(atan y) R_y(-atan x): exact on the axes; off-axis the reference's second angle uses
    # atan2(y, 1) instead of atan2(y, sqrt(1+x^2)), a quirk that is kept (the golden test pins it)
    for p in (np.array([0.3, 0.0]), np.array([0.0, -0.2])):
        v = G.look_at_rotation(p) @ np.array([p[0], p[1], 1.0])
        np.testing.assert_allclose(v[:2], 0, atol=1e-12)
        assert v[2] > 0
    v = G.look_at_rotation(np.array([0.3, -0.2])) @ np.array([0.3, -0.2, 1.0])
    assert abs(v[0]) < 1e-12 and abs(v[1]) < 0.01
    # angle_about_z inverts rot_z, also behind an x/y rotation on the left
    for a in (-2.5, -0.3, 0.0, 1.1, 3.0):
        assert abs(G.angle_about_z(G.rot_z(a)) - a) < 1e-12
        assert abs(G.angle_about_z(G.rot_x(0.4) @ G.rot_y(-0.7) @ G.rot_z(a)) - a) < 1e-12
    # quaternion of a rotation about z by 90 degrees
    np.testing.assert_allclose(G.quat2mat([np.sqrt(0.5), 0, 0, np.sqrt(0.5)]), G.rot_z(np.pi / 2), atol=1e-12)
    np.testing.assert_allclose(G.quat2mat([2, 0, 0, 0]), np.eye(3), atol=1e-12)        # normalised internally
    # identity refiner residual leaves the pose unchanged
    pose = np.concatenate([G.rot_x(0.3) @ G.rot_y(0.5), [[0.1], [-0.2], [4.0]]], 1)
    K = np.array([[300.0, 0, 64], [0, 300.0, 64], [0, 0, 1]])
    c = np.array([0.05, 0.02, -0.03])
    sim = G.compose_sim_pose(1.0, [1, 0, 0, 0], np.zeros(2), pose, c)
    np.testing.assert_allclose(G.pose_sim_to_pose_rigid(sim, pose, K, K, c), pose, atol=1e-10)
    # scale 2 halves the depth of the object centre
    sim2 = G.compose_sim_pose(2.0, [1, 0, 0, 0], np.zeros(2), pose, c)
    rigid = G.pose_sim_to_pose_rigid(sim2, pose, K, K, c)
    np.testing.assert_allclose(G.pose_apply(rigid, c)[2], G.pose_apply(pose, c)[2] / 2, rtol=1e-10)
    # pose round trips
    np.testing.assert_allclose(G.pose_compose(pose, G.pose_inverse(pose)), np.eye(4)[:3], atol=1e-12)
    off = np.array([0.3, 0.1, -0.2])
    np.testing.assert_allclose(G.denormalize_pose(G.normalize_pose(pose, 2.5, off), 2.5, off), pose, atol=1e-6)
    # a normalised reference view (object centre on the optical axis) selected with scale 1 and angle 0 at its own
    # centre reproduces its own pose
    R = pose[:, :3]
    look = np.concatenate([R, (np.array([0, 0, 4.0]) - R @ c)[:, None]], 1)
    cen = G.project_points(c[None], look, K)[0][0]
    np.testing.assert_allclose(cen, [64, 64], atol=1e-9)
    est = G.estimate_pose_from_similarity_transform_compose(cen, 1.0, 0.0, look, K, K, c)
    np.testing.assert_allclose(est, look, atol=1e-8)
    # 2-D similarity algebra
    m = G.crop_transform(np.array([10.0, 20.0]), 2.0, 0.5, 128)
    np.testing.assert_allclose(G.sim2d_apply(m, np.array([[10.0, 20.0]])), [[64, 64]], atol=1e-9)
    np.testing.assert_allclose(G.sim2d_compose(m, G.sim2d_inverse(m)), np.eye(3)[:2], atol=1e-9)


def test_fps_is_a_permutation_prefix():
    pts = np.random.RandomState(0).randn(50, 3)
    idx = G.sample_fps_points(pts, 20, True)
    assert len(idx) == 19 and len(set(idx.tolist())) == 19
