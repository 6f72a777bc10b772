"""The device pose algebra (gen6d_amd/csrc/pose_algebra.h, used by the kernels of pose_chain.hip) built for the host and
checked against gen6d_amd/geometry.py (itself pinned to the reference's utils through tests/golden/geometry.npz) and against
those golden vectors directly.  No GPU needed: the header is plain C++ in float64."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from gen6d_amd import geometry as G
from gen6d_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = C.POINTER(C.c_double)


def _p(a):
    a = np.ascontiguousarray(a, np.float64)
    return a.ctypes.data_as(D), a


@pytest.fixture(scope="module")
def pa(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("pa") / "pose_algebra.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "pose_algebra_shim.cpp"), "-o", so], check=True)
    lib = C.CDLL(so)
    lib.t_view_cos.restype = C.c_double
    return lib


def _call(fn, ins, outs):
    """ins: floats or arrays; outs: shapes -> list of numpy outputs."""
    args, keep = [], []
    for v in ins:
        if np.isscalar(v):
            args.append(C.c_double(v) if not isinstance(v, (int, np.integer)) else C.c_int(int(v)))
        else:
            ptr, arr = _p(v); keep.append(arr); args.append(ptr)
    res = [np.zeros(s, np.float64) for s in outs]
    fn(*args, *[r.ctypes.data_as(D) for r in res])
    return res


def _cams(n=12):
    poses, Ks = synth.fibonacci_cameras(n, radius=3.0, focal=250.0, size=160)
    return poses.astype(np.float64), Ks.astype(np.float64)


def test_against_golden(golden, pa):
    g = golden("geometry")
    poses, Ks, c = g["poses"], g["Ks"], g["center"]
    for i in range(4):
        (out,) = _call(pa.t_pose_from_similarity, [150.0 + 3 * i, 110.0 - 2 * i, 0.8 + 0.1 * i, 0.3 * i - 0.5, poses[i], Ks[i],
                                                   Ks[0] * np.array([[1.3], [1.3], [1]]), c], [(3, 4)])
        np.testing.assert_allclose(out, g["est_pose"][i], atol=1e-6)
    for i in range(6):
        (sa,) = _call(pa.t_scale_rot, [poses[i], poses[6 + i], Ks[i], Ks[6 + i], c], [(2,)])
        np.testing.assert_allclose(sa, [g["scale_diff"][i], g["angle_diff"][i]], atol=1e-7)
    (sim,) = _call(pa.t_compose_sim, [1.17, g["quat"], float(g["offset"][0]), float(g["offset"][1]), poses[3], c], [(3, 4)])
    np.testing.assert_allclose(sim, g["sim_pose"], atol=1e-7)
    (rig,) = _call(pa.t_sim_to_rigid, [g["sim_pose"], poses[3], Ks[3], Ks[3], c], [(3, 4)])
    np.testing.assert_allclose(rig, g["rigid_pose"], atol=1e-6)
    off = np.array([0.2, -0.1, 0.05])
    (npose,) = _call(pa.t_norm_pose, [poses[2], 1.7, off, 0], [(3, 4)])
    np.testing.assert_allclose(npose, g["norm_pose"], atol=1e-6)
    (dpose,) = _call(pa.t_norm_pose, [g["norm_pose"], 1.7, off, 1], [(3, 4)])
    np.testing.assert_allclose(dpose, g["denorm_pose"], atol=1e-6)
    K_new, pose_new, rect, H = _call(pa.t_look_at_crop, [Ks[5], poses[5], 70.0, 66.0, 0.4, 1.3, 128.0, 128.0], [(3, 3), (3, 4), (3, 4), (3, 3)])
    np.testing.assert_allclose(K_new, g["lac_K"], atol=1e-4)
    np.testing.assert_allclose(pose_new, g["lac_pose"], atol=1e-5)
    np.testing.assert_allclose(rect, g["lac_rect"], atol=1e-6)
    np.testing.assert_allclose(H, g["lac_H"], rtol=1e-4, atol=1e-4)
    (M,) = _call(pa.t_crop_transform, [55.0, 42.0, 0.7, 0.25, 128.0], [(3, 3)])
    np.testing.assert_allclose(M[:2], g["crop_M"], atol=1e-4)


def test_polar_factor_matches_svd(pa):
    rng = np.random.RandomState(0)
    for k in range(20):
        A = rng.randn(3, 3) * (0.2 if k % 3 == 0 else 1.0)
        if k == 5:
            A = 1.3 * G.quat2mat(rng.randn(4))             # the case that occurs: scale * rotation
        Q, msv = _call(pa.t_polar, [A], [(3, 3), (1,)])
        U, S, Vt = np.linalg.svd(A)
        np.testing.assert_allclose(Q, U @ Vt, atol=1e-9)
        np.testing.assert_allclose(msv[0], S.mean(), rtol=1e-10)


def test_refinement_step_geometry_matches_host_path(pa):
    """refine_prepare / align_reference / refine_update against the host sequence of VolumeRefiner.refine_que_imgs."""
    poses, Ks = _cams(20)
    rng = np.random.RandomState(3)
    center, diameter = np.array([0.05, -0.02, 0.03]), 1.3
    nscale, noff = 2 / diameter, -(2 / diameter) * center
    size, margin = 128, 0.05
    que_K = Ks[0] * np.array([[1.2], [1.2], [1.0]])
    in_pose_db = synth.perturb_pose(poses[4].astype(np.float32), 4.0, 0.02).astype(np.float64)
    in_pose_db[:, 3] -= in_pose_db[:, :3] @ center * 0          # poses look at the origin; the object centre is offset
    # host path (gen6d_amd/network/refiner.py: refine_que_imgs)
    in_pose = G.normalize_pose(in_pose_db, nscale, noff).astype(np.float64)
    c0 = np.zeros(3)
    _, new_f = G.let_me_look_at(in_pose, que_K, c0)
    in_dist = np.linalg.norm(G.pose_inverse(in_pose)[:, 3] - c0)
    scale = size * (1 - margin) / 2.0 * in_dist / new_f
    position = G.project_points(c0[None], in_pose, que_K)[0][0]
    K_warp, pose_warp, pose_rect, H = G.look_at_crop_params(que_K, in_pose, position, 0, scale, size, size)
    Kw, pw, pr, Hh = _call(pa.t_refine_prepare, [in_pose_db, que_K, nscale, noff, float(size), margin], [(3, 3), (3, 4), (3, 4), (3, 3)])
    np.testing.assert_allclose(Kw, K_warp, rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(pw, pose_warp, atol=1e-6)
    np.testing.assert_allclose(pr, pose_rect, atol=1e-6)
    np.testing.assert_allclose(Hh, H, rtol=1e-5, atol=1e-5)
    # reference alignment for a few views of the normalised database
    for i in (1, 7, 13):
        rp = G.normalize_pose(poses[i], nscale, noff).astype(np.float64)
        cen = G.project_points(c0[None], rp, Ks[i])[0][0]
        dist = np.linalg.norm(G.pose_inverse(rp)[:, 3])
        f_look = G.let_me_look_at(rp, Ks[i], c0)[1]
        _, ang = G.scale_rotation_difference_from_cameras(rp[None], pose_warp[None].astype(np.float64), Ks[i][None], K_warp[None].astype(np.float64), c0)
        K_new, pose_new, _, Hr = G.look_at_crop_params(Ks[i], rp, cen, ang[0], size * (1 - margin) / 2.0 * dist / f_look, size, size)
        Kn, pn, Hn = _call(pa.t_align_reference, [rp, Ks[i], pose_warp, K_warp, float(size), margin], [(3, 3), (3, 4), (3, 3)])
        np.testing.assert_allclose(Kn, K_new, rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(pn, pose_new, atol=1e-5)
        np.testing.assert_allclose(Hn, Hr, rtol=1e-4, atol=1e-4)
        assert abs(pa.t_view_cos(_p(pose_warp)[0], _p(rp)[0]) - G.view_correlation(pose_warp[None], rp[None], c0)[0, 0]) < 1e-9
    # pose update
    quat, off, ls = rng.randn(4), rng.randn(2) * 0.05, 0.13
    sim = G.compose_sim_pose(2 ** ls, quat, off, pose_warp, c0)
    pr_h = G.pose_sim_to_pose_rigid(sim, pose_warp, K_warp, K_warp, c0)
    pr_h = G.pose_compose(pr_h, G.pose_inverse(pose_rect))
    want = G.denormalize_pose(pr_h, nscale, noff)
    (got,) = _call(pa.t_refine_update, [quat, float(off[0]), float(off[1]), ls, K_warp, pose_warp, pose_rect, nscale, noff], [(3, 4)])
    np.testing.assert_allclose(got, want, atol=2e-6)
