"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

/root/reference is a plain Python repo whose networks import packages that are absent here (torchvision, cv2,
skimage, plyfile, transforms3d).  They are only needed at import time (or for ImageNet weights that the synthetic
state_dict overwrites anyway), so they are replaced by the minimal stubs below — no reference file is edited
(SURVEY.md §8c).  The reference modules then receive `gen6d_amd.synth.synth_state_dict(...)` weights and the
seeded synthetic inputs, run on CPU through their cuda-free entry points, and the outputs are stored as small
fixtures.  The GPU box has no /root/reference: tests there read only the .npz files.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


def install_stubs():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class Normalize(torch.nn.Module):
        def __init__(self, mean, std):
            super().__init__()
            self.mean, self.std = mean, std

        def forward(self, x):
            m = torch.tensor(self.mean, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
            s = torch.tensor(self.std, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
            return (x - m) / s

    tr.Normalize = Normalize
    models = types.ModuleType("torchvision.models")
    resnet = types.ModuleType("torchvision.models.resnet")
    resnet.Bottleneck = resnet.BasicBlock = type("Block", (), {})
    resnet.conv1x1 = lambda *a, **k: None

    class _Empty:
        def state_dict(self):
            return {}

    models.vgg11_bn = lambda *a, **k: _Empty()
    models.resnet18 = lambda *a, **k: _Empty()
    models.resnet = resnet
    tv.transforms, tv.models = tr, models
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.models": models,
                        "torchvision.models.resnet": resnet})

    cv2 = types.ModuleType("cv2")
    for i, n in enumerate(("INTER_LINEAR", "INTER_NEAREST", "SOLVEPNP_ITERATIVE", "SOLVEPNP_EPNP")):
        setattr(cv2, n, i)
    sys.modules["cv2"] = cv2
    for name, attrs in {"plyfile": ["PlyData"], "skimage": [], "skimage.io": ["imread", "imsave"],
                        "transforms3d": [], "transforms3d.euler": ["euler2mat", "mat2euler"],
                        "transforms3d.axangles": ["mat2axangle"], "transforms3d.quaternions": ["quat2mat", "mat2quat"]}.items():
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        sys.modules[name] = m
    # transforms3d is not installed; the few conversions the reference's pose utilities call are provided here from
    # their textbook definitions (static-frame Euler sequences, Hamilton quaternions w-first), written independently
    # of gen6d_amd/geometry.py so that the golden vectors are not circular.
    def _axis_rot(axis, a):
        c, s_ = np.cos(a), np.sin(a)
        return {"x": np.array([[1, 0, 0], [0, c, -s_], [0, s_, c]]), "y": np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]),
                "z": np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])}[axis]

    def euler2mat(ai, aj, ak, axes="sxyz"):
        assert axes[0] == "s"
        R = np.eye(3)
        for ax, ang in zip(axes[1:], (ai, aj, ak)):      # static frame: later rotations multiply on the left
            R = _axis_rot(ax, ang) @ R
        return R

    def mat2euler(M, axes="szyx"):
        assert axes == "szyx"                            # M = Rx(ak) Ry(aj) Rz(ai)
        aj = np.arcsin(np.clip(M[0, 2], -1, 1))
        return np.arctan2(-M[0, 1], M[0, 0]), aj, np.arctan2(-M[1, 2], M[2, 2])

    def quat2mat(q):
        w, x, y, z = np.asarray(q, np.float64) / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def mat2axangle(M):
        M = np.asarray(M, np.float64)
        ang = np.arccos(np.clip((np.trace(M) - 1) / 2, -1, 1))
        ax = np.array([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
        n = np.linalg.norm(ax)
        return (ax / n if n > 1e-12 else np.array([1.0, 0, 0])), ang

    sys.modules["transforms3d.axangles"].mat2axangle = mat2axangle
    sys.modules["transforms3d.euler"].euler2mat = euler2mat
    sys.modules["transforms3d.euler"].mat2euler = mat2euler
    sys.modules["transforms3d.quaternions"].quat2mat = quat2mat
    sys.modules["cv2"].warpPerspective = lambda img, H, size, flags=0: np.zeros((size[1], size[0]) + img.shape[2:], img.dtype)
    sys.modules["cv2"].warpAffine = sys.modules["cv2"].warpPerspective
    sys.modules["skimage"].io = sys.modules["skimage.io"]
    for sub in ("euler", "axangles", "quaternions"):
        setattr(sys.modules["transforms3d"], sub, sys.modules[f"transforms3d.{sub}"])


def load_reference():
    install_stubs()
    # VGGBNPretrain._initialize_weights loads the (empty) stub state dict strictly -> make that a no-op
    sys.path.insert(0, REF)
    import network.pretrain_models as pm
    pm.VGGBNPretrain._initialize_weights = lambda self: None
    from network import name2network
    return name2network


def geometry_golden():
    """Host pose algebra of the reference (utils/pose_utils.py, utils/database_utils.py, utils/base_utils.py,
    dataset/database.py) on seeded inputs -> tests/golden/geometry.npz."""
    import numpy as np
    if not hasattr(np, "bool"): np.bool = bool           # the reference predates numpy 2
    if not hasattr(np, "str"): np.str = str
    from gen6d_amd import synth
    import utils.base_utils as bu
    import utils.pose_utils as pu
    import utils.database_utils as du
    from dataset.database import normalize_pose, denormalize_pose
    rng = np.random.RandomState(5)
    poses, Ks = synth.fibonacci_cameras(40, radius=3.0, focal=250.0, size=160)
    poses = poses.astype(np.float64); Ks = Ks.astype(np.float64)
    center = np.array([0.1, -0.05, 0.08])
    out = {"center": center, "poses": poses, "Ks": Ks}
    out["est_pose"] = np.stack([pu.estimate_pose_from_similarity_transform_compose(
        np.array([150.0 + 3 * i, 110.0 - 2 * i]), 0.8 + 0.1 * i, 0.3 * i - 0.5, poses[i], Ks[i], Ks[0] * np.array([[1.3], [1.3], [1]]), center)
        for i in range(4)])
    la = [pu.let_me_look_at(poses[i], Ks[i], center) for i in range(4)]
    out["look_R"], out["look_f"] = np.stack([a[0] for a in la]), np.asarray([a[1] for a in la])
    sd, ad = pu.scale_rotation_difference_from_cameras(poses[:6], poses[6:12], Ks[:6], Ks[6:12], center)
    out["scale_diff"], out["angle_diff"] = sd, ad
    quat = rng.randn(4); offset = rng.randn(2) * 0.05
    sim = pu.compose_sim_pose(1.17, quat, offset, poses[3], center)
    out["quat"], out["offset"], out["sim_pose"] = quat, offset, sim
    out["rigid_pose"] = pu.pose_sim_to_pose_rigid(sim, poses[3], Ks[3], Ks[3], center)
    pts = rng.randn(200, 3)
    out["fps_pts"], out["fps_idx"] = pts, bu.sample_fps_points(pts, 33, True, index_model=True)
    out["corr"] = du.compute_normalized_view_correlation(poses[:3], poses[3:], center, False)
    out["norm_pose"] = normalize_pose(poses[2], 1.7, np.array([0.2, -0.1, 0.05]))
    out["denorm_pose"] = denormalize_pose(out["norm_pose"], 1.7, np.array([0.2, -0.1, 0.05]))
    img = np.zeros((120, 160, 3), np.uint8)
    _, K_new, pose_new, pose_rect, H = du.look_at_crop(img, Ks[5], poses[5], np.array([70.0, 66.0]), 0.4, 1.3, 128, 128)
    out.update(lac_K=K_new, lac_pose=pose_new, lac_rect=pose_rect, lac_H=H)
    _, M = bu.transformation_crop(img, np.array([55.0, 42.0]), 0.7, 0.25, 128)
    out["crop_M"] = M

    class DB:
        def get_pose(self, i): return poses[int(i)].astype(np.float32)
    ids = [str(i) for i in range(40)]
    out["refine_ids"] = du.select_reference_img_ids_refinement(DB(), center, ids, poses[7].astype(np.float32), 6, True, 16).astype(np.int64)
    # evaluation metrics of the reference (utils/pose_utils.py:149-215) on perturbed poses: per-query errors and the summary, with
    # and without the symmetric ADD — the checker of gen6d_amd/eval.compute_metrics
    import importlib
    pu_mod = importlib.reload(pu) if getattr(pu, "mat2axangle", None) is None else pu      # bind the stubbed mat2axangle
    m_pts = rng.randn(300, 3) * 0.05
    m_gt = poses[:12].astype(np.float64)
    m_pr = np.stack([synth.perturb_pose(p.astype(np.float32), 0.4 * i, 0.002 * i).astype(np.float64) for i, p in enumerate(m_gt)])
    m_Ks = Ks[:12]
    diam = 0.3
    errs = np.asarray([pu_mod.compute_pose_errors(m_pts, pr, gt, K)[:2] for pr, gt, K in zip(m_pr, m_gt, m_Ks)], np.float64)
    res = pu_mod.compute_metrics_impl(m_pts, diam, m_gt, m_pr, m_Ks, 1.0, symmetric=True)
    res2 = pu_mod.compute_metrics_impl(m_pts, diam, m_gt, m_pr, m_Ks, 2.5, symmetric=False)
    out.update(met_pts=m_pts, met_gt=m_gt, met_pr=m_pr, met_Ks=m_Ks, met_diameter=diam, met_prj_err=errs[:, 0], met_obj_err=errs[:, 1],
               met_res=np.asarray([res["add-0.1d"], res["prj-5"], res["add-0.1d-sym"]]), met_res_scale25=np.asarray([res2["add-0.1d"], res2["prj-5"]]))
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **out)
    print("geometry golden ok", out["fps_idx"][:5], out["refine_ids"])


def np_(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def main():
    import warnings
    warnings.filterwarnings("ignore")
    from gen6d_amd import synth
    torch.manual_seed(0)
    torch.set_num_threads(8)
    name2network = load_reference()

    # ---- detector: 8 refs, 128x128 query; and 32 refs with a 160x192 query
    for tag, rfn, hq, wq in (("det_small", 8, 128, 128), ("det_mid", 32, 160, 192)):
        net = name2network["detector"]({"network": "detector", "name": "g"}).eval()
        net.load_state_dict(synth.synth_state_dict("detector"))
        case = synth.detector_case(rfn, hq, wq)
        with torch.no_grad():
            out = net({"ref_imgs_info": {"imgs": case["ref_imgs"]}, "que_imgs_info": {"imgs": case["que_imgs"]}})
            pos, scl = net.parse_detection(out["scores"], out["select_pr_scale"], out["select_pr_offset"], 8)
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), rfn=rfn, hq=hq, wq=wq, positions=pos.numpy(),
                            sha_inputs=synth.fingerprint(case), sha_weights=synth.fingerprint(synth.synth_state_dict("detector")),
                            scales=scl.numpy(), **np_({k: out[k] for k in ("scores", "select_pr_offset",
                                                                           "select_pr_scale", "que_select_id")}))
        print(tag, "pos", pos.numpy(), "scale", scl.numpy())

    # ---- selector: 8 refs x 5 rotations ; 16 refs x 5
    for tag, rfn, an in (("sel_small", 8, 5), ("sel_mid", 16, 5)):
        net = name2network["selector"]({"network": "selector", "name": "g", "selector_angle_num": an}).eval()
        net.load_state_dict(synth.synth_state_dict("selector", an=an))
        case = synth.selector_case(rfn, an)
        with torch.no_grad():
            out = net({"ref_imgs": case["ref_imgs"], "ref_imgs_info": {"poses": case["ref_poses"]},
                       "object_center": case["object_center"], "object_vert": case["object_vert"],
                       "que_imgs_info": {"imgs": case["que_imgs"]}, "eval": True})
            embed = net.ref_pose_embed
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), rfn=rfn, an=an, logits=out["ref_vp_logits"].numpy(),
                            sha_inputs=synth.fingerprint(case), sha_weights=synth.fingerprint(synth.synth_state_dict("selector", an=an)),
                            angles=out["angles_pr"].numpy(), pose_embed=embed.numpy()[:, :16])
        print(tag, "argmax", out["ref_vp_logits"].argmax(1).numpy(), "logits[:4]", out["ref_vp_logits"][0, :4].numpy())

    geometry_golden()

    # ---- refiner: one step, 6 refs
    net = name2network["refiner"]({"network": "refiner", "name": "g"}).eval()
    net.load_state_dict(synth.synth_state_dict("refiner"))
    case = synth.refiner_case()
    with torch.no_grad():
        data = {"que_imgs_info": {"imgs": case["que_imgs"], "Ks_in": case["Ks_in"], "poses_in": case["poses_in"]},
                "ref_imgs_info": {"imgs": case["ref_imgs"], "Ks": case["ref_Ks"], "poses": case["ref_poses"]},
                "inference": True}
        out = net(data)
        mean, std, vin, _ = net.construct_feature_volume(data["que_imgs_info"], data["ref_imgs_info"], net.feature_net, 32)
        qf = net.feature_net(case["que_imgs"])
    sl = (slice(None), slice(0, 8), slice(None, None, 4), slice(None, None, 4), slice(None, None, 4))
    np.savez_compressed(os.path.join(HERE, "ref_step.npz"), **np_(out), sha_inputs=synth.fingerprint(case),
                        sha_weights=synth.fingerprint(synth.synth_state_dict("refiner")), vol_mean=mean[sl].numpy(), vol_std=std[sl].numpy(),
                        vol_in=vin[sl].numpy(), que_feats=qf[:, :8].numpy())
    print("ref_step", np_(out))


if __name__ == "__main__":
    main()
