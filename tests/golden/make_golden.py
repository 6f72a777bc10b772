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
                            angles=out["angles_pr"].numpy(), pose_embed=embed.numpy()[:, :16])
        print(tag, "argmax", out["ref_vp_logits"].argmax(1).numpy(), "logits[:4]", out["ref_vp_logits"][0, :4].numpy())

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
    np.savez_compressed(os.path.join(HERE, "ref_step.npz"), **np_(out), vol_mean=mean[sl].numpy(), vol_std=std[sl].numpy(),
                        vol_in=vin[sl].numpy(), que_feats=qf[:, :8].numpy())
    print("ref_step", np_(out))


if __name__ == "__main__":
    main()
