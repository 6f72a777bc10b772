"""Deterministic synthetic weights and inputs.

No pretrained checkpoint or dataset exists offline, so parity tests, golden vectors and bench.py all draw their
weights from `synth_state_dict` (per-key seeded normal/uniform draws, so the same numbers appear in this
container, in the golden-vector script that feeds the reference modules, and on the GPU box) and their images /
cameras from the generators below.  Everything is produced with CPU `torch.Generator`s so results do not depend
on the device.
"""
import math
import zlib

import numpy as np
import torch

from . import specs


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def synth_state_dict(kind, seed=1234, an=5):
    """state_dict with the reference's keys/shapes for kind in {'detector','selector','refiner'}."""
    rows = specs.ROWS[kind](an) if kind == "selector" else specs.ROWS[kind]()
    sd = {}
    for key, shape, role in specs.expand(rows):
        g = _gen(kind + "/" + key, seed)
        if role == "weight":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif role == "bias":
            t = (torch.rand(shape, generator=g) - 0.5) * 0.1
        elif role in ("gamma", "rvar"):
            t = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif role in ("beta", "rmean"):
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        elif role == "count":
            t = torch.zeros((), dtype=torch.long)
        else:
            raise ValueError(role)
        sd[key] = t
    return sd


def damp_refiner_head(sd, gain=1e-3):
    """Copy of a refiner state_dict whose pose heads (regressor.fcr / fct / fcs, reference refiner.py:157-166) are scaled by `gain`
    and biased to the identity update (quaternion (1,0,0,0), zero offset, zero log-scale): a TRAINED refiner predicts small
    residuals around the identity, whereas the seeded random head produces O(1) residuals and amplifies single grey levels of the
    crops a hundredfold.  Estimator-level tests use it so that pose tolerances can be 1e-4 instead of 3e-2."""
    out = {k: v.clone() for k, v in sd.items()}
    for head, bias in (("fcr", [1.0, 0.0, 0.0, 0.0]), ("fct", [0.0, 0.0]), ("fcs", [0.0])):
        out[f"regressor.{head}.weight"] = sd[f"regressor.{head}.weight"] * gain
        out[f"regressor.{head}.bias"] = torch.tensor(bias, dtype=sd[f"regressor.{head}.bias"].dtype)
    return out


def synth_images(n, h, w, seed, structured=True):
    """uint8 [n,h,w,3]. structured: smooth low-frequency pattern + noise (avoids degenerate ties)."""
    g = _gen(f"img/{n}/{h}/{w}", seed)
    if structured:
        low = torch.rand((n, 3, max(h // 16, 2), max(w // 16, 2)), generator=g)
        img = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)
        img = 0.75 * img + 0.25 * torch.rand((n, 3, h, w), generator=g)
    else:
        img = torch.rand((n, 3, h, w), generator=g)
    return (img.permute(0, 2, 3, 1) * 255).round().clamp(0, 255).to(torch.uint8).contiguous().numpy()


def fibonacci_cameras(n, radius=5.0, focal=300.0, size=128):
    """n look-at-origin cameras on a Fibonacci sphere. Returns poses [n,3,4] (x_cam = R x + t), Ks [n,3,3]."""
    poses = np.zeros((n, 3, 4), np.float32)
    ga = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(n):
        z = 1.0 - 2.0 * (i + 0.5) / n
        r = math.sqrt(max(0.0, 1.0 - z * z))
        c = np.array([r * math.cos(ga * i), r * math.sin(ga * i), z]) * radius
        fwd = -c / np.linalg.norm(c)                      # camera z axis looks at the origin
        up = np.array([0.0, 0.0, 1.0]) if abs(fwd[2]) < 0.99 else np.array([0.0, 1.0, 0.0])
        right = np.cross(up, fwd); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], 0)
        poses[i, :, :3] = R
        poses[i, :, 3] = -R @ c
    K = np.array([[focal, 0, size / 2], [0, focal, size / 2], [0, 0, 1]], np.float32)
    return poses, np.repeat(K[None], n, 0)


def perturb_pose(pose, rot_deg=5.0, trans_frac=0.03):
    """Rotate about the camera y axis by rot_deg and scale the translation (refiner input pose, §8(d))."""
    a = math.radians(rot_deg)
    Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], np.float32)
    out = pose.copy()
    out[:, :3] = Ry @ pose[:, :3]
    out[:, 3] = (Ry @ pose[:, 3]) * (1.0 + trans_frac)
    return out.astype(np.float32)


def rotated_copies(imgs, an):
    """[rfn,h,w,3] uint8 -> [an,rfn,h,w,3]: in-plane rotations about the image centre (bilinear, zero fill).

    Angles follow the reference for an=5 (estimator.py:152: -pi/2..pi/2) and are spread over (-pi,pi] otherwise.
    Bit-reproducible on any host by construction (round 3): source coordinates are formed with element-wise float64
    numpy operations (no BLAS, no vectorised transcendental), quantised to 1/256 pixel, and the four taps are blended in
    integer arithmetic.  Round 2 went through affine_grid (an sgemm) + grid_sample + round(), whose uint8 results differed
    between the build container's and the GPU box's CPU - the unexplained 7.7e-4 of r02_parity.md."""
    rfn, h, w, _ = imgs.shape
    angles = np.linspace(-np.pi / 2, np.pi / 2, an) if an == 5 else np.linspace(-np.pi, np.pi, an, endpoint=False)
    src = np.zeros((rfn, h + 2, w + 2, 3), np.int64)                    # one-pixel zero border = the zero fill
    src[:, 1:-1, 1:-1] = imgs
    X = ((2.0 * np.arange(w, dtype=np.float64) + 1.0) / w - 1.0)[None, :]
    Y = ((2.0 * np.arange(h, dtype=np.float64) + 1.0) / h - 1.0)[:, None]
    out = np.empty((an, rfn, h, w, 3), np.uint8)
    for k, a in enumerate(angles):
        c, s = math.cos(float(a)), math.sin(float(a))
        px = ((c * X - s * Y + 1.0) * w - 1.0) * 0.5                     # source pixel coordinates (align_corners=False)
        py = ((s * X + c * Y + 1.0) * h - 1.0) * 0.5
        fx = np.floor(px * 256.0 + 0.5).astype(np.int64)
        fy = np.floor(py * 256.0 + 0.5).astype(np.int64)
        x0, wx = fx >> 8, (fx & 255)[None, :, :, None]
        y0, wy = fy >> 8, (fy & 255)[None, :, :, None]
        x0c, x1c = np.clip(x0 + 1, 0, w + 1), np.clip(x0 + 2, 0, w + 1)  # +1: border offset; fully outside -> border zeros
        y0c, y1c = np.clip(y0 + 1, 0, h + 1), np.clip(y0 + 2, 0, h + 1)
        inside = ((x0 >= -1) & (x0 <= w - 1) & (y0 >= -1) & (y0 <= h - 1))[None, :, :, None]
        v = (src[:, y0c, x0c] * (256 - wx) * (256 - wy) + src[:, y0c, x1c] * wx * (256 - wy) +
             src[:, y1c, x0c] * (256 - wx) * wy + src[:, y1c, x1c] * wx * wy + 32768) >> 16
        out[k] = np.where(inside, v, 0).astype(np.uint8)
    return out


def fingerprint(*objs):
    """SHA-256 over the raw bytes of tensors / arrays / (nested) dicts of them, keys in sorted order: how the golden fixtures
    pin the synthetic inputs and weights they were generated from (tests assert the same hash on the GPU box)."""
    import hashlib
    hsh = hashlib.sha256()

    def feed(o):
        if isinstance(o, dict):
            for k in sorted(o):
                hsh.update(str(k).encode()); feed(o[k])
        elif isinstance(o, (list, tuple)):
            for v in o: feed(v)
        else:
            a = o.detach().cpu().numpy() if torch.is_tensor(o) else np.asarray(o)
            hsh.update(str(a.dtype).encode()); hsh.update(str(a.shape).encode()); hsh.update(np.ascontiguousarray(a).tobytes())
    feed(list(objs))
    return hsh.hexdigest()


def imgs_to_tensor(imgs_u8):
    """uint8 [...,h,w,3] -> float32 [...,3,h,w] in [0,1] (reference utils/base_utils.py:117-118 + permute)."""
    t = torch.from_numpy(np.ascontiguousarray(imgs_u8)).float() / 255
    return t.movedim(-1, -3).contiguous()


def selector_case(rfn, an, seed=1):
    """Synthetic selector workload: ref images [an,rfn,3,128,128], poses, centre, vert, query [1,3,128,128]."""
    refs = synth_images(rfn, 128, 128, seed)
    rots = rotated_copies(refs, an)
    poses, _ = fibonacci_cameras(rfn)
    que = synth_images(1, 128, 128, seed + 100)
    return {
        "ref_imgs": imgs_to_tensor(rots), "ref_poses": torch.from_numpy(poses),
        "object_center": torch.zeros(3), "object_vert": torch.tensor([0.0, 0.0, 1.0]),
        "que_imgs": imgs_to_tensor(que),
    }


def detector_case(rfn, hq, wq, seed=2):
    refs = synth_images(rfn, 128, 128, seed)
    que = synth_images(1, hq, wq, seed + 100)
    return {"ref_imgs": imgs_to_tensor(refs), "que_imgs": imgs_to_tensor(que)}


def refiner_case(rfn=6, seed=3, size=128):
    """One refiner step: 6 reference crops with poses/Ks around view 0 and a perturbed input pose."""
    poses, Ks = fibonacci_cameras(64, radius=3.0, focal=180.0, size=size)
    # the rfn views closest to view 0 (by camera direction)
    dirs = np.stack([-(p[:, :3].T @ p[:, 3]) for p in poses])
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    idx = np.argsort(-(dirs @ dirs[0]))[:rfn]
    refs = synth_images(rfn, size, size, seed)
    que = synth_images(1, size, size, seed + 100)
    return {
        "que_imgs": imgs_to_tensor(que), "Ks_in": torch.from_numpy(Ks[:1].copy()),
        "poses_in": torch.from_numpy(perturb_pose(poses[0])[None]),
        "ref_imgs": imgs_to_tensor(refs)[None], "ref_Ks": torch.from_numpy(Ks[idx].copy())[None],
        "ref_poses": torch.from_numpy(poses[idx].copy())[None],
    }
