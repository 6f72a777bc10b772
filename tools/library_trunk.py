"""A/B helper (tools and tests only): the VGG trunk on PyTorch-ROCm / MIOpen convolutions (NCHW) with the small glue kernels
g6d_vgg_conv1_pool / g6d_bias_relu_pool_nchw / g6d_nchw_to_nhwc — what the networks ran on in rounds 1-2, which BASELINE.json's
north_star allows for the backbones.  The product trunk is gen6d_amd/network/backbone.py (own channels-last Winograd trunk);
`install()` swaps this one in for a measurement or a parity run, `uninstall()` restores the product functions."""
import torch
import torch.nn.functional as F

from gen6d_amd import ops, specs
from gen6d_amd.network import backbone, detector, refiner, selector

OWN_CONV1 = True          # first layer on g6d_vgg_conv1_pool (False: MIOpen too)
_POOL_BEFORE = (1, 2, 4, 6)
_NORM = {}


def img_norm(x):
    """torchvision.transforms.Normalize(ImageNet) on [n,3,h,w] in [0,1] (constants cached per device: capturable)."""
    key = (str(x.device), x.dtype)
    if key not in _NORM:
        _NORM[key] = (torch.tensor(specs.IMAGENET_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1),
                      torch.tensor(specs.IMAGENET_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1))
    m, s = _NORM[key]
    return (x - m) / s


def vgg_taps(folded, x, taps):
    """Folded trunk [(w, b)] x 8 on F.conv2d; `taps` among 'c3' (256ch @1/4, post-ReLU), 'c5' (512 @1/8, post-ReLU), 'c7_pre' (512 @1/16,
    pre-ReLU), 'p7' (max-pool of c7_pre)."""
    out = {}
    for i, (w, b) in enumerate(folded):
        if i == 0 and OWN_CONV1 and x.shape[1] == 3 and w.shape[0] == 64 and x.shape[2] >= 2 and x.shape[3] >= 2:
            x = ops.vgg_conv1_pool(x.contiguous(), w, b)
            continue
        y = F.conv2d(x, w, None, padding=1)
        if i == 7:
            out["c7_pre"] = ops.bias_relu_pool_nchw(y, b, False, False)       # BN output WITHOUT the last ReLU
            if "p7" in taps:
                out["p7"] = ops.bias_relu_pool_nchw(y, b, False, True)
            break
        pool_next = (i + 1) in _POOL_BEFORE
        tap = {3: "c3", 5: "c5"}.get(i)
        if tap in taps and pool_next:
            out[tap] = ops.bias_relu_pool_nchw(y, b, True, False)
            x = ops.bias_relu_pool_nchw(y, b, True, True)
        else:
            x = ops.bias_relu_pool_nchw(y, b, True, pool_next)
            if tap in taps: out[tap] = x
    return {k: v for k, v in out.items() if k in taps or k == "c7_pre"}


def trunk_features(folded, imgs, keys, l2norm, f43=False):
    t = vgg_taps(folded, img_norm(imgs), set(keys))
    outs = []
    for k in keys:
        f = t[k].contiguous()
        n, c, h, w = f.shape
        outs.append(ops.nchw_to_nhwc(f, torch.empty((n, 1, h, w, c), dtype=torch.float32, device=f.device), l2norm))
    return outs


def trunk_features_multi(folded, imgs_list, keys, f43=False, taps16=()):
    return [trunk_features(folded, im, keys, False) for im in imgs_list]


_SAVED = {}


def install():
    """Route the three networks' trunk calls to the library trunk (their modules bind the names at import)."""
    for mod in (backbone, detector, selector, refiner):
        for name, fn in (("pack_trunk", lambda folded: folded), ("trunk_features", trunk_features), ("trunk_features_multi", trunk_features_multi)):
            if hasattr(mod, name):
                _SAVED.setdefault((mod, name), getattr(mod, name))
                setattr(mod, name, fn)


def uninstall():
    for (mod, name), fn in _SAVED.items():
        setattr(mod, name, fn)
    _SAVED.clear()
