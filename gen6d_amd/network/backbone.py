"""VGG-11-BN trunk on PyTorch-ROCm (MIOpen convolutions), as BASELINE.json's north_star prescribes for the backbones.
BatchNorm (eval) is folded into the conv weights once.  Output taps follow the reference exactly, including its
quirk that the 1/16 output is the BatchNorm output of conv 25 WITHOUT the final ReLU
(reference network/pretrain_models.py:17-25,66-72,109-111)."""
import os

import torch
import torch.nn.functional as F

from .. import ops, specs

# The 3 -> 64 layer in front of the first pool runs on the fused HIP kernel g6d_vgg_conv1_pool (G6D_OWN_CONV1=0: MIOpen).
_OWN_CONV1 = os.environ.get("G6D_OWN_CONV1", "1") != "0"

if os.environ.get("G6D_MIOPEN_FIND", "0") == "1":
    torch.backends.cudnn.benchmark = True        # MIOpen Find (measured solver choice) instead of the immediate-mode heuristic

_POOL_BEFORE = (1, 2, 4, 6)          # positions (in the list of 8 convs) preceded by a 2x2 max-pool


_NORM = {}


def img_norm(x):
    """torchvision.transforms.Normalize(ImageNet) on [n,3,h,w] in [0,1]. Constants are cached per device so that the
    call is capturable in a hipGraph (no host-to-device copy on the query path)."""
    key = (str(x.device), x.dtype)
    if key not in _NORM:
        _NORM[key] = (torch.tensor(specs.IMAGENET_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1),
                      torch.tensor(specs.IMAGENET_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1))
    m, s = _NORM[key]
    return (x - m) / s


def vgg_taps(folded, x, taps):
    """Run the folded trunk; `taps` is a set of names among
       'c3' (256ch @1/4, post-ReLU), 'c5' (512 @1/8, post-ReLU), 'c7_pre' (512 @1/16, pre-ReLU), 'p7' (max-pool of c7_pre)."""
    out = {}
    for i, (w, b) in enumerate(folded):
        if i == 0 and _OWN_CONV1 and x.shape[1] == 3 and w.shape[0] == 64 and x.shape[2] >= 2 and x.shape[3] >= 2:
            x = ops.vgg_conv1_pool(x.contiguous(), w, b)                      # conv + bias + ReLU + pool in one kernel
            continue
        y = F.conv2d(x, w, None, padding=1)                                  # MIOpen; bias/ReLU/pool fused below
        if i == 7:
            out["c7_pre"] = ops.bias_relu_pool_nchw(y, b, False, False)       # BN output WITHOUT the last ReLU
            if "p7" in taps:
                out["p7"] = ops.bias_relu_pool_nchw(y, b, False, True)
            break
        pool_next = (i + 1) in _POOL_BEFORE
        tap = {3: "c3", 5: "c5"}.get(i)
        if tap in taps and pool_next:                                         # tapped feature is the un-pooled one
            out[tap] = ops.bias_relu_pool_nchw(y, b, True, False)
            x = ops.bias_relu_pool_nchw(y, b, True, True)
        else:
            x = ops.bias_relu_pool_nchw(y, b, True, pool_next)
            if tap in taps: out[tap] = x
    return {k: v for k, v in out.items() if k in taps or k == "c7_pre"}
