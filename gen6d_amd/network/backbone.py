"""VGG-11-BN trunk on PyTorch-ROCm (MIOpen convolutions), as BASELINE.json's north_star prescribes for the backbones.
BatchNorm (eval) is folded into the conv weights once.  Output taps follow the reference exactly, including its
quirk that the 1/16 output is the BatchNorm output of conv 25 WITHOUT the final ReLU
(reference network/pretrain_models.py:17-25,66-72,109-111)."""
import torch
import torch.nn.functional as F

from .. import specs

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
        if i in _POOL_BEFORE:
            x = F.max_pool2d(x, 2, 2)
        x = F.conv2d(x, w, b, padding=1)
        if i == 7:
            out["c7_pre"] = x
            if "p7" in taps:
                out["p7"] = F.max_pool2d(x, 2, 2)
            break
        x = F.relu(x)
        if i == 3: out["c3"] = x
        if i == 5: out["c5"] = x
    return out
