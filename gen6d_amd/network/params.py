"""ParamBank: an nn.Module whose state_dict() has exactly the reference's keys and shapes (gen6d_amd/specs.py), so
`load_state_dict(torch.load('data/model/<name>/model_best.pth')['network_state_dict'])` works unchanged
(reference estimator.py:117-125), plus helpers that repack weights into the layouts the HIP kernels consume."""
import torch
import torch.nn as nn

from .. import specs


class ConvW(tuple):
    """(weight [Cout,taps,Cin], bias) as the conv kernels take them, plus `.u`: the same filters transformed for the Winograd
    kernel (None when the layer never qualifies), and `.u43`: the same for the F(4x4,3x3) kernel (G6dConv.weight_wino43; only for the
    layers that ask for it).  Unpacks like the plain pair."""

    def __new__(cls, w, b, u=None, u43=None):
        t = super().__new__(cls, (w, b))
        t.u, t.u43 = u, u43
        return t


class ParamBank(nn.Module):
    def __init__(self, rows):
        super().__init__()
        self._roles = {}
        for key, shape, role in specs.expand(rows):
            *path, leaf = key.split(".")
            mod = self
            for name in path:
                if name not in mod._modules:
                    mod.add_module(name, nn.Module())
                mod = mod._modules[name]
            if role in ("weight", "bias", "gamma", "beta"):
                init = torch.ones(shape) if role == "gamma" else torch.zeros(shape)
                mod.register_parameter(leaf, nn.Parameter(init, requires_grad=False))
            elif role == "count":
                mod.register_buffer(leaf, torch.zeros((), dtype=torch.long))
            else:
                mod.register_buffer(leaf, torch.ones(shape) if role == "rvar" else torch.zeros(shape))
            self._roles[key] = role
        self._packed = None

    # any weight change invalidates the packed copies
    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def p(self, key):
        *path, leaf = key.split(".")
        mod = self
        for name in path:
            mod = mod._modules[name]
        t = mod._parameters.get(leaf)
        return (t if t is not None else mod._buffers[leaf]).detach()

    def device_(self):
        return next(self.parameters()).device

    def conv_w(self, prefix, cin_pad=None, wino_kd=0, f43=False):
        """[Cout,Cin,*k] -> ConvW([Cout,taps,Cin] contiguous (Cin zero-padded to cin_pad), bias).  wino_kd = 1 / 3: the layer is
        a stride-1 (1,3,3) / (3,3,3) convolution — also keep its Winograd-domain filters (`.u`, see G6dConv.weight_wino); f43: and those of
        the F(4x4,3x3) kernel (`.u43`, Cout % 64 == 0)."""
        w = self.p(prefix + ".weight")
        co, ci = w.shape[:2]
        w = w.reshape(co, ci, -1).permute(0, 2, 1)
        if cin_pad is not None and cin_pad != ci:
            w = torch.nn.functional.pad(w, (0, cin_pad - ci))
        w = w.contiguous()
        u = u43 = None
        if wino_kd and ci % 8 == 0 and co % 32 == 0 and w.shape[1] == 9 * wino_kd:
            from .backbone import winograd_filters_taps, winograd43_filters_taps
            u = winograd_filters_taps(w, wino_kd)
            if f43 and co % 64 == 0:
                u43 = winograd43_filters_taps(w, wino_kd)
        return ConvW(w, self.p(prefix + ".bias").contiguous(), u, u43)


def fold_vgg(bank, prefix):
    """Fold eval-mode BatchNorm into the preceding conv: returns [(w, b)] for the 8 VGG-11 convs
    (reference pretrain_models.py:86-104; BN eps 1e-5, running statistics)."""
    out = []
    for i in specs.VGG11_BN_CONVS:
        w, b = bank.p(f"{prefix}.{i}.weight"), bank.p(f"{prefix}.{i}.bias")
        g, beta = bank.p(f"{prefix}.{i + 1}.weight"), bank.p(f"{prefix}.{i + 1}.bias")
        mu, var = bank.p(f"{prefix}.{i + 1}.running_mean"), bank.p(f"{prefix}.{i + 1}.running_var")
        s = g / torch.sqrt(var + 1e-5)
        out.append(((w * s.view(-1, 1, 1, 1)).contiguous(), ((b - mu) * s + beta).contiguous()))
    return out
