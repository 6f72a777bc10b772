"""Debug aid: error of g6d_corr16_multi on a few shapes.  python tools/ubench/corr16_debug.py"""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools")); import toolenv
from gen6d_amd import lib, ops
lib.load()
def run(N, sizes, Cin, k, mode=2):
    g = torch.Generator().manual_seed(3)
    T = k * k
    w = ((torch.rand((32, T, Cin), generator=g) * 2 - 1) * (3.0 / (T * Cin)) ** 0.5).half().float()
    xs = [(torch.rand((N, h, ww, Cin), generator=g) * 2 - 1).half().float() for h, ww in sizes]
    filt = ops.corr16_pack(w.cuda(), mode)
    outs = [torch.zeros((N, 1, h, ww, 32), device="cuda") for h, ww in sizes]
    ops.corr16_multi([x.half().cuda() for x in xs], filt, outs)
    torch.cuda.synchronize()
    for x, o in zip(xs, outs):
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().reshape(32, k, k, Cin).permute(0, 3, 1, 2), None, padding=k // 2).permute(0, 2, 3, 1)
        e = (o[:, 0].cpu().double() - ref).abs() / float(ref.abs().max())
        bad = e > 1e-4
        print(f"N{N} {tuple(x.shape[1:3])} Cin {Cin} k {k}: max err {float(e.max()):.3g}, bad fraction {float(bad.float().mean()):.4f}")
        if bad.any():
            pm = bad[0].float().mean(-1)
            H, W = pm.shape
            for y in range(0, H, max(1, H // 24)):
                print("   ", "".join("#" if pm[y, x] > 0.5 else ("+" if pm[y, x] > 0 else ".") for x in range(0, W, max(1, W // 60))))
for args in ((1, [(16, 16)], 512, 15), (1, [(88, 116)], 64, 15), (1, [(32, 40)], 512, 15), (1, [(88, 116)], 512, 15), (1, [(88, 116), (60, 80), (44, 60), (32, 40)], 256, 15)):
    run(*args)
