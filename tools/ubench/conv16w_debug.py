"""Debug aid: error map of the halo-patch kernel (conv16w) on a small case.  python tools/ubench/conv16w_debug.py"""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gen6d_amd import lib, ops
lib.load()
def run(N, H, W, Cin, Cout, onehot=None):
    g = torch.Generator().manual_seed(3)
    w = ((torch.rand((Cout, 9, Cin), generator=g) * 2 - 1) * 0.1).half().float()
    x = (torch.rand((N, H, W, Cin), generator=g) * 2 - 1).half().float()
    if onehot == "tap":      # only the centre tap
        w[:, [0, 1, 2, 3, 5, 6, 7, 8], :] = 0
    if onehot == "ch":       # only input channels < 16
        w[:, :, 16:] = 0
    if isinstance(onehot, tuple):   # (tap or None, channel range)
        tp, c0, c1 = onehot
        m = torch.zeros_like(w)
        m[:, (slice(None) if tp is None else [tp]), c0:c1] = 1
        w = w * m
    filt = ops.conv16_pack(w.cuda(), 2, 1)
    fulls, _ = ops.conv16_direct_multi([x.half().cuda()], filt, torch.zeros(Cout).cuda(), relu=False, full=torch.float32, pool=None)
    torch.cuda.synchronize()
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), None, padding=1).permute(0, 2, 3, 1)
    e = (fulls[0].cpu().double() - ref).abs()
    print(f"--- N{N} {H}x{W} {Cin}->{Cout} {onehot}: max err {float(e.max()):.3g} of range {float(ref.abs().max()):.3g}")
    bad = e > 1e-3 * float(ref.abs().max())
    print("  bad fraction", float(bad.float().mean()))
    print("  bad by channel block of 32:", [round(float(bad[..., i:i + 32].float().mean()), 3) for i in range(0, Cout, 32)])
    print("  bad by image:", [round(float(bad[n].float().mean()), 3) for n in range(N)])
    pm = bad[0].float().mean(-1)
    for y in range(H):
        print("  ", "".join("#" if v > 0.5 else ("+" if v > 0 else ".") for v in pm[y]))
for oh in ((None, 16, 32), (None, 32, 48), (None, 48, 64), (0, 0, 64), (1, 0, 64), (3, 0, 64), (8, 0, 64), (0, 16, 32), (0, 32, 48), (5, 16, 32)):
    run(1, 16, 8, 64, 128, oh)
