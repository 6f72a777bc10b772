"""Halo-patch kernel: time against the number of input channels on one map shape — the slope is the cost of a 32-channel slice, the
intercept the fixed cost per block (prologue, epilogue).  python tools/ubench/conv16w_scaling.py [N H W Cout]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import toolenv  # noqa
from gen6d_amd import lib, ops
lib.load()
N, H, W, Cout = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (112, 32, 32, 256)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(1)
tiles = N * H * W / 128
print(f"# {N}x{H}x{W} -> {Cout}: {tiles:.0f} pixel tiles, {tiles * Cout / (256 if Cout % 256 == 0 else 128) / (1 if Cout % 256 == 0 else 2):.0f} blocks")
print("| Cin | fp16 us | TFLOP/s | us per slice and round | pairs us | TFLOP/s (direct) |\n|---|---|---|---|---|---|")
prev = None
for Cin in (64, 128, 256, 512, 1024, 2048):
    w = (torch.rand((Cout, 9, Cin), generator=g) * 2 - 1) * (3.0 / (9 * Cin)) ** 0.5
    x = torch.rand((N, H, W, Cin), generator=g) * 2 - 1
    bias = torch.zeros(Cout).cuda()
    f16 = ops.conv16_pack(w.cuda(), 2, 1)
    fp = ops.conv16_pack(w.cuda(), 3, 1)
    x16 = x.half().cuda()
    hi = x.half(); lo = (x - hi.float()).half()
    xp = torch.stack([hi, lo], -2).contiguous().cuda()
    t16 = timed(lambda: ops.conv16_direct_multi([x16], f16, bias, relu=True, full="t16", pool=None))
    tp = timed(lambda: ops.conv16_direct_multi([xp], fp, bias, relu=True, full="t16", pool=None))
    fl = 2.0 * N * H * W * Cout * 9 * Cin
    slope = "" if prev is None else f"{(t16 - prev[1]) / ((Cin - prev[0]) / 32):.2f}"
    print(f"| {Cin} | {t16:.0f} | {fl / t16 / 1e6:.0f} | {slope} | {tp:.0f} | {fl / tp / 1e6:.0f} |")
    prev = (Cin, t16)
