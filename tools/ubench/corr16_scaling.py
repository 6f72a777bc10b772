"""corr16_kernel: rate against the size of the filter set (Cin) on the headline's pyramid — does re-streaming the filters per tile bound it?
python tools/ubench/corr16_scaling.py [batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import toolenv  # noqa
from gen6d_amd import lib, ops
lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sizes = [(88, 116), (60, 80), (44, 60), (32, 40)]
def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(1)
print("| Cin | filters MB (fp16 / pairs) | fp16 us | TFLOP/s | pairs us | TFLOP/s direct (executed) |\n|---|---|---|---|---|---|")
for Cin in (64, 128, 256, 512):
    w = (torch.rand((32, 225, Cin), generator=g) * 2 - 1) * (3.0 / (225 * Cin)) ** 0.5
    xs = [torch.rand((B, h, ww, Cin), generator=g) * 2 - 1 for h, ww in sizes]
    outs = [torch.empty((B, 1, h, ww, 32), device="cuda") for h, ww in sizes]
    f2, f3 = ops.corr16_pack(w.cuda(), 2), ops.corr16_pack(w.cuda(), 3)
    x16 = [x.half().cuda() for x in xs]
    xp = [torch.stack([x.half(), (x - x.half().float()).half()], -2).contiguous().cuda() for x in xs]
    t2 = timed(lambda: ops.corr16_multi(x16, f2, outs)); t3 = timed(lambda: ops.corr16_multi(xp, f3, outs))
    fl = sum(2.0 * B * h * ww * 32 * 225 * Cin for h, ww in sizes)
    print(f"| {Cin} | {32 * 225 * Cin * 2 / 1e6:.1f} / {32 * 225 * Cin * 4 / 1e6:.1f} | {t2:.0f} | {fl / t2 / 1e6:.0f} | {t3:.0f} | {fl / t3 / 1e6:.0f} ({3 * fl / t3 / 1e6:.0f}) |")
