#!/usr/bin/env python
"""F(4x4,3x3) kernel against the F(2x2,3x3) kernel, layer by layer at the headline sizes (batch of 8 queries): the seven Winograd layers
of the detector's pyramid trunk, the 15x15 correlation, the refiner's 32^3 volume layers and the crops' trunk (56 images).
Usage (GPU box): python tools/w43_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import toolenv  # noqa: E402,F401  (G6D_LIB_PATH / KNOBS)
from gen6d_amd import lib, ops  # noqa: E402
from gen6d_amd.network import backbone as B  # noqa: E402

lib.load()
dev = torch.device("cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator().manual_seed(1)


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


def rnd(*shape, scale=1.0):
    return ((torch.rand(shape, generator=g) * 2 - 1) * scale).to(dev)


print("| layer | F(2,3) us | F(4,3) us | speed-up | F(4,3) executed TFLOP/s | max diff / range |")
print("|---|---|---|---|---|---|")
tot = [0.0, 0.0]


def trunk_layer(name, sizes, Cin, Cout, full, pool, count=True):
    w = rnd(Cout, Cin, 3, 3, scale=(2.0 / (9 * Cin)) ** 0.5)
    b = rnd(Cout, scale=0.1)
    xs = ops.alloc_like_segments([(n, h, ww, Cin) for n, h, ww in sizes], dev)
    for x in xs:
        x.copy_(torch.relu(rnd(*x.shape)))
    U2, U4 = B.winograd_filters(w), B.winograd43_filters(w)
    t2 = timeit(lambda: ops.wino_conv3x3_multi(xs, U2, b, relu=True, full=full, pool=pool))
    t4 = timeit(lambda: ops.wino43_conv3x3_multi(xs, U4, b, relu=True, full=full, pool=pool))
    a = ops.wino_conv3x3_multi(xs, U2, b, relu=True, full=full, pool=pool)
    c = ops.wino43_conv3x3_multi(xs, U4, b, relu=True, full=full, pool=pool)
    ya, yc = (a[0], c[0]) if full else (a[1], c[1])
    diff = max(((p_ - q_).abs().max() / (p_.max() - p_.min())).item() for p_, q_ in zip(ya, yc))
    fl = sum(2.0 * n * h * ww * Cout * 9 * Cin for n, h, ww in sizes)
    print(f"| {name} {'+'.join(f'{n}x{h}x{ww}' for n, h, ww in sizes)}x{Cin}->{Cout} | {t2:.1f} | {t4:.1f} | {t2 / t4:.2f} | {fl / 4 / t4 / 1e6:.1f} | {diff:.1e} |", flush=True)
    if count:
        tot[0] += t2; tot[1] += t4


Bq = 8
pyr = lambda d: [(Bq, 704 // d, 928 // d), (Bq, 480 // d, 640 // d), (Bq, 352 // d, 480 // d), (Bq, 256 // d, 320 // d)]
trunk_layer("pyramid", pyr(2), 64, 128, False, True)
trunk_layer("pyramid", pyr(4), 128, 256, True, False)
trunk_layer("pyramid", pyr(4), 256, 256, False, True)
trunk_layer("pyramid", pyr(8), 256, 512, True, False)
trunk_layer("pyramid", pyr(8), 512, 512, True, True)
trunk_layer("pyramid", pyr(16), 512, 512, True, False)
trunk_layer("pyramid", pyr(16), 512, 512, True, True)
print(f"| **pyramid trunk** | {tot[0]:.0f} | {tot[1]:.0f} | {tot[0] / tot[1]:.2f} | | |", flush=True)
tot = [0.0, 0.0]
for nm, hw, ci, co, full, pool in (("crops", 64, 64, 128, False, True), ("crops", 32, 128, 256, True, False), ("crops", 32, 256, 256, True, True),
                                   ("crops", 16, 256, 512, True, False), ("crops", 16, 512, 512, True, True), ("crops", 8, 512, 512, True, False)):
    trunk_layer(nm, [(56, hw, hw)], ci, co, full, pool)
print(f"| **crops trunk (56 images)** | {tot[0]:.0f} | {tot[1]:.0f} | {tot[0] / tot[1]:.2f} | | |", flush=True)

# 15x15 correlation
k, Cin, Cout = 15, 512, 32
w = rnd(Cout, k * k, Cin, scale=(1.0 / (k * k * Cin)) ** 0.5)
sizes = [(88, 116), (60, 80), (44, 60), (32, 40)]
xs = ops.alloc_like_segments([(Bq, 1, h, ww, Cin) for h, ww in sizes], dev)
for x in xs:
    x.copy_(rnd(*x.shape))
o2 = ops.alloc_like_segments([(Bq, 1, h, ww, Cout) for h, ww in sizes], dev)
o4 = ops.alloc_like_segments([(Bq, 1, h, ww, Cout) for h, ww in sizes], dev)
U2, U4 = B.winograd_corr_filters(w, k), B.winograd43_corr_filters(w, k)
t2 = timeit(lambda: ops.corr2d_wino_multi(xs, U2, o2, 5))
t4 = timeit(lambda: ops.corr2d_wino43_multi(xs, U4, o4, 5))
diff = max(((p_ - q_).abs().max() / (p_.max() - p_.min())).item() for p_, q_ in zip(o2, o4))
fl = sum(2.0 * Bq * h * ww * Cout * k * k * Cin for h, ww in sizes)
print(f"| corr 15x15 x{Bq} | {t2:.1f} | {t4:.1f} | {t2 / t4:.2f} | {fl / 4 / t4 / 1e6:.1f} | {diff:.1e} |", flush=True)

# volume layers
for Cin, Cout, aff in ((256, 64, False), (64, 64, True), (128, 64, False)):
    w = rnd(Cout, 27, Cin, scale=(1.0 / (27 * Cin)) ** 0.5)
    b = rnd(Cout, scale=0.1)
    x = rnd(Bq, 32, 32, 32, Cin)
    sc = (0.5 + torch.rand((Bq, Cin), generator=g)).to(dev) if aff else None
    sh = rnd(Bq, Cin, scale=0.3) if aff else None
    U2, U4 = B.winograd_filters_taps(w, 3), B.winograd43_filters_taps(w, 3)
    oa = torch.empty((Bq, 32, 32, 32, Cout), device=dev); ob = torch.empty_like(oa)

    def run(out, **kw):
        ops.stats_arena_begin(dev)
        st = ops.new_stats(Bq, Cout, dev)
        ops.conv(x, w, b, out, ksize=(3, 3, 3), pad=(1, 1, 1), in_scale=sc, in_shift=sh, in_relu=aff, per_n=1 if aff else 0, stats=st,
                 rows_per_group=32768, **kw)
    t2 = timeit(lambda: run(oa, w_wino=U2))
    t4 = timeit(lambda: run(ob, w_wino43=U4))
    diff = ((oa - ob).abs().max() / (oa.max() - oa.min())).item()
    fl = 2.0 * Bq * 32768 * Cout * 27 * Cin
    print(f"| volume 8x32^3x{Cin}->{Cout}{' aff' if aff else ''} | {t2:.1f} | {t4:.1f} | {t2 / t4:.2f} | {fl / 4 / t4 / 1e6:.1f} | {diff:.1e} |", flush=True)
