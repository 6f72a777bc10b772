#!/usr/bin/env python
"""Where a wave of the F(4x4,3x3) kernel spends its cycles (profiling build of tools/w43_timing.sh): shader-clock intervals between the
phase boundaries of the chunk loop, averaged per chunk over all waves of a launch.  Usage on the GPU box:
G6D_LIB_PATH=$PWD/gen6d_amd/csrc/_abl/libgen6d_t.so python tools/w43_timing.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import toolenv  # noqa: E402,F401
from gen6d_amd import lib, ops  # noqa: E402
from gen6d_amd.network import backbone as B  # noqa: E402

l = lib.load()
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
rnd = lambda *s, scale=1.0: ((torch.rand(s, generator=g) * 2 - 1) * scale).to(dev)
Bq = 8
pyr = lambda d: [(Bq, 704 // d, 928 // d), (Bq, 480 // d, 640 // d), (Bq, 352 // d, 480 // d), (Bq, 256 // d, 320 // d)]
SLOTS = ("prologue", "raw reads + transform", "phase X", "barrier after X", "phase Y", "barrier after Y", "epilogue")
print("cycles per chunk and wave (shader clock; 144 MFMAs of a chunk = 4608 matrix-pipe cycles); prologue / epilogue per block\n")
print("| launch | us | chunks per block | " + " | ".join(SLOTS) + " | chunk total |")
print("|---|---|---|" + "---|" * (len(SLOTS) + 1))
for name, sizes, Cin, Cout, full, pool in (("pyr/2 64->128", pyr(2), 64, 128, False, True), ("pyr/4 256->256", pyr(4), 256, 256, False, True),
                                           ("pyr/8 512->512", pyr(8), 512, 512, True, True), ("pyr/16 512->512", pyr(16), 512, 512, True, False),
                                           ("crops16 512->512", [(56, 16, 16)], 512, 512, True, True)):
    w = rnd(Cout, Cin, 3, 3, scale=(2.0 / (9 * Cin)) ** 0.5)
    b = rnd(Cout, scale=0.1)
    xs = ops.alloc_like_segments([(n, h, ww, Cin) for n, h, ww in sizes], dev)
    for x in xs:
        x.copy_(torch.relu(rnd(*x.shape)))
    U4 = B.winograd43_filters(w)
    lib.set_knob("w43_split_max", 1)                  # un-split launches: every block runs the whole reduction and reaches the final stamp
    fn = lambda: ops.wino43_conv3x3_multi(xs, U4, b, relu=True, full=full, pool=pool)
    fn(); fn()
    buf = (C.c_ulonglong * 32)()
    l.g6d_w43_timing_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    l.g6d_w43_timing_read(buf)
    t = [[buf[w_ * 8 + i] for i in range(8)] for w_ in range(4)]
    chunks = sum(r[7] for r in t)
    nblk = chunks / 4 / (Cin // 8)
    per = [sum(r[i] for r in t) / chunks for i in range(1, 6)]
    pro, epi = sum(r[0] for r in t) / (4 * nblk), sum(r[6] for r in t) / (4 * nblk)
    print(f"| {name} | {e0.elapsed_time(e1) * 1e3:.0f} | {Cin // 8} | {pro:.0f} | " + " | ".join(f"{v:.0f}" for v in per) + f" | {epi:.0f} | {sum(per):.0f} |")
