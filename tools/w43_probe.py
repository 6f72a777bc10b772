#!/usr/bin/env python
"""Times of the F(4x4,3x3) kernel alone on four representative launches (tools/w43_ablate_run.sh runs it per ablated build)."""
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
g = torch.Generator().manual_seed(1)
rnd = lambda *s, scale=1.0: ((torch.rand(s, generator=g) * 2 - 1) * scale).to(dev)
Bq = 8
pyr = lambda d: [(Bq, 704 // d, 928 // d), (Bq, 480 // d, 640 // d), (Bq, 352 // d, 480 // d), (Bq, 256 // d, 320 // d)]
out = []
for name, sizes, Cin, Cout, full, pool in (("pyr/2 64->128", pyr(2), 64, 128, False, True), ("pyr/8 512->512", pyr(8), 512, 512, True, True),
                                           ("pyr/16 512->512", pyr(16), 512, 512, True, False), ("crops16 512->512", [(56, 16, 16)], 512, 512, True, True)):
    w = rnd(Cout, Cin, 3, 3, scale=(2.0 / (9 * Cin)) ** 0.5)
    b = rnd(Cout, scale=0.1)
    xs = ops.alloc_like_segments([(n, h, ww, Cin) for n, h, ww in sizes], dev)
    for x in xs:
        x.copy_(torch.relu(rnd(*x.shape)))
    U4 = B.winograd43_filters(w)
    fn = lambda: ops.wino43_conv3x3_multi(xs, U4, b, relu=True, full=full, pool=pool)
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(int(os.environ.get("REPS", 5))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    out.append(f"{name} {min(ts):.0f}")
print(" | ".join(out))
