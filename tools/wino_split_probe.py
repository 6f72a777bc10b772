#!/usr/bin/env python
"""Own-trunk layer times at the detector / crop map sizes (one line per layer); run with KNOBS="wino_split_max=..,wino_split_gain=.." (tools/toolenv.py) to
compare split choices.  KNOBS=wino_debug=1 prints the chosen split of every launch to stderr."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import toolenv  # noqa: E402,F401  (G6D_LIB_PATH / KNOBS)
from gen6d_amd import ops  # noqa: E402
from gen6d_amd.network import backbone as B  # noqa: E402
from trunk_bench import LAYERS, timeit  # noqa: E402


def main():
    dev = torch.device("cuda")
    reps = int(os.environ.get("REPS", "20"))
    tot = 0.0
    sizes = [(1, 704, 928), (1, 480, 640), (1, 352, 480), (1, 256, 320), (7, 128, 128), (1, 128, 128)]
    if os.environ.get("SIZES") == "big":
        sizes = [(64, 128, 128)]            # full grids: the kernel's own rate, no under-fill
    for n, h, w in sizes:
        for cin, cout, ds, pool in LAYERS:
            hh, ww = h // ds, w // ds
            x = torch.randn((n, hh, ww, cin), device=dev)
            U = B.winograd_filters(torch.randn((cout, cin, 3, 3), device=dev) * 0.05)
            b = torch.randn((cout,), device=dev) * 0.1
            if os.environ.get("LOWP"):            # LOWP=fp16 / bf16: the 16-bit trunk kernel
                dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[os.environ["LOWP"]]
                U16 = B.winograd_filters16(torch.randn((cout, cin, 3, 3), device=dev) * 0.05, dt)
                with ops.math_mode(os.environ["LOWP"]):
                    t = timeit(lambda: ops.wino16_conv3x3_multi([x], U16, b, relu=True, full=not pool, pool=pool), reps)
            else:
                t = timeit(lambda: ops.wino_conv3x3(x, U, b, relu=True, full=not pool, pool=pool), reps)
            tot += t
            print(f"{n}x{hh}x{ww} {cin}->{cout}{' pool' if pool else ''}: {t:.1f}")
    print(f"total {tot:.1f}")


if __name__ == "__main__":
    main()
