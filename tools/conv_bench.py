#!/usr/bin/env python
"""Micro-benchmark of g6d_conv_igemm at the layer shapes of the Gen6D hot path (TFLOP/s per shape, HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import toolenv  # noqa: E402,F401  (G6D_LIB_PATH / KNOBS)
from gen6d_amd import ops  # noqa: E402

SHAPES = [
    # name, N, D, H, W, Cin, Cout, k, stride, pad, mode
    ("sel L0 conv1 (q*r, IN)", 320, 1, 16, 16, 512, 64, (1, 3, 3), 1, (0, 1, 1), "mul"),
    ("sel L0 conv2", 320, 1, 16, 16, 64, 64, (1, 3, 3), 1, (0, 1, 1), "aff"),
    ("sel L0 conv3 8x8", 320, 1, 8, 8, 64, 128, (1, 3, 3), 1, (0, 1, 1), ""),
    ("sel L0 conv5 4x4", 320, 1, 4, 4, 128, 256, (1, 3, 3), 1, (0, 1, 1), ""),
    ("sel L1 conv1", 320, 1, 8, 8, 512, 128, (1, 3, 3), 1, (0, 1, 1), "mul"),
    ("sel fuse 768->512", 320, 1, 4, 4, 768, 512, (1, 1, 1), 1, (0, 0, 0), ""),
    ("vol mean_embed.0", 1, 32, 32, 32, 256, 64, (3, 3, 3), 1, (1, 1, 1), ""),
    ("vol conv0 128->64", 1, 32, 32, 32, 128, 64, (3, 3, 3), 1, (1, 1, 1), "aff"),
    ("vol conv1 s2", 1, 32, 32, 32, 64, 128, (3, 3, 3), 2, (1, 1, 1), "aff"),
    ("vol conv2 16^3", 1, 16, 16, 16, 128, 128, (3, 3, 3), 1, (1, 1, 1), "aff"),
    ("vol conv3 s2", 1, 16, 16, 16, 128, 256, (3, 3, 3), 2, (1, 1, 1), "aff"),
    ("vol conv4 8^3", 1, 8, 8, 8, 256, 256, (3, 3, 3), 1, (1, 1, 1), "aff"),
    ("vol conv5.0 s2", 1, 8, 8, 8, 256, 512, (3, 3, 3), 2, (1, 1, 1), "aff"),
    ("vol conv5.3 4^3", 1, 4, 4, 4, 512, 512, (3, 3, 3), 1, (1, 1, 1), "aff"),
    ("det corr 15x15 s0.5", 1, 1, 88, 116, 512, 32, (1, 15, 15), 1, (0, 7, 7), ""),
    ("det corr 15x15 s0", 1, 1, 60, 80, 512, 32, (1, 15, 15), 1, (0, 7, 7), ""),
    ("det corr 7x7 s0.5", 1, 1, 44, 58, 512, 32, (1, 7, 7), 1, (0, 3, 3), ""),
    ("feat conv1.0 16x16", 7, 1, 16, 16, 512, 256, (1, 3, 3), 1, (0, 1, 1), "pn"),
    ("feat conv_out.0", 7, 1, 32, 32, 192, 128, (1, 3, 3), 1, (0, 1, 1), "pn"),
    ("feat conv0.0 32x32", 7, 1, 32, 32, 256, 64, (1, 3, 3), 1, (0, 1, 1), "pn"),
    ("feat conv0.1 32x32", 7, 1, 32, 32, 64, 64, (1, 3, 3), 1, (0, 1, 1), "affpn"),
    ("feat conv1.1 16x16", 7, 1, 16, 16, 256, 64, (1, 3, 3), 1, (0, 1, 1), "affpn"),
    ("feat conv2.0 8x8", 7, 1, 8, 8, 512, 256, (1, 3, 3), 1, (0, 1, 1), "pn"),
    ("feat conv2.1 8x8", 7, 1, 8, 8, 256, 64, (1, 3, 3), 1, (0, 1, 1), "affpn"),
    ("feat conv_out.1", 7, 1, 32, 32, 128, 128, (1, 3, 3), 1, (0, 1, 1), "affpn"),
]


def main():
    dev = torch.device("cuda")
    reps = int(os.environ.get("REPS", "10"))
    only = os.environ.get("ONLY")
    tot_f = tot_t = 0.0
    for name, N, D, H, W, Cin, Cout, k, s, p, mode in SHAPES:
        if only and not any(o in name for o in only.split(",")):
            continue
        Do, Ho, Wo = [(i + 2 * pp - kk) // s + 1 for i, kk, pp in zip((D, H, W), k, p)]
        x = torch.randn((N, D, H, W, Cin), device=dev)
        w = torch.randn((Cout, k[0] * k[1] * k[2], Cin), device=dev) * 0.01
        b = torch.randn((Cout,), device=dev)
        out = torch.empty((N, Do, Ho, Wo, Cout), device=dev)
        kw = {}
        pn = mode.endswith("pn")                       # per-image statistics (InstanceNorm2d) / per-image affine tables
        if mode == "affpn":
            kw.update(in_scale=torch.rand((N, Cin), device=dev) + 0.5, in_shift=torch.randn((N, Cin), device=dev), in_relu=True, per_n=True)
        if mode in ("aff", "mul"):
            kw.update(in_scale=torch.rand((1, Cin), device=dev) + 0.5, in_shift=torch.randn((1, Cin), device=dev), in_relu=mode == "aff")
        if mode == "mul":
            kw.update(mul=torch.randn((H, W, Cin), device=dev))
        stats = ops.new_stats(N if pn else 1, Cout, dev)
        if pn:
            kw["rows_per_group"] = Do * Ho * Wo
        kw["split_k"] = int(os.environ.get("SPLITK", "0"))
        if os.environ.get("WINO", "0") == "1" and k[1] == 3 and s == 1 and Cin % 8 == 0 and Cout % 32 == 0:
            # direct-form TFLOP/s are printed: a layer g6d_conv_igemm runs on the Winograd kernel executes 1/2.25 of those FLOPs
            from gen6d_amd.network.backbone import winograd_filters_taps
            kw["w_wino"] = winograd_filters_taps(w, k[0])
        for _ in range(2):
            ops.conv(x, w, b, out, ksize=k, stride=(s,) * 3, pad=p, stats=stats, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.conv(x, w, b, out, ksize=k, stride=(s,) * 3, pad=p, stats=stats, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if name.startswith("det corr"):
            xs, os_ = x[0:1, 0:1], out[0:1, 0:1]
            for _ in range(2): ops.corr2d_patch(xs, w, os_, k[1])
            torch.cuda.synchronize(); e0.record()
            for _ in range(reps): ops.corr2d_patch(xs, w, os_, k[1])
            e1.record(); torch.cuda.synchronize()
            ms2 = e0.elapsed_time(e1) / reps
            fl2 = 2.0 * N * Do * Ho * Wo * Cout * k[0] * k[1] * k[2] * Cin
            print(f"{'  -> corr2d_patch':28s} {'':30s} {ms2 * 1e3:8.1f} us  {fl2 / ms2 / 1e9:7.1f} TFLOP/s")
        fl = 2.0 * N * Do * Ho * Wo * Cout * k[0] * k[1] * k[2] * Cin
        tot_f += fl; tot_t += ms
        print(f"{name:28s} M={N * Do * Ho * Wo:6d} N={Cout:4d} K={k[0] * k[1] * k[2] * Cin:6d}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    print(f"{'total':28s} {tot_t * 1e3:8.1f} us  {tot_f / tot_t / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
