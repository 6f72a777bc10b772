#!/usr/bin/env python
"""Per-layer timing of the VGG trunk: own Winograd/MFMA kernel (g6d_wino_conv3x3, channels-last, bias/ReLU/pool fused)
against the library path it replaces (F.conv2d on MIOpen + g6d_bias_relu_pool_nchw), at the map sizes of the detector
scales and of the 128x128 crops, plus the whole trunk both ways.  HIP events, REPS repetitions, random data."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import toolenv  # noqa: E402,F401  (G6D_LIB_PATH / KNOBS)
import library_trunk as LT  # noqa: E402
from gen6d_amd import ops, synth  # noqa: E402
from gen6d_amd.network import backbone as B  # noqa: E402

LAYERS = [(64, 128, 2, True), (128, 256, 4, False), (256, 256, 4, True), (256, 512, 8, False), (512, 512, 8, True),
          (512, 512, 16, False), (512, 512, 16, True)]      # Cin, Cout, downscale of the input map, pool after


def timeit(fn, reps):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def main():
    dev = torch.device("cuda")
    reps = int(os.environ.get("REPS", "10"))
    sizes = [(1, 704, 928), (1, 480, 640), (1, 352, 480), (1, 256, 320), (7, 128, 128), (1, 128, 128), (64, 128, 128)]
    print("| images | layer | map | own us | own TF/s eff (actual) | lib conv us | lib conv+glue us | speed-up |")
    print("|---|---|---|---|---|---|---|---|")
    for n, h, w in sizes:
        tot_own = tot_lib = 0.0
        for cin, cout, ds, pool in LAYERS:
            hh, ww = h // ds, w // ds
            x_cl = torch.randn((n, hh, ww, cin), device=dev)
            wt = torch.randn((cout, cin, 3, 3), device=dev) * (2.0 / (9 * cin)) ** 0.5
            b = torch.randn((cout,), device=dev) * 0.1
            U = B.winograd_filters(wt)
            x_nchw = x_cl.permute(0, 3, 1, 2).contiguous()
            t_own = timeit(lambda: ops.wino_conv3x3(x_cl, U, b, relu=True, full=not pool, pool=pool), reps)
            t_conv = timeit(lambda: F.conv2d(x_nchw, wt, None, padding=1), reps)
            t_lib = timeit(lambda: ops.bias_relu_pool_nchw(F.conv2d(x_nchw, wt, None, padding=1), b, True, pool), reps)
            fl = 2.0 * n * hh * ww * cout * 9 * cin
            print(f"| {n}x{h}x{w} | {cin}->{cout}{' +pool' if pool else ''} | {hh}x{ww} | {t_own:.1f} | {fl / t_own / 1e6:.1f} "
                  f"({fl / 2.25 / t_own / 1e6:.1f}) | {t_conv:.1f} | {t_lib:.1f} | {t_lib / t_own:.2f}x |")
            tot_own += t_own; tot_lib += t_lib
        print(f"| {n}x{h}x{w} | **7 layers** | | {tot_own:.1f} | | | {tot_lib:.1f} | {tot_lib / tot_own:.2f}x |")
    # whole trunk incl. conv1 and feature hand-over, both implementations, detector-style taps
    from gen6d_amd.network import name2network
    from gen6d_amd.network.params import fold_vgg
    net = name2network["detector"]({"name": "t"}).eval()
    net.load_state_dict(synth.synth_state_dict("detector"))
    net = net.cuda()
    folded = fold_vgg(net, "backbone.features")
    packed = [folded[0]] + [(B.winograd_filters(wt), b) for wt, b in folded[1:]]
    print("\n| images | whole trunk own us | whole trunk library us (incl. NCHW->NHWC of the 3 taps) |")
    print("|---|---|---|")
    for n, h, w in sizes:
        img = torch.rand((n, 3, h, w), device=dev)
        def own():
            B.vgg_taps_cl(packed, img, {"c5", "c7_pre", "p7"})
        def lib():
            t = LT.vgg_taps(folded, img, {"c5", "c7_pre", "p7"})
            for k in ("c5", "c7_pre", "p7"):
                f = t[k].contiguous()
                ops.nchw_to_nhwc(f, torch.empty((f.shape[0], 1, f.shape[2], f.shape[3], f.shape[1]), device=dev), False)
        with torch.no_grad():
            print(f"| {n}x{h}x{w} | {timeit(own, reps):.1f} | {timeit(lib, reps):.1f} |")


if __name__ == "__main__":
    main()
