"""Per-layer times of the direct 16-bit convolution (g6d_conv16_direct_multi) next to the 16-bit Winograd kernel it replaces
(g6d_wino16_conv3x3_multi, fp32 activations) on the trunk shapes of the batched pipeline: the detector's pyramid (4 map sizes per
launch) and the refiner's crops.   python tools/conv16_bench.py [batch=16] [fp16|bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gen6d_amd import lib, ops                                  # noqa: E402
from gen6d_amd.network.backbone import winograd_filters16       # noqa: E402


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    lib.load()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    mode = sys.argv[2] if len(sys.argv) > 2 else "fp16"
    t16 = {"fp16": torch.float16, "bf16": torch.bfloat16}[mode]
    dev = torch.device("cuda", 0)
    pyr = [(352, 464), (240, 320), (176, 240), (128, 160)]          # the four scales of a 480x640 query AFTER the first layer's pool (1/2 resolution)
    layers = [("pyr/2", 1, 64, 128, False, True), ("pyr/4", 2, 128, 256, True, False), ("pyr/4", 2, 256, 256, False, True),
              ("pyr/8", 4, 256, 512, True, False), ("pyr/8", 4, 512, 512, True, True), ("pyr/16", 8, 512, 512, True, False),
              ("pyr/16", 8, 512, 512, True, True)]
    crops = [("crop/2", 64, 64, 128, False, True), ("crop/4", 32, 128, 256, True, False), ("crop/4", 32, 256, 256, True, True),
             ("crop/8", 16, 256, 512, True, False), ("crop/8", 16, 512, 512, True, True), ("crop/16", 8, 512, 512, True, False),
             ("crop/16", 8, 512, 512, True, False)]
    g = torch.Generator().manual_seed(1)
    print(f"# batch {B}, {mode}: direct 16-bit convolution on 16-bit activations vs the 16-bit Winograd kernel on fp32 activations (us per launch, direct-form TFLOP/s)")
    print("| layer | maps | Cin -> Cout | outputs | conv16 direct us | TFLOP/s | wino16 us | direct-form TFLOP/s | speed-up |\n|---|---|---|---|---|---|---|---|---|")
    tot = [0.0, 0.0]
    for group in ("pyramid", "crops"):
        for spec in (layers if group == "pyramid" else crops):
            if group == "pyramid":
                tag, div, ci, co, full, pool = spec
                shapes = [(B, h // div, w // div) for h, w in pyr]
            else:
                tag, hw, ci, co, full, pool = spec
                shapes = [(7 * B, hw, hw)]
            w = ((torch.rand((co, ci, 3, 3), generator=g) * 2 - 1) * (1.0 / (9 * ci)) ** 0.5 * 3).to(dev)
            bias = torch.zeros(co, device=dev)
            w16 = w.permute(0, 2, 3, 1).reshape(co, 9, ci).contiguous().to(t16)
            xs16 = [torch.rand((n, h, ww, ci), generator=g).to(dev).to(t16) for n, h, ww in shapes]
            xs32 = [x.float() for x in xs16]
            u16 = winograd_filters16(w, t16)
            flops = sum(2.0 * n * h * ww * co * 9 * ci for n, h, ww in shapes)
            with ops.math_mode(mode):
                td = timed(lambda: ops.conv16_direct_multi(xs16, w16, bias, relu=True, full=t16 if full else None, pool=t16 if pool else None))
                tw = timed(lambda: ops.wino16_conv3x3_multi(xs32, u16, bias, relu=True, full=full, pool=pool))
            tot[0] += td; tot[1] += tw
            print(f"| {tag} | {'+'.join(f'{n}x{h}x{ww}' for n, h, ww in shapes)} | {ci} -> {co} | {'full ' if full else ''}{'pool' if pool else ''} | "
                  f"{td:.1f} | {flops / td / 1e6:.0f} | {tw:.1f} | {flops / tw / 1e6:.0f} | {tw / td:.2f}x |")
    print(f"| **total** | | | | {tot[0]:.0f} | | {tot[1]:.0f} | | {tot[1] / tot[0]:.2f}x |")


if __name__ == "__main__":
    main()
