"""Per-layer times of the direct 16-bit convolution (g6d_conv16_direct_multi) next to the 16-bit Winograd kernel it replaces
(g6d_wino16_conv3x3_multi, fp32 activations) on the trunk shapes of the batched pipeline: the detector's pyramid (4 map sizes per
launch) and the refiner's crops.   python tools/conv16_bench.py [batch=16] [fp16|bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import toolenv                                                  # noqa: E402,F401  (G6D_LIB_PATH / KNOBS)
from gen6d_amd import lib, ops                                  # noqa: E402
from gen6d_amd.network.backbone import winograd43_filters, winograd_filters16       # noqa: E402


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
    print(f"# batch {B}: the direct kernel on 16-bit activations (g6d_conv16_direct_multi) per trunk layer, us per launch (direct-form TFLOP/s)")
    print(f"# reduced precision ({mode}): halo-patch kernel | per-tap kernel, filters in registers | per-tap kernel, filters through LDS | the 16-bit Winograd kernel they replace")
    print("# fp32 path: fp16 hi / lo pairs (3 MFMAs per product, fp32-class results) on the halo-patch kernel | on the per-tap kernel | the F(4x4,3x3) fp32 kernel they replace")
    print(f"| layer | maps | Cin -> Cout | outputs | {mode} halo | {mode} reg-B | {mode} LDS-B | wino16 | halo vs wino16 | pairs halo | pairs reg-B | wino43 fp32 | halo vs wino43 |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    tot = [0.0] * 7
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
            wt = w.permute(0, 2, 3, 1).reshape(co, 9, ci).contiguous()
            mm = {"fp16": 2, "bf16": 1}[mode]
            f_reg, f_lds, f_pair = ops.conv16_pack(wt, mm, 1), ops.conv16_pack(wt, mm, 0), ops.conv16_pack(wt, 3, 1)
            xs32 = ops.alloc_like_segments([(n, h, ww, ci) for n, h, ww in shapes], dev)
            for x in xs32:
                x.copy_(torch.rand(x.shape, generator=g))
            xs16 = [x.to(t16) for x in xs32]
            xsp = [torch.stack([x.half(), (x - x.half().float()).half()], -2).contiguous() for x in xs32]
            u16, u43 = winograd_filters16(w, t16), winograd43_filters(w)
            flops = sum(2.0 * n * h * ww * co * 9 * ci for n, h, ww in shapes)
            o16 = lambda on: "t16" if on else None
            run16 = lambda f: timed(lambda: ops.conv16_direct_multi(xs16, f, bias, relu=True, full=o16(full), pool=o16(pool)))
            runp = lambda: timed(lambda: ops.conv16_direct_multi(xsp, f_pair, bias, relu=True, full=o16(full), pool=o16(pool)))
            lib.set_knob("conv16_halo", 1)
            t_halo, t_phalo = run16(f_reg), runp()
            lib.set_knob("conv16_halo", 0)
            t_reg, t_pair = run16(f_reg), runp()
            lib.reset_knobs()
            t_lds = run16(f_lds)
            with ops.math_mode(mode):
                t_w16 = timed(lambda: ops.wino16_conv3x3_multi(xs32, u16, bias, relu=True, full=full, pool=pool))
            t_w43 = timed(lambda: ops.wino43_conv3x3_multi(xs32, u43, bias, relu=True, full=full, pool=pool))
            for i, v in enumerate((t_halo, t_reg, t_lds, t_w16, t_phalo, t_pair, t_w43)):
                tot[i] += v
            tf = lambda us: f"{us:.0f} ({flops / us / 1e6:.0f})"
            print(f"| {tag} | {'+'.join(f'{n}x{h}x{ww}' for n, h, ww in shapes)} | {ci} -> {co} | {'full ' if full else ''}{'pool' if pool else ''} | "
                  f"{tf(t_halo)} | {tf(t_reg)} | {tf(t_lds)} | {tf(t_w16)} | {t_w16 / t_halo:.2f}x | {tf(t_phalo)} | {tf(t_pair)} | {tf(t_w43)} | {t_w43 / t_phalo:.2f}x |")
    print(f"| **total us** | | | | {tot[0]:.0f} | {tot[1]:.0f} | {tot[2]:.0f} | {tot[3]:.0f} | {tot[3] / tot[0]:.2f}x | {tot[4]:.0f} | {tot[5]:.0f} | {tot[6]:.0f} | {tot[6] / tot[4]:.2f}x |")


if __name__ == "__main__":
    main()
