#!/usr/bin/env python
"""Per-layer table of the MFMA kernel families inside the real pipeline (one query, or a batch of BATCH=n queries that share
every launch; serialised eager launches, HIP events around every launch): which layers the time of a query is made of, plus the
whole stage's time (everything incl. the non-MFMA kernels) per query."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import toolenv  # noqa: E402,F401  (G6D_LIB_PATH / KNOBS)
from gen6d_amd import ops, synth  # noqa: E402
from gen6d_amd.pipeline import TensorPipeline  # noqa: E402


def main():
    dev = torch.device("cuda")
    pipe = TensorPipeline(dev); pipe.build()
    B = int(os.environ.get("BATCH", "1"))
    idx = torch.arange(B, device=dev) % 4
    full = synth.imgs_to_tensor(synth.synth_images(4, 480, 640, seed=100)).to(dev)[idx]
    crop = synth.imgs_to_tensor(synth.synth_images(4, 128, 128, seed=200)).to(dev)[idx]
    r = pipe.ref_dev
    ex = lambda t: t.expand(B, *t.shape[1:]).contiguous()
    if B == 1:
        ref_step = lambda: pipe.refiner._step(crop, r["Ks_in"][0], pipe.iter_poses[0][0], r["ref_imgs"][0], r["ref_Ks"][0], r["ref_poses"][0])
    else:
        ref_step = lambda: pipe.refiner._step(crop, ex(r["Ks_in"]), ex(pipe.iter_poses[0]), r["ref_imgs"].expand(B, *r["ref_imgs"].shape[1:]),
                                              ex(r["ref_Ks"]), ex(r["ref_poses"]))
    stages = [
        ("detector", lambda: pipe.detector.detect_impl(full)),
        ("selector", lambda: pipe.selector.compute_view_point_feats(crop)),
        ("refiner step", ref_step),
    ]
    print(f"# batch of {B} quer{'y' if B == 1 else 'ies'} per launch; us columns are per launch, totals also per query")
    ops.SERIAL = True
    if os.environ.get("LOWP"):                          # LOWP=fp16 / bf16: the reduced-precision mode's table
        ops.MATH_MODE = {"bf16": 1, "fp16": 2}[os.environ["LOWP"]]
        print(f"# reduced-precision matrix-core mode: {os.environ['LOWP']}")
    reps = 5
    with torch.no_grad():
        for name, fn in stages:
            fn(); fn()
            torch.cuda.synchronize()
            runs = []
            for _ in range(reps):
                ops.PROFILE = []
                fn()
                torch.cuda.synchronize()
                runs.append(ops.PROFILE)
            ops.PROFILE = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            stage_us = e0.elapsed_time(e1) / reps * 1e3
            n = len(runs[0])
            tot_us = tot_fl = 0.0
            print(f"\n## {name}: {n} MFMA-family launches; whole stage {stage_us:.0f} us = {stage_us / B:.0f} us per query\n\n| layer | us | TFLOP/s | algorithmic GB/s |\n|---|---|---|---|")
            for i in range(n):
                us = min(run[i][1].elapsed_time(run[i][2]) for run in runs) * 1e3
                fl = runs[0][i][0]
                tot_us += us; tot_fl += fl
                gbs = f" {runs[0][i][4] / us / 1e3:.0f} |" if len(runs[0][i]) > 4 else ""     # algorithmic bytes (every operand once) / time
                print(f"| {runs[0][i][3]} | {us:.1f} | {fl / us / 1e6:.1f} |{gbs}")
            print(f"| **total** | {tot_us:.0f} ({tot_us / B:.0f} per query) | {tot_fl / tot_us / 1e6:.1f} |")


if __name__ == "__main__":
    main()
