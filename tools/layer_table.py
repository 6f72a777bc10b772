#!/usr/bin/env python
"""Per-layer table of the g6d_conv_igemm family inside the real pipeline (one query, serialised eager launches, HIP
events around every launch incl. its split-K reduce): which layers the conv time of a query is made of."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gen6d_amd import ops, synth  # noqa: E402
from gen6d_amd.pipeline import TensorPipeline  # noqa: E402


def main():
    dev = torch.device("cuda")
    pipe = TensorPipeline(dev); pipe.build()
    full = synth.imgs_to_tensor(synth.synth_images(1, 480, 640, seed=100)).to(dev)
    crop = synth.imgs_to_tensor(synth.synth_images(1, 128, 128, seed=200)).to(dev)
    r = pipe.ref_dev
    stages = [
        ("detector", lambda: pipe.detector.detect_impl(full)),
        ("selector", lambda: pipe.selector.compute_view_point_feats(crop)),
        ("refiner step", lambda: pipe.refiner._step(crop, r["Ks_in"][0], pipe.iter_poses[0][0], r["ref_imgs"][0], r["ref_Ks"][0], r["ref_poses"][0])),
    ]
    ops.SERIAL = True
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
            n = len(runs[0])
            tot_us = tot_fl = 0.0
            print(f"\n## {name}: {n} launches\n\n| layer | us | TFLOP/s |\n|---|---|---|")
            for i in range(n):
                us = min(run[i][1].elapsed_time(run[i][2]) for run in runs) * 1e3
                fl = runs[0][i][0]
                tot_us += us; tot_fl += fl
                print(f"| {runs[0][i][3]} | {us:.1f} | {fl / us / 1e6:.1f} |")
            print(f"| **total** | {tot_us:.0f} | {tot_fl / tot_us / 1e6:.1f} |")


if __name__ == "__main__":
    main()
