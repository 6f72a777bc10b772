#!/usr/bin/env python
"""Per-stage GPU time of the tensor pipeline (HIP events, eager launches): detector / selector / one refiner step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gen6d_amd import ops, synth  # noqa: E402
from gen6d_amd.pipeline import TensorPipeline  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda")
    pipe = TensorPipeline(dev); pipe.build()
    full = synth.imgs_to_tensor(synth.synth_images(1, 480, 640, seed=100)).to(dev)
    crop = synth.imgs_to_tensor(synth.synth_images(1, 128, 128, seed=200)).to(dev)
    r = pipe.ref_dev
    with torch.no_grad():
        for serial in (False, True):
            ops.SERIAL = serial
            d = timed(lambda: pipe.detector.detect_impl(full))
            s = timed(lambda: pipe.selector.compute_view_point_feats(crop))
            f = timed(lambda: pipe.refiner._step(crop, r["Ks_in"][0], pipe.iter_poses[0][0], r["ref_imgs"][0], r["ref_Ks"][0], r["ref_poses"][0]))
            trunk = timed(lambda: pipe.detector.extract_feats(torch.nn.functional.interpolate(full, size=(704, 928), mode="bilinear")))
            print(f"fork_join {'off' if serial else 'on '}: detector {d:.2f} ms  selector {s:.2f} ms  refiner step {f:.2f} ms (x3 = {3 * f:.2f})  "
                  f"sum {d + s + 3 * f:.2f} ms   [detector trunk @704x928 alone {trunk:.2f} ms]")


if __name__ == "__main__":
    main()
