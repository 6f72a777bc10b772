#!/usr/bin/env python
"""Reduced-precision selector: logit error against the reference's own logits (tests/golden/pipeline_rows.npz, 4 synthetic queries) and
time per batch of 8 under mixed schemes — which parts have to stay on fp32 operands for the error to stay below a quarter of the
smallest top-2 margin.  Usage (GPU box): python tools/lowp_selector_schemes.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import toolenv  # noqa: E402,F401
from gen6d_amd import ops, synth  # noqa: E402
from gen6d_amd.pipeline import TensorPipeline  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    dev = torch.device("cuda")
    pipe = TensorPipeline(dev); pipe.build()
    gold = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "pipeline_rows.npz"))["logits"]).float()
    top2 = gold.topk(2, 1)[0]
    margin = float((top2[:, 0] - top2[:, 1]).min())
    crops = synth.imgs_to_tensor(synth.synth_images(4, 128, 128, seed=200)).to(dev)
    c8 = crops[torch.arange(8, device=dev) % 4]
    ops.SERIAL = True
    print(f"smallest top-2 margin of the 4 queries: {margin:.4f}; bar = margin / 4 = {margin / 4:.4f}\n")
    print("| mode | kept on fp32 operands | max logit err | err / margin | arg-max equal | selector ms per batch of 8 |")
    print("|---|---|---|---|---|---|")
    for mode in ("fp32", "fp16", "bf16"):
        for keep in ((), ("trunk",), ("product",), ("trunk", "product")):
            if mode == "fp32" and keep:
                continue
            pipe.selector.cfg["lowp_keep_fp32"] = keep
            with ops.math_mode(mode), torch.no_grad():
                lg = pipe.selector.compute_view_point_feats(crops)[0].cpu()
                for _ in range(2): pipe.selector.compute_view_point_feats(c8)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(5): pipe.selector.compute_view_point_feats(c8)
                e1.record(); torch.cuda.synchronize()
            err = float((lg - gold).abs().max())
            print(f"| {mode} | {', '.join(keep) or '-'} | {err:.2e} | {err / margin:.3f} | {bool((lg.argmax(1) == gold.argmax(1)).all())} | {e0.elapsed_time(e1) / 5:.3f} |")


if __name__ == "__main__":
    main()
