#!/usr/bin/env python
"""Reduced-precision selector, part by part: which parts of the network carry the fp16 logit error to the 1/4-margin
bar.  For every part P of {trunk, product, corr<l>.<i>, fuse, tail}: the logit error against the reference's own logits
(tests/golden/pipeline_rows.npz, 4 synthetic queries) with ONLY P on fp16 operands (its isolated contribution) and with everything
BUT P on fp16 (what keeping P on fp32 buys), then candidate keep-lists with their selector time per batch of 8.
Usage (GPU box): python tools/lowp_selector_sensitivity.py [fp16|bf16]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import toolenv  # noqa: E402,F401
from gen6d_amd import ops, synth  # noqa: E402
from gen6d_amd.network.selector import _CORR  # noqa: E402
from gen6d_amd.pipeline import TensorPipeline  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    dev = torch.device("cuda")
    pipe = TensorPipeline(dev); pipe.build()
    sel = pipe.selector
    gold = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "pipeline_rows.npz"))["logits"]).float()
    top2 = gold.topk(2, 1)[0]
    margin = float((top2[:, 0] - top2[:, 1]).min())
    crops = synth.imgs_to_tensor(synth.synth_images(4, 128, 128, seed=200)).to(dev)
    c8 = crops[torch.arange(8, device=dev) % 4]
    ops.SERIAL = True

    def run(keep=(), only=None, timed=False):
        sel.cfg["lowp_keep_fp32"], sel.cfg["lowp_only"] = tuple(keep), (tuple(only) if only is not None else None)
        with ops.math_mode(mode), torch.no_grad():
            lg = sel.compute_view_point_feats(crops)[0].cpu()
            ms = None
            if timed:
                for _ in range(2): sel.compute_view_point_feats(c8)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(5): sel.compute_view_point_feats(c8)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
        sel.cfg["lowp_keep_fp32"], sel.cfg["lowp_only"] = (), None
        err = float((lg - gold).abs().max())
        return err, bool((lg.argmax(1) == gold.argmax(1)).all()), ms

    parts = ["trunk"] + [f"corr{l}.{i}" for l, layers in enumerate(_CORR) for i in range(len(layers))] + ["fuse", "tail"]
    print(f"mode {mode}; smallest top-2 margin of the 4 queries {margin:.4f}; bar = margin / 4 = {margin / 4:.4f}\n")
    e32 = run(only=())[0]
    eall, aall, tall = run(timed=True)
    print(f"all parts on fp32 operands: {e32:.2e}; all on {mode}: {eall:.2e} ({eall / margin:.3f} of the margin), arg-max equal {aall}, {tall:.2f} ms per batch of 8\n")
    print(f"| part | error with ONLY this part on {mode} | / margin | error with everything BUT this part on {mode} | / margin |")
    print("|---|---|---|---|---|")
    iso = {}
    for p_ in parts:
        e1, _, _ = run(only=(p_,))
        e2, _, _ = run(keep=(p_,))
        iso[p_] = e1
        print(f"| {p_} | {e1:.2e} | {e1 / margin:.3f} | {e2:.2e} | {e2 / margin:.3f} |")
    # greedy keep-list: move the part with the largest isolated error to fp32 until the bar holds with 20 % head-room
    print("\n| kept on fp32 operands (greedy by isolated error) | max logit err | err / margin | arg-max equal | selector ms per batch of 8 |")
    print("|---|---|---|---|---|")
    keep = []
    for p_ in sorted(parts, key=lambda k: -iso[k]):
        keep.append(p_)
        e, a, t = run(keep=keep, timed=True)
        print(f"| {', '.join(keep)} | {e:.2e} | {e / margin:.3f} | {a} | {t:.3f} |")
        if e <= 0.2 * margin:
            break
    for name, kl in (("trunk, tail", ("trunk", "tail")), ("trunk, tail, fuse", ("trunk", "tail", "fuse")),
                     ("trunk, tail, fuse, corr0.0, corr1.0", ("trunk", "tail", "fuse", "corr0.0", "corr1.0")),
                     ("product, stack (the InstanceNorm stacks)", ("product", "stack")), ("trunk, product, stack", ("trunk", "product", "stack")),
                     ("stack", ("stack",)), ("fuse, tail", ("fuse", "tail"))):
        e, a, t = run(keep=kl, timed=True)
        print(f"| {name} | {e:.2e} | {e / margin:.3f} | {a} | {t:.3f} |")
    e, a, t = run(only=(), timed=True)
    print(f"| everything (fp32 selector) | {e:.2e} | {e / margin:.3f} | {a} | {t:.3f} |")


if __name__ == "__main__":
    main()
