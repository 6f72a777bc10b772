"""Which part of the refiner makes the reduced-precision pose-head error?  Runs the three refinement steps of the 20 gate queries
(4 bench + 16 held-out crops) with ONE part on 16-bit matrix-core operands at a time (cfg lowp_only), then the candidate keep-lists,
and prints the worst relative error of the 21 pose-head columns against the fp32 path (bar: gen6d_amd/bars.LOWP_REL) with the
time of the three steps per batch of 16.   python tools/lowp_refiner_sensitivity.py [fp16|bf16]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gen6d_amd import bars, lib, ops, synth          # noqa: E402
from gen6d_amd.pipeline import TensorPipeline        # noqa: E402

PARTS = ("trunk", "featnet", "embed", "stack", "tail")


def heads(pipe, crops):
    """[n, 21] pose-head columns of the three refinement steps (TensorPipeline.query without detector / selector)."""
    r, n = pipe.ref_dev, crops.shape[0]
    ex = lambda t: t.expand(n, *t.shape[1:])
    cols = []
    for it in range(pipe.refine_iter):
        cam = pipe._canned(n, it)
        cols += list(pipe.refiner._step(crops, cam[0], cam[1], ex(r["ref_imgs"]), cam[2], cam[3]))
    return torch.cat(cols, 1)


def main():
    lib.load()
    mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    dev = torch.device("cuda", 0)
    pipe = TensorPipeline(dev)
    pipe.build()
    ops.SERIAL = True
    crops = torch.cat([synth.imgs_to_tensor(synth.synth_images(4, 128, 128, seed=200)),
                       synth.imgs_to_tensor(synth.synth_images(16, 128, 128, seed=400))]).to(dev)
    cfg = pipe.refiner.cfg

    def run(only=None, keep=()):
        cfg["lowp_only"], cfg["lowp_keep_fp32"] = only, keep
        with ops.math_mode(mode), torch.no_grad():
            out = torch.cat([heads(pipe, crops[0:4]), heads(pipe, crops[4:20])]).cpu()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                heads(pipe, crops[4:20])
            torch.cuda.synchronize()
        cfg["lowp_only"], cfg["lowp_keep_fp32"] = None, ()
        return out, (time.perf_counter() - t0) / 3 * 1e3

    with torch.no_grad():
        ref = torch.cat([heads(pipe, crops[0:4]), heads(pipe, crops[4:20])]).cpu()
    rel = lambda o: float(((o - ref).abs() / ref.abs().clamp(min=1.0)).max())
    print(f"# refiner pose heads, {mode} operands vs the fp32 path, 20 queries x 3 steps (bar {bars.LOWP_REL}); ms = 3 steps of 16 queries, eager")
    print("| scheme | worst rel err | ms |\n|---|---|---|")
    for p in PARTS:
        o, ms = run(only=(p,))
        print(f"| only {p} on {mode} | {rel(o):.2e} | {ms:.2f} |")
    for keep in ((), ("tail",), ("stack", "tail"), ("embed",), ("embed", "stack", "tail"), ("trunk",), ("trunk", "featnet"), ("featnet",),
                 ("trunk", "featnet", "tail"), PARTS):
        o, ms = run(keep=keep)
        print(f"| all {mode}, kept on fp32: {', '.join(keep) or 'nothing'} | {rel(o):.2e} | {ms:.2f} |")


if __name__ == "__main__":
    main()
