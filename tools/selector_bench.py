#!/usr/bin/env python
"""Selector-only throughput (BASELINE.json configs[1]: one synthetic 128x128 query against rfn reference views x an
in-plane rotations) for the reference-count sweep the north star names (32 / 64 / 128 views) and the an = 36 stress
case.  hipGraph replay of `compute_view_point_feats`, HIP events; MAC count from SURVEY.md 8(d):
    2.444 G (query VGG) + rfn*an*0.21392 G + rfn*0.0055 G  (+0.26 M per reference per rotation beyond 5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import toolenv  # noqa: E402,F401  (G6D_LIB_PATH / KNOBS)
from gen6d_amd import synth  # noqa: E402
from gen6d_amd.network import name2network  # noqa: E402


def main():
    dev = torch.device("cuda")
    reps = int(os.environ.get("REPS", "20"))
    print("| rfn | an | ref cache MB | ms / query | queries/s | GMAC | TFLOP/s |\n|---|---|---|---|---|---|---|")
    for rfn, an in ((32, 5), (64, 5), (128, 5), (64, 36)):
        sel = name2network["selector"]({"name": "selector_synth", "selector_angle_num": an})
        sel.load_state_dict(synth.synth_state_dict("selector", 1234, an))
        sel.to(dev).eval()
        case = synth.selector_case(rfn, an, 1)
        crop = synth.imgs_to_tensor(synth.synth_images(1, 128, 128, seed=200)).to(dev)
        with torch.no_grad():
            sel.extract_ref_feats(case["ref_imgs"].to(dev), case["ref_poses"].to(dev), case["object_center"].to(dev),
                                  case["object_vert"].to(dev))
            stream = torch.cuda.Stream(device=dev)
            stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(stream):
                for _ in range(2):
                    sel.compute_view_point_feats(crop)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                out = sel.compute_view_point_feats(crop)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                graph.replay()
                e0.record()
                for _ in range(reps):
                    graph.replay()
                e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gmac = 2.444 + rfn * an * 0.21392 + rfn * (0.0055 + 0.00026 * (an - 5))
        cache_mb = rfn * an * 688128 / 1e6
        print(f"| {rfn} | {an} | {cache_mb:.0f} | {ms:.3f} | {1e3 / ms:.1f} | {gmac:.1f} | {2 * gmac / ms:.1f} |")
        del sel, graph, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
