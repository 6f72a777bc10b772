"""All-rows parity of the reduced-precision modes (gen6d_amd/bars.py) on the 4 bench queries + 16 held-out queries, per mode, with the
per-column worst errors: `python tools/lowp_rows.py [mode ...]` on the GPU box (modes: fp32 fp16 fp16all bf16mix bf16 bf16all)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gen6d_amd import bars, lib, ops, synth          # noqa: E402
from gen6d_amd.pipeline import TensorPipeline        # noqa: E402

SCHEMES = {   # name -> (ops.math_mode of detector / refiner, selector math_mode override, selector keep-list or None = cfg default)
    "fp32": ("fp32", None, None), "fp16": ("fp16", None, None), "fp16all": ("fp16", None, ()),
    "bf16mix": ("bf16", "fp16", None), "bf16": ("bf16", None, None), "bf16all": ("bf16", None, ()),
}


def query_sets(dev):
    out = {}
    for tag, n, fs, cs in (("bench4", 4, 100, 200), ("heldout16", 16, 300, 400)):
        g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_rows.npz" if n == 4 else "pipeline_rows_heldout.npz"))
        out[tag] = (synth.imgs_to_tensor(synth.synth_images(n, 480, 640, seed=fs)).to(dev),
                    synth.imgs_to_tensor(synth.synth_images(n, 128, 128, seed=cs)).to(dev),
                    torch.from_numpy(g["rows"]).float(), torch.from_numpy(g["logits"]).float())
    return out


def run_mode(pipe, name, sets):
    mode, sel_mode, keep = SCHEMES[name]
    pipe.selector.cfg["math_mode"] = sel_mode
    pipe.selector.cfg["lowp_keep_fp32"] = pipe.selector.default_cfg["lowp_keep_fp32"] if keep is None else keep
    res = {}
    try:
        with ops.math_mode(mode), torch.no_grad():
            for tag, (fulls, crops, _, _) in sets.items():
                res[tag] = (pipe.query(fulls, crops).cpu(), pipe.selector.compute_view_point_feats(crops)[0].cpu())
    finally:
        pipe.selector.cfg["math_mode"] = None
        pipe.selector.cfg["lowp_keep_fp32"] = pipe.selector.default_cfg["lowp_keep_fp32"]
    return res


def main():
    lib.load()
    dev = torch.device("cuda", 0)
    modes = sys.argv[1:] or ["fp16", "fp16all", "bf16mix", "bf16"]
    pipe = TensorPipeline(dev)
    pipe.build()
    ops.SERIAL = True
    sets = query_sets(dev)
    r32 = run_mode(pipe, "fp32", sets)
    out = {"fp32": {tag: {"vs_reference": bars.row_errors(r32[tag][0], sets[tag][2]), "logits": bars.logit_errors(r32[tag][1], sets[tag][3])}
                    for tag in sets}}
    for m in modes:
        if m == "fp32":
            continue
        r = run_mode(pipe, m, sets)
        out[m] = {tag: bars.lowp_all_rows(r[tag][0], sets[tag][2], r32[tag][0], r[tag][1], sets[tag][3]) for tag in sets}
        # per column: worst relative error against the reference's rows over all 20 queries
        rows = torch.cat([r[t][0] for t in sets]); ref = torch.cat([sets[t][2] for t in sets])
        rel = ((rows - ref).abs() / ref.abs().clamp(min=1.0)).max(0)[0]
        out[m]["worst_rel_per_column"] = [float(f"{v:.3g}") for v in rel.tolist()]
    print(json.dumps(out, indent=1))
    for m, v in out.items():
        if m != "fp32":
            print(m, {t: (v[t]["ok_rows"], v[t]["ok_logits"]) for t in sets})


if __name__ == "__main__":
    main()
