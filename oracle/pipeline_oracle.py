"""CPU ORACLE (test infrastructure): the per-query tensor pipeline detect -> select -> refine x N restated with
oracle/gen6d_oracle.py, mirroring gen6d_amd.pipeline.TensorPipeline on the same synthetic state.  Used only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import torch

from . import gen6d_oracle as O


def build_state(state_dicts, det_refs, sel_case):
    """One-time reference state (not part of the per-query time): detector filters, selector cache + embedding."""
    with torch.no_grad():
        det_feats = O.detector_ref_feats(state_dicts["detector"], det_refs)
        cache, embed = O.selector_ref_state(state_dicts["selector"], sel_case["ref_imgs"], sel_case["ref_poses"],
                                            sel_case["object_center"], sel_case["object_vert"])
    return {"det_feats": det_feats, "sel_cache": cache, "sel_embed": embed}


def query(state_dicts, state, ref_case, iter_poses, que_full, que_crop, stage_s=None):
    """One query: returns the same [1, 5 + 7 * steps] row as TensorPipeline.query (every refinement step's outputs) plus the selector logits.
    `stage_s`: optional dict that receives the wall seconds of the detector / selector / refiner stages."""
    import time
    with torch.no_grad():
        t0 = time.perf_counter()
        out = O.detector_detect(state_dicts["detector"], que_full, state["det_feats"])
        pos, scl = O.detector_parse(out)
        t1 = time.perf_counter()
        logits, angles = O.selector_forward(state_dicts["selector"], que_crop, state["sel_cache"], state["sel_embed"])
        idx, ang = O.selector_select(logits, angles)
        t2 = time.perf_counter()
        steps = []
        for p in iter_poses:
            o = O.refiner_forward(state_dicts["refiner"], que_crop, ref_case["Ks_in"], p, ref_case["ref_imgs"],
                                  ref_case["ref_Ks"], ref_case["ref_poses"])
            steps += [o["rotation"], o["offset"], o["scale"]]
        t3 = time.perf_counter()
    if stage_s is not None:
        stage_s.update(detector=t1 - t0, selector=t2 - t1, refiner=t3 - t2)
    row = torch.cat([pos, scl[:, None], idx[:, None].float(), ang[:, None]] + steps, 1)
    return row, logits
