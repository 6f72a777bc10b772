"""Round-2 golden vectors, again outputs of the REFERENCE's own modules run in the build container (same import stubs
as make_golden.py, no reference file edited).  They pin exactly what bench.py times and what VERDICT r01 listed as
timed-but-unchecked:

    det_head.npz      Detector, 480x640 query vs 32 refs (the headline detector call)            detector.py:232-266
    sel_head.npz      ViewpointSelector, 64 refs x 5 rotations (the headline selector call)     selector.py:177-215
    sel_128x5.npz     128 refs x 5 rotations   (BASELINE configs[3] size)
    sel_32x5.npz      32 refs x 5 rotations    (north_star's 32 / 64 / 128 sweep; round 5)
    sel_64x36.npz     64 refs x 36 rotations   (BASELINE configs[1])
    operator.npz      network/operator.py:4-24 normalize_coords / pose_apply_th / generate_coords
    ref_grids.npz     VolumeRefiner.forward(...)["grids"] (inference=False branch)               refiner.py:262-268
    pipeline_rows.npz the [1,26] result rows (5 + 7 per refinement step) of bench.py's four synthetic queries (detector + selector + 3 refiner
                      steps on gen6d_amd.pipeline.TensorPipeline's synthetic state), so that bench.py can state parity
                      against the reference itself on a box that has no /root/reference

    python tests/golden/make_golden_r02.py [name ...]      # default: all
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG          # noqa: E402  (stubs + reference loader)


def np_(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    print(f"[{name}] written", {k: np.asarray(v).shape for k, v in arrs.items()})


def det_head(n2n, synth):
    net = n2n["detector"]({"network": "detector", "name": "g"}).eval()
    net.load_state_dict(synth.synth_state_dict("detector"))
    case = synth.detector_case(32, 480, 640)
    with torch.no_grad():
        out = net({"ref_imgs_info": {"imgs": case["ref_imgs"]}, "que_imgs_info": {"imgs": case["que_imgs"]}})
        pos, scl = net.parse_detection(out["scores"], out["select_pr_scale"], out["select_pr_offset"], 8)
    save("det_head", rfn=32, hq=480, wq=640, sha_inputs=synth.fingerprint(case),
         sha_weights=synth.fingerprint(synth.synth_state_dict("detector")), positions=pos.numpy(), scales=scl.numpy(),
         **np_({k: out[k] for k in ("scores", "select_pr_offset", "select_pr_scale", "que_select_id")}))


def _selector(n2n, synth, tag, rfn, an):
    net = n2n["selector"]({"network": "selector", "name": "g", "selector_angle_num": an}).eval()
    net.load_state_dict(synth.synth_state_dict("selector", an=an))
    case = synth.selector_case(rfn, an)
    t0 = time.time()
    with torch.no_grad():
        out = net({"ref_imgs": case["ref_imgs"], "ref_imgs_info": {"poses": case["ref_poses"]},
                   "object_center": case["object_center"], "object_vert": case["object_vert"],
                   "que_imgs_info": {"imgs": case["que_imgs"]}, "eval": True})
    save(tag, rfn=rfn, an=an, logits=out["ref_vp_logits"].numpy(), angles=out["angles_pr"].numpy(),
         sha_inputs=synth.fingerprint(case), sha_weights=synth.fingerprint(synth.synth_state_dict("selector", an=an)))
    print(tag, f"{time.time() - t0:.1f}s argmax", out["ref_vp_logits"].argmax(1).numpy())


def operator_golden():
    import network.operator as rop
    g = torch.Generator().manual_seed(11)
    coords = torch.rand((5, 7, 2), generator=g) * torch.tensor([96.0, 64.0])
    poses = torch.randn((3, 3, 4), generator=g)
    pts = torch.randn((3, 9, 3), generator=g)
    save("operator", coords=coords.numpy(), h=64, w=96, norm=rop.normalize_coords(coords, 64, 96).numpy(),
         poses=poses.numpy(), pts=pts.numpy(), applied=rop.pose_apply_th(poses, pts).numpy(),
         gen=rop.generate_coords(6, 9, "cpu").numpy())


def ref_grids(n2n, synth):
    net = n2n["refiner"]({"network": "refiner", "name": "g"}).eval()
    net.load_state_dict(synth.synth_state_dict("refiner"))
    case = synth.refiner_case()
    with torch.no_grad():
        out = net({"que_imgs_info": {"imgs": case["que_imgs"], "Ks_in": case["Ks_in"], "poses_in": case["poses_in"]},
                   "ref_imgs_info": {"imgs": case["ref_imgs"], "Ks": case["ref_Ks"], "poses": case["ref_poses"]}})
    save("ref_grids", grids=out["grids"][:, ::7].numpy(), stride=7, sha_inputs=synth.fingerprint(case),
         sha_weights=synth.fingerprint(synth.synth_state_dict("refiner")), **np_({k: out[k] for k in ("rotation", "offset", "scale")}))


def pipeline_rows(n2n, synth):
    """The four synthetic queries of bench.py (rank 0) through the reference's Detector / ViewpointSelector /
    VolumeRefiner with gen6d_amd.pipeline.TensorPipeline's synthetic reference state (mirrors TensorPipeline.build/query;
    per iteration the refiner receives the pipeline's canned input pose, as in the bench)."""
    an, sel_rfn, det_rfn, iters = 5, 64, 32, 3
    det = n2n["detector"]({"network": "detector", "name": "g"}).eval()
    det.load_state_dict(synth.synth_state_dict("detector"))
    sel = n2n["selector"]({"network": "selector", "name": "g", "selector_angle_num": an}).eval()
    sel.load_state_dict(synth.synth_state_dict("selector", an=an))
    ref = n2n["refiner"]({"network": "refiner", "name": "g"}).eval()
    ref.load_state_dict(synth.synth_state_dict("refiner"))
    sel_case = synth.selector_case(sel_rfn, an, 1)
    det_refs = sel_case["ref_imgs"][an // 2, :det_rfn].contiguous()
    rc = synth.refiner_case()
    iter_poses = [torch.from_numpy(synth.perturb_pose(rc["poses_in"][0].numpy(), 2.0 * i, 0.01 * i))[None] for i in range(iters)]
    fulls = synth.imgs_to_tensor(synth.synth_images(4, 480, 640, seed=100))
    crops = synth.imgs_to_tensor(synth.synth_images(4, 128, 128, seed=200))
    rows, logits_all = [], []
    with torch.no_grad():
        det.load_impl(det_refs)
        sel.extract_ref_feats(sel_case["ref_imgs"], sel_case["ref_poses"], sel_case["object_center"], sel_case["object_vert"])
        for j in range(4):
            d = det.detect_impl(fulls[j:j + 1])
            pos, scl = det.parse_detection(d["scores"], d["select_pr_scale"], d["select_pr_offset"], 8)
            logits, angles = sel.compute_view_point_feats(crops[j:j + 1])
            idx = torch.argmax(logits, 1)
            ang = angles[torch.arange(1), idx]
            steps = []
            for p in iter_poses:
                o = ref({"que_imgs_info": {"imgs": crops[j:j + 1], "Ks_in": rc["Ks_in"], "poses_in": p},
                         "ref_imgs_info": {"imgs": rc["ref_imgs"], "Ks": rc["ref_Ks"], "poses": rc["ref_poses"]},
                         "inference": True})
                steps += [o["rotation"], o["offset"], o["scale"]]          # every step's outputs (round 4: was the last step only)
            rows.append(torch.cat([pos, scl[:, None], idx[:, None].float(), ang[:, None]] + steps, 1))
            logits_all.append(logits)
            print("query", j, rows[-1].numpy().round(4))
    save("pipeline_rows", rows=torch.cat(rows, 0).numpy(), logits=torch.cat(logits_all, 0).numpy(),
         sha_inputs=synth.fingerprint(sel_case, det_refs, rc, iter_poses, fulls, crops),
         sha_weights=synth.fingerprint([synth.synth_state_dict(k, an=an) for k in ("detector", "selector", "refiner")]),
         cfg=np.asarray([sel_rfn, det_rfn, an, iters]))


def main():
    import warnings
    warnings.filterwarnings("ignore")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    n2n = MG.load_reference()
    if not hasattr(np, "bool"): np.bool = bool
    from gen6d_amd import synth
    todo = sys.argv[1:] or ["operator", "ref_grids", "det_head", "sel_head", "sel_128x5", "sel_32x5", "sel_64x36", "pipeline_rows"]
    for name in todo:
        t0 = time.time()
        if name == "operator": operator_golden()
        elif name == "ref_grids": ref_grids(n2n, synth)
        elif name == "det_head": det_head(n2n, synth)
        elif name == "sel_head": _selector(n2n, synth, "sel_head", 64, 5)
        elif name == "sel_128x5": _selector(n2n, synth, "sel_128x5", 128, 5)
        elif name == "sel_32x5": _selector(n2n, synth, "sel_32x5", 32, 5)
        elif name == "sel_64x36": _selector(n2n, synth, "sel_64x36", 64, 36)
        elif name == "pipeline_rows": pipeline_rows(n2n, synth)
        else: raise SystemExit(f"unknown fixture {name}")
        print(f"  ({name}: {time.time() - t0:.1f}s)")


if __name__ == "__main__":
    main()
