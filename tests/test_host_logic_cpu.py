"""Host orchestration of gen6d_amd/network/* checked on CPU: gen6d_amd.ops is monkeypatched with the per-op PyTorch
references (tests/ref_ops.py), so layouts, weight repacking, InstanceNorm fusion, buffer slicing and the commuted
pooling are validated against the CPU oracle and the reference-generated golden vectors without a GPU.  The HIP
kernels themselves are compared with the same per-op references in the `-m gpu` tests."""
import numpy as np
import pytest
import torch

import ref_ops
from gen6d_amd import specs, synth
from gen6d_amd.network import name2network
from oracle import gen6d_oracle as O


@pytest.fixture(autouse=True)
def _patch(monkeypatch):
    ref_ops.patch_ops(monkeypatch)


def test_state_dict_contract():
    for kind, n in (("detector", 78), ("selector", 132), ("refiner", 104)):
        net = name2network[kind]({})
        sd = synth.synth_state_dict(kind)
        assert len(sd) == n                                   # SURVEY.md App. C key counts
        missing, unexpected = net.load_state_dict(sd, strict=True)
        assert not missing and not unexpected
        assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}


def test_detector_host_path(golden):
    g = golden("det_small")
    net = name2network["detector"]({"name": "t"}).eval()
    net.load_state_dict(synth.synth_state_dict("detector"))
    case = synth.detector_case(int(g["rfn"]), int(g["hq"]), int(g["wq"]))
    with torch.no_grad():
        out = net({"ref_imgs_info": {"imgs": case["ref_imgs"]}, "que_imgs_info": {"imgs": case["que_imgs"]}})
    for k in ("scores", "select_pr_offset", "select_pr_scale"):
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-3, atol=1e-3 * np.abs(g[k]).max())
    assert np.array_equal(out["que_select_id"].numpy(), g["que_select_id"])
    np.testing.assert_allclose(out["positions"].numpy(), g["positions"], rtol=1e-3, atol=5e-2)
    np.testing.assert_allclose(out["scales"].numpy(), g["scales"], rtol=2e-3)


def test_selector_host_path(golden):
    g = golden("sel_small")
    an = int(g["an"])
    net = name2network["selector"]({"name": "t", "selector_angle_num": an}).eval()
    net.load_state_dict(synth.synth_state_dict("selector", an=an))
    case = synth.selector_case(int(g["rfn"]), an)
    with torch.no_grad():
        out = net({"ref_imgs": case["ref_imgs"], "ref_imgs_info": {"poses": case["ref_poses"]},
                   "object_center": case["object_center"], "object_vert": case["object_vert"],
                   "que_imgs_info": {"imgs": case["que_imgs"]}, "eval": True})
    np.testing.assert_allclose(out["ref_vp_logits"].numpy(), g["logits"], atol=2e-3)
    np.testing.assert_allclose(out["angles_pr"].numpy(), g["angles"], atol=2e-3)
    assert np.array_equal(out["ref_vp_logits"].argmax(1).numpy(), g["logits"].argmax(1))


def test_refiner_host_path(golden):
    g = golden("ref_step")
    net = name2network["refiner"]({"name": "t"}).eval()
    net.load_state_dict(synth.synth_state_dict("refiner"))
    c = synth.refiner_case()
    with torch.no_grad():
        out = net({"que_imgs_info": {"imgs": c["que_imgs"], "Ks_in": c["Ks_in"], "poses_in": c["poses_in"]},
                   "ref_imgs_info": {"imgs": c["ref_imgs"], "Ks": c["ref_Ks"], "poses": c["ref_poses"]},
                   "inference": True})
    for k in ("rotation", "offset", "scale"):
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-3, atol=1e-3)


def test_tensor_pipeline_matches_oracle_pipeline():
    """detect -> select -> refine x2 through TensorPipeline (emulated ops) vs oracle/pipeline_oracle.py."""
    from gen6d_amd.pipeline import TensorPipeline
    from oracle import pipeline_oracle as PO
    pipe = TensorPipeline("cpu", sel_rfn=8, det_rfn=8, refine_iter=2)
    pipe.build()
    full = synth.imgs_to_tensor(synth.synth_images(1, 96, 128, seed=100))
    crop = synth.imgs_to_tensor(synth.synth_images(1, 128, 128, seed=200))
    row = pipe.query(full, crop)
    st = PO.build_state(pipe.state_dicts, pipe.det_refs, pipe.sel_case)
    ref, logits = PO.query(pipe.state_dicts, st, pipe.ref_case, [p for p in pipe.iter_poses], full, crop)
    assert row.shape == (1, 5 + 7 * 2)                 # every refine step in the row
    assert int(row[0, 3]) == int(ref[0, 3])
    np.testing.assert_allclose(row.numpy(), ref.numpy(), rtol=2e-3, atol=2e-2)


def test_batched_queries_equal_single_queries():
    """qn = 3 queries through ONE set of launches (the query-batch path of every network: shared reference cache through
    `in_mod`, per-query multiplier maps, per-query InstanceNorm groups / tables, batched tail and volume ops) give the rows of the
    three single-query calls — on the CPU emulation of the ops, i.e. this checks the host orchestration of the batch."""
    from gen6d_amd.pipeline import TensorPipeline
    pipe = TensorPipeline("cpu", sel_rfn=6, det_rfn=6, refine_iter=1)
    pipe.build()
    fulls = synth.imgs_to_tensor(synth.synth_images(3, 64, 96, seed=100))
    crops = synth.imgs_to_tensor(synth.synth_images(3, 128, 128, seed=200))
    rows = pipe.query(fulls, crops)
    assert rows.shape == (3, 5 + 7 * 1)
    for j in range(3):
        one = pipe.query(fulls[j:j + 1], crops[j:j + 1])
        assert int(rows[j, 3]) == int(one[0, 3])
        np.testing.assert_allclose(rows[j].numpy(), one[0].numpy(), rtol=1e-4, atol=1e-4)
    # and the per-network [qn,...] contracts
    with torch.no_grad():
        logits, angles = pipe.selector.compute_view_point_feats(crops)
        l1, a1 = pipe.selector.compute_view_point_feats(crops[1:2])
        det = pipe.detector.detect_impl(fulls)
        d1 = pipe.detector.detect_impl(fulls[2:3])
    assert logits.shape == (3, 6) and angles.shape == (3, 6) and det["scores"].shape == (3, 1, 8, 12)
    np.testing.assert_allclose(logits[1].numpy(), l1[0].numpy(), atol=1e-4)
    np.testing.assert_allclose(angles[1].numpy(), a1[0].numpy(), atol=1e-4)
    np.testing.assert_allclose(det["scores"][2].numpy(), d1["scores"][0].numpy(), atol=1e-4 * float(d1["scores"].abs().max()))
    assert torch.equal(det["que_select_id"][2], d1["que_select_id"][0])


def test_detector_winograd_correlation_host_path(golden):
    """32 reference views: the 15x15 correlation level takes the Winograd route (block-cut, transformed reference filters built at
    load time) — emulated on the CPU by the block-wise Winograd algorithm on those filters — and must reproduce the reference's
    golden detection (det_mid: 160x192 query vs 32 references)."""
    g = golden("det_mid")
    net = name2network["detector"]({"name": "t"}).eval()
    net.load_state_dict(synth.synth_state_dict("detector"))
    case = synth.detector_case(int(g["rfn"]), int(g["hq"]), int(g["wq"]))
    with torch.no_grad():
        out = net({"ref_imgs_info": {"imgs": case["ref_imgs"]}, "que_imgs_info": {"imgs": case["que_imgs"]}})
    assert net.ref_wino15 is None and tuple(net.ref_wino15_43.shape) == (25 * 64, 2, 1, 18, 1, 4, 16, 4)      # F(4x4,3x3) route (detector.F43)
    for k in ("scores", "select_pr_offset", "select_pr_scale"):
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-3, atol=1e-3 * np.abs(g[k]).max())
    assert np.array_equal(out["que_select_id"].numpy(), g["que_select_id"])


def test_more_queries_than_one_launch_batch(monkeypatch):
    """qn = 9 > MAX_BATCH (set to 8 here; 32 in the product): the networks cut the call into chunks of <= MAX_BATCH queries that share
    a set of launches each; the [qn, ...] results are those of the single-query calls (selector and refiner forward; CPU emulation
    of the ops)."""
    from gen6d_amd.network import refiner as refiner_mod, selector as selector_mod
    monkeypatch.setattr(selector_mod, "MAX_BATCH", 8)
    monkeypatch.setattr(refiner_mod, "MAX_BATCH", 2)
    an, rfn = 5, 4
    sel = name2network["selector"]({"name": "t", "selector_angle_num": an}).eval()
    sel.load_state_dict(synth.synth_state_dict("selector", an=an))
    case = synth.selector_case(rfn, an)
    ques = synth.imgs_to_tensor(synth.synth_images(9, 128, 128, seed=31))
    with torch.no_grad():
        sel.extract_ref_feats(case["ref_imgs"], case["ref_poses"], case["object_center"], case["object_vert"])
        logits, angles = sel.compute_view_point_feats(ques)
        assert logits.shape == (9, rfn) and angles.shape == (9, rfn)
        for j in (0, 7, 8):                                          # last of the first chunk, the lone query of the second
            l1, a1 = sel.compute_view_point_feats(ques[j:j + 1])
            np.testing.assert_allclose(logits[j].numpy(), l1[0].numpy(), atol=1e-4)
            np.testing.assert_allclose(angles[j].numpy(), a1[0].numpy(), atol=1e-4)
    ref = name2network["refiner"]({"name": "t"}).eval()
    ref.load_state_dict(synth.synth_state_dict("refiner"))
    c = synth.refiner_case()
    n = 3
    data = {"que_imgs_info": {"imgs": synth.imgs_to_tensor(synth.synth_images(n, 128, 128, seed=41)), "Ks_in": c["Ks_in"].expand(n, 3, 3),
                              "poses_in": torch.stack([torch.from_numpy(synth.perturb_pose(c["poses_in"][0].numpy(), 1.0 * i, 0.01 * i)) for i in range(n)])},
            "ref_imgs_info": {"imgs": c["ref_imgs"].expand(n, *c["ref_imgs"].shape[1:]), "Ks": c["ref_Ks"].expand(n, 6, 3, 3),
                              "poses": c["ref_poses"].expand(n, 6, 3, 4)}, "inference": True}
    with torch.no_grad():
        out = ref(data)
        assert out["rotation"].shape == (n, 4) and out["offset"].shape == (n, 2) and out["scale"].shape == (n, 1)
        one = ref({"que_imgs_info": {k: v[1:2] for k, v in data["que_imgs_info"].items()},
                   "ref_imgs_info": {k: v[1:2] for k, v in data["ref_imgs_info"].items()}, "inference": True})
    for k in ("rotation", "offset", "scale"):
        np.testing.assert_allclose(out[k][1].numpy(), one[k][0].numpy(), atol=2e-4)


def test_detector_chunk_follows_the_image_size(monkeypatch):
    """the queries of a detector chunk share launches whose 32-bit offsets reach 2^29 floats from one base — the chunk is
    derived from the image size (16 queries of 480x640, 9 of 720x1280), and an image whose pyramid exceeds the reach by itself runs
    one trunk pass per scale instead of failing with G6D_EINVAL."""
    det = name2network["detector"]({"name": "t"}).eval()
    calls = []

    def fake_batch(que, multi=True):
        qn, _, hq, wq = que.shape
        calls.append((qn, multi))
        hs, ws = hq // 8, wq // 8
        return torch.zeros(qn, hs, ws, 4), torch.zeros(qn, 5), (hs, ws)
    monkeypatch.setattr(det, "_detect_batch", fake_batch)
    for (h, w, n), want in (((480, 640, 20), [(16, True), (4, True)]), ((720, 1280, 11), [(9, True), (2, True)]),
                            ((2160, 3840, 2), [(1, True), (1, True)])):
        calls.clear()
        out = det._detect_impl_fp(torch.zeros(n, 3, h, w).expand(n, 3, h, w))
        assert calls == want, (h, w, calls)
        assert out["scores"].shape == (n, 1, h // 8, w // 8) and out["positions"].shape == (n, 2)
    # (an image of 4000 x 6000 pixels: the pyramid alone is 1.4e9 floats at 64 channels -> per-scale trunk passes)
    per_query = sum(det._scale_size(4000, 6000, s)[0] * det._scale_size(4000, 6000, s)[1] for s in det.cfg["detection_scales"]) // 4 * 64
    assert per_query >= (1 << 29)
    assert det._scale_size(480, 640, 0.5) == (704, 928) and det._scale_size(480, 640, -1.0) == (256, 320)      # reference detector.py:237-239


def test_padded_correlation_filters_for_the_7x7_level():
    """backbone.winograd43_corr_filters_padded: the 7x7 correlation filters zero-extended to 9x9 and cut into 3x3 blocks of 3x3 reproduce
    the direct 7x7 "same" correlation when accumulated block-wise in the F(4x4,3x3) domain (the algorithm of g6d_corr2d_wino43_multi with
    kblocks = 3, emulated in float64 on the filters the kernel receives)."""
    from gen6d_amd.network.backbone import winograd43_corr_filters_padded
    g = torch.Generator().manual_seed(5)
    w = (torch.rand((32, 49, 16), generator=g) * 2 - 1).double()
    x = (torch.rand((2, 1, 11, 13, 16), generator=g) * 2 - 1).double()
    U, kb = winograd43_corr_filters_padded(w, 7)
    assert kb == 3 and tuple(U.shape) == (2 * 9, 2, 1, 18, 1, 4, 16, 4)
    out = torch.empty((2, 1, 11, 13, 32), dtype=torch.float64)
    ref_ops.corr2d_wino43_multi([x], U, [out], kb, k_true=7)
    ref = torch.empty_like(out)
    ref_ops.corr2d_patch(x, w, ref, 7)
    assert float((out - ref).abs().max()) < 1e-12
    with pytest.raises(ValueError):
        winograd43_corr_filters_padded(torch.zeros((32, 64, 16), dtype=torch.float64), 8)        # 8 -> 9 cannot be centred
