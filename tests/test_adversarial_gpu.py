"""Network-level adversarial inputs (VERDICT r03 weak #4 / next #5d): the two places where the reference's
semantics are not smooth —

  * the detector's score normalisation clips at +-vgg_score_max (reference network/detector.py:226-230): with the level
    statistics shrunk so that most normalised scores saturate, the drop-in must clip exactly where the reference does;
  * arg-max ties: `torch.argmax` returns the FIRST maximal index (reference network/detector.py:84-95 over the flattened score
    map, network/selector.py:172-173 over the reference views) — the drop-in's decode kernel, the chain's selection kernel
    and the host path must do the same when scores / logits are EXACTLY equal.

Every case runs through the C ABI on the GPU against the CPU oracle (oracle/gen6d_oracle.py) on the same seeded inputs."""
import numpy as np
import pytest
import torch

from gen6d_amd import synth
from oracle import gen6d_oracle as O

pytestmark = pytest.mark.gpu


def _net(kind, sd, **cfg):
    from gen6d_amd import lib
    from gen6d_amd.network import name2network
    lib.load()
    net = name2network[kind]({"name": "t", **cfg}).eval()
    net.load_state_dict(sd)
    return net.cuda()


@pytest.mark.parametrize("spread,min_each", [(80.0, 0.25), (20.0, 0.01)])
def test_detector_scores_saturate_the_clip(monkeypatch, spread, min_each):
    """With the seeded random trunk the raw correlation scores sit far above the trained level statistics: every detector test already
    runs with ~99 % of the normalised scores clipped at +10 and none at -10.  Here the statistics are re-centred on the raw scores'
    per-level median with sigma = (q90 - q10) / spread: at spread 80 more than a quarter of the values sit at EACH bound (35 % / 30 %), at 20 fewer (15 % / 2 %)
    and most values lie in the linear range close to it.  Scores / offsets / scales stay within 1e-4 of the range of the
    oracle's and the detection cell is the same (reference network/detector.py:226-230)."""
    from parity_log import record
    sd = synth.synth_state_dict("detector")
    case = synth.detector_case(32, 160, 192)
    sd64 = O.to_double(sd)
    with torch.no_grad():
        rf32, rf64 = O.detector_ref_feats(sd, case["ref_imgs"]), O.detector_ref_feats(sd64, case["ref_imgs"].double())
        monkeypatch.setattr(O, "SCORE_STATS", [[0.0, 1e20]] * 3)                # raw scores (nothing clips at this sigma)
        raw = O.detector_detect(sd, case["que_imgs"], rf32, return_intermediates=True)["stacked"] * 1e20      # [1, 4 scales x 3 levels, rfn, hs, ws]
    stats = []
    for l in range(3):
        q = torch.quantile(raw[:, l::3].flatten()[::7], torch.tensor([0.1, 0.5, 0.9]))
        stats.append([float(q[1]), float(q[2] - q[0]) / spread])
    monkeypatch.setattr(O, "SCORE_STATS", stats)
    net = _net("detector", sd, vgg_score_stats=stats)
    with torch.no_grad():
        out = net({"ref_imgs_info": {"imgs": case["ref_imgs"].cuda()}, "que_imgs_info": {"imgs": case["que_imgs"].cuda()}})
        o32 = O.detector_detect(sd, case["que_imgs"], rf32, return_intermediates=True)
        o64 = O.detector_detect(sd64, case["que_imgs"].double(), rf64)
    st = o32["stacked"]
    hi, lo = float((st == 10).float().mean()), float((st == -10).float().mean())
    assert hi >= min_each and lo >= min_each, (hi, lo)
    for k in ("scores", "select_pr_offset", "select_pr_scale"):
        rng = max(float(o64[k].abs().max()), 1.0)
        e_new = float((out[k].cpu().double() - o64[k]).abs().max()) / rng
        e_ref = float((o32[k].double() - o64[k]).abs().max()) / rng
        record(f"test_detector_scores_saturate_the_clip[spread {spread:g}]", f"{k} ({100 * hi:.0f} % at +10, {100 * lo:.0f} % at -10)", e_new,
               max(1e-4, 1.5 * e_ref), e_ref, "relative to range")
        assert e_new <= max(1e-4, 1.5 * e_ref), (k, e_new, e_ref)
    assert torch.equal(out["que_select_id"].cpu(), o64["que_select_id"])


def test_detector_all_scores_tied_first_cell_wins():
    """score_predict's last conv zeroed: every cell of the score map is exactly the bias -> the flattened arg-max is cell 0
    (reference detector.py:84-95); the decode kernel must pick (0, 0) too and gather offset / scale THERE."""
    sd = synth.synth_state_dict("detector")
    sd = {k: v.clone() for k, v in sd.items()}
    sd["score_predict.4.weight"].zero_()
    sd["score_predict.4.bias"].fill_(0.25)
    case = synth.detector_case(8, 96, 128, )
    net = _net("detector", sd)
    with torch.no_grad():
        out = net({"ref_imgs_info": {"imgs": case["ref_imgs"].cuda()}, "que_imgs_info": {"imgs": case["que_imgs"].cuda()}})
        o32 = O.detector_detect(sd, case["que_imgs"], O.detector_ref_feats(sd, case["ref_imgs"]))
        pos, scl = O.detector_parse(o32)
    assert float(out["scores"].min()) == float(out["scores"].max()) == 0.25
    assert torch.equal(out["que_select_id"].cpu(), torch.zeros_like(o32["que_select_id"])) and int(o32["que_select_id"].abs().sum()) == 0
    np.testing.assert_allclose(out["positions"].cpu().numpy(), pos.numpy(), atol=1e-3)
    np.testing.assert_allclose(out["scales"].cpu().numpy(), scl.numpy(), rtol=1e-4)


def test_detector_two_way_tie_between_cells():
    """An exact TWO-way tie at the top of a real (non-constant) score map: the network's own score map with its maximum copied to a
    LATER and, in turn, to an EARLIER cell goes through the decode kernel with the network's offset / scale maps; the first of the two
    equal maxima in raster order must win (torch.argmax's rule, reference detector.py:84-95)."""
    from gen6d_amd import ops
    sd = synth.synth_state_dict("detector")
    case = synth.detector_case(8, 96, 128)
    net = _net("detector", sd)
    with torch.no_grad():
        out = net({"ref_imgs_info": {"imgs": case["ref_imgs"].cuda()}, "que_imgs_info": {"imgs": case["que_imgs"].cuda()}})
    qn, _, hs, ws = out["scores"].shape
    sc = out["scores"].reshape(qn, hs * ws).clone()
    best = int(sc[0].argmax())
    for other in ((best + 7) % (hs * ws), (best - 5) % (hs * ws)):
        s2 = sc.clone()
        s2[0, other] = s2[0, best]                                   # exact two-way tie
        want = min(best, other)
        o4 = torch.cat([s2.reshape(-1, 1), out["select_pr_scale"].reshape(-1, 1),
                        out["select_pr_offset"].permute(0, 2, 3, 1).reshape(-1, 2)], 1).contiguous()
        res = ops.detector_decode(o4[:, 0:1], o4[:, 2:4], o4[:, 1:2], hs, ws, 8, batch=qn).view(qn, 5).cpu()
        assert int(torch.argmax(s2[0])) == want                      # the reference's rule (torch.argmax: first maximal index)
        assert (int(res[0, 3].round()), int(res[0, 4].round())) == (want % ws, want // ws)


@pytest.mark.parametrize("rfn,an", [(8, 5), (64, 5)])
def test_selector_tied_logits_first_reference_wins(rfn, an):
    """(a) score_predict's last conv zeroed: all rfn logits are exactly the bias -> reference 0 is selected (selector.py:172-173),
    by the numpy API and by the device chain's selection kernel; (b) the angle returned is the one predicted FOR reference 0."""
    from gen6d_amd import ops
    sd = {k: v.clone() for k, v in synth.synth_state_dict("selector", an=an).items()}
    sd["score_predict.2.weight"].zero_()
    sd["score_predict.2.bias"].fill_(-0.5)
    case = synth.selector_case(rfn, an)
    net = _net("selector", sd, selector_angle_num=an)
    with torch.no_grad():
        net.extract_ref_feats(case["ref_imgs"].cuda(), case["ref_poses"].cuda(), case["object_center"].cuda(), case["object_vert"].cuda())
        logits, angles = net.compute_view_point_feats(case["que_imgs"].cuda())
        cache, embed = O.selector_ref_state(sd, case["ref_imgs"], case["ref_poses"], case["object_center"], case["object_vert"])
        l32, a32 = O.selector_forward(sd, case["que_imgs"], cache, embed)
        idx_o, ang_o = O.selector_select(l32, a32)
    assert float(logits.min()) == float(logits.max()) == -0.5 and int(idx_o[0]) == 0
    assert int(torch.argmax(logits, 1)[0]) == 0
    np.testing.assert_allclose(float(angles[0, 0]), float(ang_o[0]), atol=2e-4)
    # the chain's selection kernel (arg-max + pose from detection and selection on the device): same rule
    det = torch.tensor([[64.0, 64.0, 1.0, 8.0, 8.0]], device="cuda")
    poses = case["ref_poses"].float().reshape(rfn, 12).cuda().contiguous()
    Ks = torch.eye(3, device="cuda").reshape(1, 9).repeat(rfn, 1).contiguous()
    K = torch.eye(3, device="cuda").reshape(1, 9).contiguous()
    res = ops.chain_pose_from_selection(det, logits.contiguous(), angles.contiguous(), poses, Ks, K, torch.zeros(3, device="cuda"))
    sel = res[1] if isinstance(res, (tuple, list)) else res
    assert int(sel.reshape(-1)[0].round()) == 0


def test_selector_duplicate_views_tie_at_the_top():
    """Every reference view appears twice (views 2k and 2k+1 are the same image with the same pose): whatever wins, its twin ties
    with it wherever the per-view arithmetic is identical — the even (first) index must be selected, as the oracle's arg-max does."""
    an, half = 5, 8
    sd = synth.synth_state_dict("selector", an=an)
    base = synth.selector_case(half, an)
    dup = lambda t, dim: torch.repeat_interleave(t, 2, dim=dim)
    case = dict(base, ref_imgs=dup(base["ref_imgs"], 1), ref_poses=dup(base["ref_poses"], 0))
    net = _net("selector", sd, selector_angle_num=an)
    with torch.no_grad():
        net.extract_ref_feats(case["ref_imgs"].cuda(), case["ref_poses"].cuda(), case["object_center"].cuda(), case["object_vert"].cuda())
        logits, angles = net.compute_view_point_feats(case["que_imgs"].cuda())
        cache, embed = O.selector_ref_state(sd, case["ref_imgs"], case["ref_poses"], case["object_center"], case["object_vert"])
        l32, a32 = O.selector_forward(sd, case["que_imgs"], cache, embed)
    lg = logits[0].cpu()
    top_o = int(l32[0].argmax())
    assert abs(float(l32[0, top_o] - l32[0, top_o ^ 1])) <= 1e-5          # the oracle's twins agree (to its own rounding)
    assert float((lg - l32[0]).abs().max()) <= 2e-4
    top = int(lg.argmax())
    assert top // 2 == top_o // 2, (top, top_o)                             # the same pair wins
    if float(lg[top]) == float(lg[top ^ 1]):                                # bit-equal twins (the usual case): first index wins
        assert top % 2 == 0
