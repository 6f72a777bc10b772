"""Detector — drop-in for the reference's network/detector.py:Detector (same constructor, state_dict keys, methods and
output dicts), driving hand-written HIP kernels end to end.

For the 4 detection scales of a batch of queries at once (reference detect_impl, detector.py:232-266):
    own Winograd trunk              F(4x4,3x3) on fp32 MFMA, one launch per layer over the whole pyramid -> x0 @1/8, x1 @1/16, x2 @1/32
    correlation x3                  query features correlated with the reference feature maps used as filters
                                    (F.conv2d(que_x, ref_x, padding=7/3/1), detector.py:222-224): the 15x15 level as 5x5 blocks of 3x3
                                    and the 7x7 level as 3x3 blocks (filters zero-extended to 9x9) in the F(4x4,3x3) domain
                                    (g6d_corr2d_wino43_multi), the 3x3 level on g6d_corr2d_patch_multi
    g6d_detector_assemble           nearest up-sampling, (x-mu)/sigma, clip, bilinear resize to (h/8,w/8), stack
then g6d_detector_score_mlp_max     score_conv MLP + max over references, never materialising [64,rfn,hs,ws]
     g6d_conv_igemm x7              the three 3x3 heads (first layers merged into one 64->192 conv)
     g6d_detector_decode            arg-max + offset/scale gather.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import ops, parallel, specs
from .backbone import (pack_trunk, trunk_features, trunk_features_multi, winograd_corr_filters, winograd43_corr_filters,
                       winograd43_corr_filters_padded)
from .params import ParamBank, fold_vgg

# Launch-structure switches; the product runs with all of them True, tools / tests flip the attribute for A/B runs:
TRUNK_MULTI = True       # one launch per trunk layer over all pyramid scales (False: one trunk pass per scale)
CORR3_MULTI = True       # the 3x3 correlation level as one corr_patch launch over all scales (False: one generic conv launch per scale)
CORR_WINO = True         # the 15x15 correlation level in the Winograd domain, 5x5 blocks of 3x3, when rfn % 32 == 0 and fp32 (False: corr_patch)
# Winograd F(4x4,3x3) (csrc/wino43_conv.hip) for the query pyramid's trunk and the 15x15 correlation level: 1.78x fewer fp32
# multiplications than F(2x2,3x3) at ~5x its rounding error — the detector holds ~4e-6 of the score range against the 1e-4 bar
# (tests/test_parity_timed_gpu.py).  False: the F(2x2,3x3) kernels of round 3 (tools/ A/B runs and tests flip this attribute).
F43 = True
CORR16 = True            # round 6: the 15x15 correlation level on the halo-patch kernel of the 16-bit matrix cores (g6d_corr16_multi) whenever the
                         # trunk hands its input over as 16-bit activations (reduced precision) or fp16 hi / lo pairs (fp32 path: fp32-class
                         # results); False (tools / tests): the F(4x4,3x3) / corr16_patch kernels of rounds 4-5
CORR16_7 = True          # ... and the 7x7 level (its input c7_pre in the same format)
CORR7_F43 = True         # the 7x7 level as 3x3 blocks of 3x3 on zero-extended 9x9 filters in the F(4x4,3x3) domain (20.25 instead of 49
                         # multiplications per output) when rfn % 32 == 0 and fp32; False: corr_patch
MAX_BATCH = 16       # most queries that share one set of launches; _detect_impl_fp cuts the chunk further for larger images (the pyramid's
                     # first layers address all scales of the batch with 32-bit offsets from one base, < 2^29 floats: 16 images of
                     # 480x640 at 64 channels; 32 would not fit)


class Detector(ParamBank):
    default_cfg = {
        "vgg_score_stats": [[36.264317, 13.151907], [13910.291, 5345.965], [829.70807, 387.98788]],
        "vgg_score_max": 10,
        "detection_scales": [-1.0, -0.5, 0.0, 0.5],
        "train_feats": False,
    }

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}
        super().__init__(specs.detector_rows())
        if len(self.cfg["detection_scales"]) != 4:
            raise NotImplementedError("score_conv expects 3 levels x 4 detection scales (12 channels)")
        self.pool_ratio = 8
        self.ref_center_feats = None     # three [rfn, k*k, 512] correlation filters
        self.ref_wino15 = None           # the 15x15 level's filters in the Winograd domain (winograd_corr_filters), F(2x2,3x3)
        self.ref_wino15_43 = None        # ... for the F(4x4,3x3) kernel (winograd43_corr_filters); one of the two is built
        self.ref_wino7_43 = None         # the 7x7 level's filters, zero-extended to 9x9, for the F(4x4,3x3) kernel
        self.ref_shape = None
        self.rank, self.world, self.group = 0, 1, None
        self.sharded = False

    def set_shard(self, rank, world, group=None, force_collectives=False):
        """Reference-sharded mode: this rank correlates the query against references
        parallel.shard_range(rfn, rank, world) only; correlation, score assembly and the score MLP are per-reference, and
        `torch.max(scores, 2)` over the references (detector.py:247) becomes one all-reduce(MAX) of the [hs*ws, 64]
        feature map (1.2 MB at 60x80) over torch.distributed (RCCL).  The query trunk and the heads are replicated.
        `force_collectives`: issue the all-reduce at world size 1 as well (see ViewpointSelector.set_shard)."""
        self.rank, self.world, self.group = int(rank), int(world), group
        self.sharded = self.world > 1 or bool(force_collectives)

    # ------------------------------------------------------------------ weights
    def _pack(self):
        if self._packed is None:
            pk = {"vgg": pack_trunk(fold_vgg(self, "backbone.features"))}
            w0, b0 = self.conv_w("score_conv.0")
            w1, b1 = self.conv_w("score_conv.2")
            pk["mlp"] = (w0.reshape(64, 12).contiguous(), b0, w1.reshape(64, 64).contiguous(), b1)
            heads = ("score_predict", "scale_predict", "offset_predict")
            l0 = [self.conv_w(f"{h}.0") for h in heads]
            pk["h0"] = (torch.cat([w for w, _ in l0], 0).contiguous(), torch.cat([b for _, b in l0], 0).contiguous())
            pk["h1"] = [self.conv_w(f"{h}.2") for h in heads]
            # the three heads' last convs (1 + 1 + 2 output channels on their own 64-channel inputs) as ONE conv 192 -> 4 with
            # block-diagonal weights: the zero blocks add exact zeros, one launch instead of three latency-bound ones
            l2 = [self.conv_w(f"{h}.4") for h in heads]
            w4 = torch.zeros((sum(w.shape[0] for w, _ in l2), 9, 192), dtype=torch.float32, device=l2[0][0].device)
            row = 0
            for i, (w, _) in enumerate(l2):
                w4[row:row + w.shape[0], :, 64 * i:64 * i + 64] = w
                row += w.shape[0]
            pk["h2"] = (w4.contiguous(), torch.cat([b for _, b in l2], 0).contiguous())
            self._packed = pk
        return self._packed

    # ------------------------------------------------------------------ trunk
    def extract_feats(self, imgs):
        """imgs [n,3,h,w] in [0,1] -> channels-last x0,x1,x2: [n,1,h/8,w/8,512], [.. /16 ..], [.. /32 ..]."""
        return trunk_features(self._pack()["vgg"], imgs, ("c5", "c7_pre", "p7"), False)

    def load_impl(self, ref_imgs):
        """ref_imgs [rfn,3,h,w] in [0,1]; nearest resize to 120x120, trunk, keep as correlation filters
        (reference detector.py:199-205)."""
        b, e = parallel.shard_range(ref_imgs.shape[0], self.rank, self.world)
        if e == b:
            raise ValueError("more ranks than reference views")
        ref_imgs = F.interpolate(ref_imgs[b:e], size=(120, 120))
        feats = self.extract_feats(ref_imgs)                       # [rfn_local,1,k,k,512]
        self.ref_center_feats = [f.reshape(f.shape[0], f.shape[2] * f.shape[3], 512).contiguous() for f in feats]
        self.ref_ksize = [f.shape[2] for f in feats]               # 15, 7, 3
        self._corr16 = {}                                          # (packed filters of g6d_corr16_multi, built on first use)
        self.ref_shape = [120, 120]
        # Winograd-domain filters of the 15x15 level (the reference views used as filters: transformed once per object)
        rfn = self.ref_center_feats[0].shape[0]
        ok15 = CORR_WINO and self.ref_ksize[0] == 15 and rfn % 32 == 0
        self.ref_wino15 = winograd_corr_filters(self.ref_center_feats[0], 15) if (ok15 and not F43) else None
        self.ref_wino15_43 = winograd43_corr_filters(self.ref_center_feats[0], 15) if (ok15 and F43) else None
        ok7 = CORR_WINO and F43 and CORR7_F43 and self.ref_ksize[1] == 7 and rfn % 32 == 0
        self.ref_wino7_43 = winograd43_corr_filters_padded(self.ref_center_feats[1], 7)[0] if ok7 else None

    def _corr16_filters(self, level, mode):
        """The reference-centre features of a level packed for g6d_corr16_multi (ops.corr16_pack), built on first use per (level, mode)."""
        cache = self.__dict__.setdefault("_corr16", {})
        key = (level, mode, self.ref_center_feats[level].data_ptr())
        if key not in cache:
            cache[key] = ops.corr16_pack(self.ref_center_feats[level], mode)
        return cache[key]

    # ------------------------------------------------------------------ detection
    def _scores_one_scale(self, que_img, scale_idx, stacked, hs, ws):
        self._scores_from_feats(self.extract_feats(que_img), scale_idx, stacked, hs, ws)

    def _scores_from_pyramid(self, feats, scale_ids, stacked, hs, ws):
        """All scales (and all queries of the batch) at once: every correlation level is ONE launch over the maps of all scales
        (the tiles of all maps form one work list: fewer splits, the small maps fill the chip the large one leaves over); the
        assembly is one launch per scale.  feats[i][l]: [qn,1,h,w,512]."""
        rfn = self.ref_center_feats[0].shape[0]
        dev = stacked.device
        qn = feats[0][0].shape[0]
        maps = [[None] * 3 for _ in feats]
        for l, (wref, k) in enumerate(zip(self.ref_center_feats, self.ref_ksize)):
            xs = [f[l] for f in feats]
            if xs[0].dtype != torch.float32:
                # the trunk handed this level over as 16-bit activations (reduced precision) or fp16 hi / lo pairs (fp32 path): the halo-patch
                # correlation kernel on the 16-bit matrix cores (csrc/conv16_direct.hip, corr16_kernel)
                mode = 3 if xs[0].dim() == 5 else (1 if xs[0].dtype == torch.bfloat16 else 2)
                filt = self._corr16_filters(l, mode)
                outs = [torch.empty((qn, 1, x.shape[1], x.shape[2], rfn), dtype=torch.float32, device=dev) for x in xs]
                ops.corr16_multi(xs, filt, outs)
                for i, o in enumerate(outs):
                    maps[i][l] = o.reshape(qn * o.shape[2] * o.shape[3], rfn)
                continue
            # one launch over all scales addresses the maps with 32-bit offsets from a common base (offset + extent < 2^29 floats on the
            # Winograd routes): maps that do not come out of one buffer (a trunk that allocates every scale on its own) are gathered first
            lo = min(x.data_ptr() for x in xs)
            hi = max(x.data_ptr() + x.numel() * 4 for x in xs)
            if hi - lo >= (1 << 31) or not all(x.is_contiguous() for x in xs):
                seg = ops.alloc_like_segments([tuple(x.shape) for x in xs], dev)
                for d_, x in zip(seg, xs):
                    d_.copy_(x)
                xs = seg
            if k == 15 and self.ref_wino15_43 is not None and ops.MATH_MODE == 0 and len(xs) <= 4:
                outs = ops.alloc_like_segments([(qn, 1, x.shape[2], x.shape[3], rfn) for x in xs], dev)
                ops.corr2d_wino43_multi([x.contiguous() for x in xs], self.ref_wino15_43, outs, 5)
            elif k == 7 and self.ref_wino7_43 is not None and ops.MATH_MODE == 0 and len(xs) <= 4:
                outs = ops.alloc_like_segments([(qn, 1, x.shape[2], x.shape[3], rfn) for x in xs], dev)
                ops.corr2d_wino43_multi([x.contiguous() for x in xs], self.ref_wino7_43, outs, 3, k_true=7)
            elif k == 15 and self.ref_wino15 is not None and ops.MATH_MODE == 0 and len(xs) <= 4:
                outs = ops.alloc_like_segments([(qn, 1, x.shape[2], x.shape[3], rfn) for x in xs], dev)
                ops.corr2d_wino_multi([x.contiguous() for x in xs], self.ref_wino15, outs, 5)
            elif rfn <= 32 and len(xs) <= 4 and (k >= 7 or CORR3_MULTI):
                outs = ops.alloc_like_segments([(qn, 1, x.shape[2], x.shape[3], rfn) for x in xs], dev)
                ops.corr2d_patch_multi(xs, wref, outs, k)
            else:
                outs = [torch.empty((qn, 1, x.shape[2], x.shape[3], rfn), dtype=torch.float32, device=dev) for x in xs]
                for x, o in zip(xs, outs):
                    if k >= 7 and rfn <= 32 and qn == 1:
                        ops.corr2d_patch(x, wref, o, k)
                    else:
                        ops.conv(x, wref, None, o, ksize=(1, k, k), pad=(0, k // 2, k // 2))
            for i, o in enumerate(outs):
                maps[i][l] = o.reshape(qn * o.shape[2] * o.shape[3], rfn)
        for f, si, m in zip(feats, scale_ids, maps):
            hc, wc = (f[0].shape[2], f[0].shape[3]) if f[0].dtype == torch.float32 else (f[0].shape[1], f[0].shape[2])
            ops.detector_assemble(m[0], m[1], m[2], hc, wc, self.cfg["vgg_score_stats"],
                                  float(self.cfg["vgg_score_max"]), hs, ws, si, stacked, batch=qn)

    def _scores_from_feats(self, feats, scale_idx, stacked, hs, ws):
        x0, x1, x2 = feats
        rfn = self.ref_center_feats[0].shape[0]
        qn = x0.shape[0]
        maps = []
        for x, wref, k in zip((x0, x1, x2), self.ref_center_feats, self.ref_ksize):
            _, _, h, w, _ = x.shape
            o = torch.empty((qn, 1, h, w, rfn), dtype=torch.float32, device=x.device)
            if k >= 7 and rfn <= 32 and qn == 1:       # 15x15 and 7x7 levels: input patch kept in LDS and walked by the kx taps
                ops.corr2d_patch(x, wref, o, k)
            else:
                ops.conv(x, wref, None, o, ksize=(1, k, k), pad=(0, k // 2, k // 2))
            maps.append(o.reshape(qn * h * w, rfn))
        hc, wc = x0.shape[2], x0.shape[3]
        ops.detector_assemble(maps[0], maps[1], maps[2], hc, wc, self.cfg["vgg_score_stats"],
                              float(self.cfg["vgg_score_max"]), hs, ws, scale_idx, stacked, batch=qn)

    @staticmethod
    def _scale_size(hq, wq, scale):
        """Size of the query at a detection scale: round(h * 2^s), rounded UP to a multiple of 32 (reference detector.py:237-239)."""
        ht, wt = int(np.round(hq * 2 ** scale)), int(np.round(wq * 2 ** scale))
        if ht % 32 != 0: ht = (ht // 32 + 1) * 32
        if wt % 32 != 0: wt = (wt // 32 + 1) * 32
        return ht, wt

    def _detect_batch(self, que_imgs, multi=True):
        """que_imgs [qn,3,hq,wq]: the whole batch goes through every launch together (trunk pyramid segments of qn images,
        correlation tiles of qn maps per scale, heads with M = qn*hs*ws) — reference API: detector.py:291-304 takes [qn,H,W,3]."""
        pk = self._pack()
        qn, _, hq, wq = que_imgs.shape
        hs, ws = hq // 8, wq // 8
        dev = que_imgs.device
        rfn = self.ref_center_feats[0].shape[0]
        P = hs * ws
        stacked = torch.empty((qn * P, rfn, 12), dtype=torch.float32, device=dev)
        def resized(scale):
            return F.interpolate(que_imgs, size=self._scale_size(hq, wq, scale), mode="bilinear")

        # the scales are independent until `stacked` is complete: largest first on the main stream
        order = sorted(enumerate(self.cfg["detection_scales"]), key=lambda t: -t[1])
        if TRUNK_MULTI and multi and len(order) <= 4:
            # every trunk layer is ONE launch over the whole pyramid (the small scales fill the blocks the large ones leave
            # over); the correlations of the scales then run side by side
            # the image pyramid in ONE launch (g6d_resize_bilinear_pyramid; the scale of the query's own size is the query itself)
            pyr = ops.resize_bilinear_pyramid(que_imgs, [self._scale_size(hq, wq, sc) for _, sc in order])
            # (CORR16: the 15x15 level's input in the trunk kernel's 16-bit / pair format -> g6d_corr16_multi)
            t16 = ()
            if CORR16 and self.ref_center_feats[0].shape[0] == 32:
                t16 = (("c5",) if self.ref_ksize[0] == 15 else ()) + (("c7_pre",) if (CORR16_7 and self.ref_ksize[1] == 7) else ())
            feats = trunk_features_multi(pk["vgg"], pyr, ("c5", "c7_pre", "p7"), f43=F43, taps16=t16)
            self._scores_from_pyramid(feats, [si for si, _ in order], stacked, hs, ws)
        else:
            ops.fork_join([(lambda si=si, sc=sc: self._scores_one_scale(resized(sc), si, stacked, hs, ws)) for si, sc in order], dev)
        feats = ops.detector_score_mlp_max(stacked, *pk["mlp"])               # [qn*P,64], max over the local references
        if self.sharded:
            parallel.all_reduce_(feats, "max", self.group)
        k3, p3 = (1, 3, 3), (0, 1, 1)
        a = torch.empty((qn, 1, hs, ws, 192), dtype=torch.float32, device=dev)
        ops.conv(feats.view(qn, 1, hs, ws, 64), pk["h0"][0], pk["h0"][1], a, ksize=k3, pad=p3, out_act=1)
        b = torch.empty_like(a)
        o4 = torch.empty((qn, 1, hs, ws, 4), dtype=torch.float32, device=dev)
        for i in range(3):
            w1, b1 = pk["h1"][i]
            ops.conv(a[..., 64 * i:64 * i + 64], w1, b1, b[..., 64 * i:64 * i + 64], ksize=k3, pad=p3, out_act=1)
        ops.conv(b, pk["h2"][0], pk["h2"][1], o4, ksize=k3, pad=p3)            # score | scale | offset x, y
        o4 = o4.view(qn * P, 4)                                                # score, scale, offset x, offset y
        res = ops.detector_decode(o4[:, 0:1], o4[:, 2:4], o4[:, 1:2], hs, ws, self.pool_ratio, batch=qn)
        return o4.view(qn, hs, ws, 4), res.view(qn, 5), (hs, ws)

    def detect_impl(self, *a, **k):
        """cfg key 'math_mode' ('bf16' / 'fp16'; default fp32) selects the matrix-core operand precision of this network's conv /
        correlation launches; absent, an enclosing `ops.math_mode(...)` context applies."""
        with ops.math_mode(self.cfg.get("math_mode"), inherit_if_none=True):
            return self._detect_impl_fp(*a, **k)

    def _detect_impl_fp(self, que_imgs):
        """que_imgs [qn,3,hq,wq] in [0,1] -> the reference's output dict (detector.py:232-266) plus
        'positions' [qn,2] and 'scales' [qn] already decoded on the device."""
        outs, results = [], []
        # the queries of a chunk share every launch; the chunk is sized by the image: the one-launch-per-layer pyramid addresses all
        # scales of the batch with 32-bit offsets from one base (< 2^29 floats; the widest tensor is the first layer's pooled output,
        # 64 channels at a quarter of the pixels) — 16 queries of 480x640, 9 of 720x1280; an image whose pyramid alone exceeds the
        # reach runs one trunk pass per scale
        _, _, hq, wq = que_imgs.shape
        per_query = sum(self._scale_size(hq, wq, sc)[0] * self._scale_size(hq, wq, sc)[1] for sc in self.cfg["detection_scales"]) // 4 * 64
        step = max(1, min(MAX_BATCH, ((1 << 29) - 1) // per_query))
        multi = per_query < (1 << 29)
        for q0 in range(0, que_imgs.shape[0], step):
            o4, res, _ = self._detect_batch(que_imgs[q0:q0 + step].contiguous(), multi)
            outs.append(o4.permute(0, 3, 1, 2))
            results.append(res)
        o = ops.cat1(outs, 0)                                                  # qn,4,hs,ws
        r = ops.cat1(results, 0)
        return {"scores": o[:, 0:1], "select_pr_scale": o[:, 1:2], "select_pr_offset": o[:, 2:4],
                "que_select_id": r[:, 3:5].round().long(), "pool_ratio": self.pool_ratio,
                "positions": r[:, 0:2], "scales": r[:, 2]}

    def forward(self, data):
        self.load_impl(data["ref_imgs_info"]["imgs"])
        return self.detect_impl(data["que_imgs_info"]["imgs"])

    @staticmethod
    def parse_detection(scores, scales, offsets, pool_ratio):
        """Same contract as BaseDetector.parse_detection (detector.py:97-121), torch ops on the caller's device."""
        qn, _, hs, ws = scores.shape
        flat = torch.argmax(scores.flatten(1), 1)
        sy, sx = flat // ws, flat % ws
        ar = torch.arange(qn, device=scores.device)
        pos = (torch.stack([sx, sy], -1) + offsets[ar, :, sy, sx] + 0.5) * pool_ratio - 0.5
        return pos, 2 ** scales[ar, 0, sy, sx]

    # ------------------------------------------------------------------ numpy API used by Gen6DEstimator
    def load_ref_imgs(self, ref_imgs):
        """ref_imgs: uint8 [rfn,h,w,3] (reference detector.py:277-289)."""
        x = torch.from_numpy(np.ascontiguousarray(ref_imgs)).to(self.device_()).float().div_(255).permute(0, 3, 1, 2)
        with torch.no_grad():
            self.load_impl(x.contiguous())

    def detect_que_imgs(self, que_imgs):
        """que_imgs: uint8 [qn,h,w,3] -> {'positions': [qn,2], 'scales': [qn]} numpy (reference detector.py:291-304)."""
        x = torch.from_numpy(np.ascontiguousarray(que_imgs)).to(self.device_()).float().div_(255).permute(0, 3, 1, 2)
        with torch.no_grad():
            out = self.detect_impl(x.contiguous())
        return {"positions": out["positions"].cpu().numpy(), "scales": out["scales"].cpu().numpy()}
