"""VolumeRefiner — drop-in for the reference's network/refiner.py (same constructor, state_dict keys, forward dict).

One refinement step (reference forward, refiner.py:249-269):
    own channels-last Winograd trunk (backbone.py) on the 6 reference crops + the query crop, g6d_l2norm_rows on its taps
    RefineFeatureNet 2-D convs on g6d_conv_igemm; every InstanceNorm2d is a per-image (sum, sumsq) epilogue of the
        producing conv and an affine(+ReLU) in the loader of the consuming conv / up-sampler      refiner.py:24-51,64-78
    g6d_refiner_volume: projection + bilinear sampling + mean/std over references, fused            refiner.py:183-247
    RefineVolumeEncodingNet 3x3x3 convs on g6d_conv_igemm (32^3 layers: Winograd F(4x4,3x3), depth taps folded into the reduction;
        16^3: F(2x2,3x3); 8^3/4^3 and stride-2 layers: implicit 3-D GEMM with split-K)
    g6d_linear_gemv_batch: the 32768->512 FC is a 67 MB weight stream, read once per 8 queries                                    refiner.py:153-166
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import ops, specs
from .backbone import pack_trunk, trunk_features
from .operator import pose_apply_th
from .params import ParamBank, fold_vgg

# Winograd F(4x4,3x3) filters for the stride-1 3x3x3 layers of the volume net (csrc/wino43_conv.hip); VOLUME_F43_LAYERS: which of
# conv0 (32^3), conv2 (16^3), conv4 (8^3) take it beside the two embed pairs (32^3)
# Both pay from ~4 queries per launch on: one query's 7 crops / single volume leave most CUs without a block and the kernel's longer
# per-block prologue / epilogue shows (measured per step at batch 1: crops' trunk 475 vs 320 us, volume layers 541 vs 418 us on
# F(2x2,3x3); at batch 8: 1.06 vs 1.43 ms and 1.96 vs 3.06 ms the other way round)
FEAT16 = False          # round 6 experiment: the 2-D feature net's convs on the direct split-precision kernel where a map fills whole 128-pixel
                        # tiles (32 x 32 and 16 x 16 maps).  Correct (refiner goldens hold) but no faster at 112 crops: 300.8 against 304.1
                        # images/s — the two extra elementwise passes per pair and the small grids eat what the matrix cores save; off
F43_MIN_QUERIES = 4
TRUNK_F43 = True        # the crops' VGG trunk on the F(4x4,3x3) kernel (1.31-1.35x over F(2x2,3x3) at 56 crops, profiles/r04_w43_bench_v5.md)
# feature-net layers (name, index in the Sequential) that carry F(4x4,3x3) filters — measured per layer at 56 crops against F(2x2,3x3)
# (profiles/r04_layer_table_featnet_f43.md): 512->256 @16x16 159 vs 173 us, 192->128 @32x32 119 vs 148, 128->128 @32x32 107 vs 129 kept;
# 256->64 @32x32 131 vs 94, 256->64 @16x16 88 vs 60, 512->256 @8x8 108 vs 80 stay on F(2x2,3x3)
FEATNET_F43_LAYERS = {("conv1", 0), ("conv_out", 0), ("conv_out", 3)}
VOLUME_F43 = True
VOLUME_F43_LAYERS = ("conv0",)          # measured per batch of 8: conv2 (16^3) 147 vs 151 us, conv4 (8^3) 144 vs 94 us on F(2x2,3x3): only 32^3 pays
MAX_BATCH = 32         # queries that share one set of launches (g6d_linear_gemv_batch: 8 right-hand sides per weight pass)
_K3, _P3 = (3, 3, 3), (1, 1, 1)
_K2, _P2 = (1, 3, 3), (0, 1, 1)


_LIN = {}


def _linspace(sn, dev):
    """torch.linspace(-1, 1, sn) cached per (sn, device): a constant of the configuration, not a launch per step."""
    key = (sn, str(dev))
    if key not in _LIN:
        _LIN[key] = torch.linspace(-1, 1, sn, dtype=torch.float32, device=dev)
    return _LIN[key]


class RefFeatureCache:
    """Features of aligned reference crops ([fh,fw,C] device tensors) keyed by (database, view id, in-plane angle bucket, crop
    size): with `ref_feat_cache_deg` > 0 the alignment angle of a reference view is snapped to that grid, so a (view, bucket) pair
    always yields the same crop and its trunk + feature-net pass (6 of the 7 images of a refinement step) can be skipped when the
    pair repeats — between the refinement steps of a query and between queries (the reference re-warps and
    re-extracts all six views every step, refiner.py:289-313).  LRU-bounded (0.5 MB per entry)."""

    def __init__(self, max_entries=2048):
        from collections import OrderedDict
        self.store, self.max_entries = OrderedDict(), max_entries
        self.hits = self.misses = 0

    def get(self, key):
        f = self.store.get(key)
        if f is None:
            self.misses += 1
            return None
        self.store.move_to_end(key)
        self.hits += 1
        return f

    def put(self, key, feat):
        self.store[key] = feat
        while len(self.store) > self.max_entries:
            self.store.popitem(last=False)

    def clear(self):
        self.store.clear()
        self.hits = self.misses = 0

    @property
    def hit_rate(self):
        return self.hits / max(1, self.hits + self.misses)


class VolumeRefiner(ParamBank):
    # ref_feat_cache_deg: 0 = the reference's exact alignment, nothing cached (default); > 0 = alignment angles snapped to this
    # grid (degrees) and reference-crop features cached per (view, bucket) — opt-in: the snapped angle changes the crops
    # lowp_keep_fp32 / lowp_only: as in ViewpointSelector — parts that stay on fp32 matrix-core operands in the reduced-precision mode:
    # "trunk" (VGG on the crops), "featnet" (the four conv pairs of the 2-D feature net), "embed" (mean_embed / var_embed on the 32^3
    # volume), "stack" (volume_net conv0..conv4), "tail" (conv5 pair on 8^3 / 4^3 maps); lowp_only: ONLY the named parts on 16 bits
    # (tools/lowp_refiner_sensitivity.py)
    default_cfg = {"refiner_sample_num": 32, "ref_feat_cache_deg": 0.0, "lowp_keep_fp32": (), "lowp_only": None}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}
        super().__init__(specs.refiner_rows())
        self.ref_database = None
        self.ref_ids = None
        self.feat_cache = RefFeatureCache()

    def load_state_dict(self, *a, **k):
        self.feat_cache.clear()                      # cached features belong to the weights they were computed with
        return super().load_state_dict(*a, **k)

    def _mm(self, *names):
        """math-mode context of one part: fp32 if listed in cfg['lowp_keep_fp32'] (with cfg['lowp_only']: if NOT listed there)."""
        only = self.cfg.get("lowp_only")
        keep = self.cfg.get("lowp_keep_fp32") or ()
        fp32 = (not any(n in only for n in names)) if only is not None else any(n in keep for n in names)
        return ops.math_mode("fp32" if fp32 else None, inherit_if_none=True)

    def angle_step(self):
        """Snap grid of the reference alignment in radians (0 = exact, no caching)."""
        return float(np.radians(self.cfg.get("ref_feat_cache_deg") or 0.0))

    def cached_ref_feats(self, keys, make_imgs):
        """Features [len(keys),fh,fw,C] of the reference crops with cache keys `keys`; make_imgs(missing_positions) must return the
        crops [m,3,h,w] in [0,1] of the positions that are not cached (they go through the feature net in one batch)."""
        # features depend on the matrix-core operand precision they were computed in: the mode is part of the key, and misses run in
        # the network's own mode like _step does
        with ops.math_mode(self.cfg.get("math_mode"), inherit_if_none=True):
            keys = [(k, ops.MATH_MODE) for k in keys]
            got = [self.feat_cache.get(k) for k in keys]
            miss = [i for i, f in enumerate(got) if f is None]
            # (f43=False: a cached feature must not depend on how many keys missed together — the F(2x2,3x3) / F(4x4,3x3) choice of
            # run_feature_net follows the batch size otherwise)
            feats = self.run_feature_net(make_imgs(miss), f43=False) if miss else None
        if miss:
            for j, i in enumerate(miss):
                got[i] = feats[j].clone()
                self.feat_cache.put(keys[i], got[i])
        return torch.stack(got, 0)

    # ------------------------------------------------------------------ weights
    def _pack(self):
        if self._packed is None:
            pk = {"vgg": pack_trunk(fold_vgg(self, "feature_net.backbone.features"))}
            for name in ("conv0", "conv1", "conv2", "conv_out"):
                pk[name] = [self.conv_w(f"feature_net.{name}.{i}", wino_kd=1, f43=(name, i) in FEATNET_F43_LAYERS) for i in (0, 3)]
            # the 32^3 volume layers (mean_embed, var_embed, conv0) also carry F(4x4,3x3) filters (VOLUME_F43): 1.4-1.5x faster there
            # at 1.3-3.4e-6 of the layer's range (profiles/r04_w43_bench_v5.md)
            for name in ("mean_embed", "var_embed", "conv5"):         # conv5 works on 8^3 -> 4^3 maps: never on the Winograd kernel
                pk["v_" + name] = [self.conv_w(f"volume_net.{name}.{i}", wino_kd=0 if name == "conv5" else 3, f43=VOLUME_F43) for i in (0, 3)]
            for name in ("conv0", "conv1", "conv2", "conv3", "conv4"):
                pk["v_" + name] = self.conv_w(f"volume_net.{name}.0", wino_kd=3 if name in ("conv0", "conv2", "conv4") else 0,
                                              f43=VOLUME_F43 and name in VOLUME_F43_LAYERS)
            # fc.0.0 consumes x.flatten(1) of [512,4,4,4] (index c*64+v); our code is [v][c] -> permute once
            w = self.p("regressor.fc.0.0.weight")
            pk["fc0"] = (w.reshape(512, 512, 64).permute(0, 2, 1).reshape(512, 32768).contiguous(),
                         self.p("regressor.fc.0.0.bias").contiguous())
            pk["fc1"] = (self.p("regressor.fc.1.0.weight").contiguous(), self.p("regressor.fc.1.0.bias").contiguous())
            heads = ("fcr", "fct", "fcs")
            pk["heads"] = (torch.cat([self.p(f"regressor.{h}.weight") for h in heads], 0).contiguous(),
                           torch.cat([self.p(f"regressor.{h}.bias") for h in heads], 0).contiguous())
            self._packed = pk
        return self._packed

    def _feat16_filters(self, name, idx, w):
        cache = self.__dict__.setdefault("_feat16", {})
        key = (name, idx, w.data_ptr())
        if key not in cache:
            cache[key] = ops.conv16_pack(w, 3, layout=1)
        return cache[key]

    # ------------------------------------------------------------------ 2-D feature net
    def run_feature_net(self, imgs, f43=None):
        """imgs [n,3,h,w] in [0,1] -> channels-last features [n,h/4,w/4,128] (reference refiner.py:64-78).
        f43: None = the F(4x4,3x3) kernels from 4 queries (28 crops) per launch on (below that F(2x2,3x3) is faster); False = never
        (results that are cached per crop must not depend on the batch they were computed in)."""
        pk = self._pack()
        n, _, h, w = imgs.shape
        dev = imgs.device
        big = (n >= F43_MIN_QUERIES * 7) if f43 is None else bool(f43)
        with self._mm("trunk"):
            f3, f5, f7 = trunk_features(pk["vgg"], imgs, ("c3", "c5", "c7_pre"), True, f43=TRUNK_F43 and big)      # channels-last, L2-normalised

        def pair(name, x):
            """conv, IN, ReLU, conv, (IN returned as affine) — per-image statistics."""
            (w0, b0), (w1, b1) = pk[name]
            u0, u1 = pk[name][0].u, pk[name][1].u
            v0, v1 = (pk[name][0].u43, pk[name][1].u43) if big else (None, None)
            _, _, hh, ww, _ = x.shape
            y0 = torch.empty((n, 1, hh, ww, w0.shape[0]), dtype=torch.float32, device=dev)
            s0 = ops.new_stats(n, w0.shape[0], dev)
            y1 = torch.empty((n, 1, hh, ww, w1.shape[0]), dtype=torch.float32, device=dev)
            s1 = ops.new_stats(n, w1.shape[0], dev)
            if FEAT16 and ops.MATH_MODE == 0 and (hh * ww) % 128 == 0 and all(w_.shape[2] % 32 == 0 and w_.shape[0] % 64 == 0 for w_ in (w0, w1)):
                # round 6: both convs on the direct split-precision kernel (fp16 hi / lo pairs, fp32-class results, csrc/conv16_direct.hip): the
                # input / the first norm's affine + ReLU are written in the kernel's format by one elementwise pass each, the per-image
                # InstanceNorm sums come out of the convs' epilogues
                f0, f1 = self._feat16_filters(name, 0, w0), self._feat16_filters(name, 1, w1)
                ops.conv16_direct_multi([ops.affine_split16(x, None, None, 0, False, False, 3)], f0, b0, relu=False, full=torch.float32, pool=None,
                                        stats=s0, rows_per_group=hh * ww, out_full=[y0[:, 0]])
                sc0, sh0 = ops.stats_finalize(s0, hh * ww)
                ops.conv16_direct_multi([ops.affine_split16(y0, sc0, sh0, 1, True, False, 3)], f1, b1, relu=False, full=torch.float32, pool=None,
                                        stats=s1, rows_per_group=hh * ww, out_full=[y1[:, 0]])
                sc1, sh1 = ops.stats_finalize(s1, hh * ww)
                return y1, sc1, sh1
            sc0, sh0 = ops.conv(x, w0, b0, y0, ksize=_K2, pad=_P2, stats=s0, rows_per_group=hh * ww, w_wino=u0, w_wino43=v0, finalize=hh * ww)
            sc1, sh1 = ops.conv(y0, w1, b1, y1, ksize=_K2, pad=_P2, in_scale=sc0, in_shift=sh0, in_relu=True, per_n=True,
                                stats=s1, rows_per_group=hh * ww, w_wino=u1, w_wino43=v1, finalize=hh * ww)
            return y1, sc1, sh1

        hq, wq = h // 4, w // 4
        cat = torch.empty((n, 1, hq, wq, 192), dtype=torch.float32, device=dev)
        def b0():
            y, sc, sh = pair("conv0", f3)
            ops.affine_act_pool(y, cat[..., 0:64], sc, sh, per_n=True)

        def b1():
            y, sc, sh = pair("conv1", f5)
            ops.upsample_bilinear(y, cat[..., 64:128], 2, sc, sh, per_n=True)

        def b2():
            y, sc, sh = pair("conv2", f7)
            ops.upsample_bilinear(y, cat[..., 128:192], 4, sc, sh, per_n=True)

        with self._mm("featnet"):
            ops.fork_join([b0, b1, b2], dev)
            y, sc, sh = pair("conv_out", cat)
        out = torch.empty((n, 1, hq, wq, 128), dtype=torch.float32, device=dev)
        ops.affine_act_pool(y, out, sc, sh, per_n=True)
        return out.view(n, hq, wq, 128)

    # ------------------------------------------------------------------ 3-D volume net + regressor
    def run_volume_net(self, mean_in, std, sn):
        """mean_in [sn^3,256], std [sn^3,128] -> code [(sn/8)^3, 512] (reference refiner.py:136-143); or a batch of qn volumes
        ([qn,sn^3,256], [qn,sn^3,128] -> [qn,(sn/8)^3,512]) through the same launches: every InstanceNorm3d keeps one statistics group
        and one affine table per volume."""
        pk = self._pack()
        dev = mean_in.device
        batched = mean_in.dim() == 3
        qn = mean_in.shape[0] if batched else 1
        pn = 1 if qn > 1 else 0                      # per-volume tables / groups only when there is more than one

        def c3(x, wb, out, stride=1, aff=None, stats_c=None, count=None):
            """3x3x3 conv; with stats_c: returns the affine (scale, shift) of the InstanceNorm that follows (count values)."""
            st = ops.new_stats(qn, stats_c, dev) if stats_c else None
            sc, sh = aff if aff is not None else (None, None)
            return ops.conv(x, wb[0], wb[1], out, ksize=_K3, stride=(stride,) * 3, pad=_P3, in_scale=sc, in_shift=sh,
                            in_relu=aff is not None, per_n=pn if aff is not None else 0, stats=st,
                            rows_per_group=pn * (count or 0) if stats_c else 0,
                            w_wino=getattr(wb, "u", None) if stride == 1 else None,
                            w_wino43=getattr(wb, "u43", None) if (stride == 1 and qn >= F43_MIN_QUERIES) else None,
                            finalize=count if stats_c else None)

        def buf(s, c):
            return torch.empty((qn, s, s, s, c), dtype=torch.float32, device=dev)

        vox = sn ** 3
        cat = buf(sn, 128)
        def embed(name, x, sl):
            y = buf(sn, 64)
            aff = c3(x, pk[name][0], y, stats_c=64, count=vox)
            c3(y, pk[name][1], cat[..., sl], aff=aff)

        with self._mm("embed"):
            embed("v_mean_embed", mean_in.view(qn, sn, sn, sn, 256), slice(0, 64))
            embed("v_var_embed", std.view(qn, sn, sn, sn, 128), slice(64, 128))
        x, aff, s = cat, None, sn
        with self._mm("stack"):
            for name, co, stride in (("v_conv0", 64, 1), ("v_conv1", 128, 2), ("v_conv2", 128, 1), ("v_conv3", 256, 2),
                                     ("v_conv4", 256, 1)):
                s = s // stride
                y = buf(s, co)
                x, aff = y, c3(x, pk[name], y, stride=stride, aff=aff, stats_c=co, count=s ** 3)
        s = s // 2
        y = buf(s, 512)
        with self._mm("tail"):
            aff = c3(x, pk["v_conv5"][0], y, stride=2, aff=aff, stats_c=512, count=s ** 3)
            code = buf(s, 512)
            c3(y, pk["v_conv5"][1], code, aff=aff)
        return code.view(qn, s ** 3, 512) if batched else code.view(s ** 3, 512)

    def run_regressor(self, code):
        """code [v,512] (one query) or [qn,v,512] -> rotation [qn,4] (unit quaternion), offset [qn,2], scale [qn,1]; the 67 MB FC
        weight stream is read once per 8 queries (g6d_linear_gemv_batch)."""
        pk = self._pack()
        qn = code.shape[0] if code.dim() == 3 else 1
        x = ops.linear_gemv(code.reshape(qn, -1), pk["fc0"][0], pk["fc0"][1], act=2)
        x = ops.linear_gemv(x, pk["fc1"][0], pk["fc1"][1], act=2)
        o = ops.linear_gemv(x, pk["heads"][0], pk["heads"][1])
        ops.l2norm_rows(o[:, 0:4])                  # F.normalize(quaternion) in place (rows of four values, row stride 7)
        return o[:, 0:4], o[:, 4:6], o[:, 6:7]

    def _step(self, *a, **k):
        """cfg key 'math_mode' ('bf16' / 'fp16'; default fp32) selects the matrix-core operand precision of this network's conv /
        correlation launches; absent, an enclosing `ops.math_mode(...)` context applies."""
        with ops.math_mode(self.cfg.get("math_mode"), inherit_if_none=True):
            return self._step_fp(*a, **k)

    def _step_fp(self, que_img, K_in, pose_in, ref_imgs, ref_Ks, ref_poses, ref_feats=None):
        """One refinement step.  ref_feats [rfn,fh,fw,C] (single query only): features of the reference crops computed earlier
        (cached_ref_feats) — ref_imgs is then ignored and only the query crop goes through the trunk + feature net.  Single query: que_img [1,3,h,w], K_in [3,3], pose_in [3,4], ref_imgs [rfn,3,h,w], ref_Ks [rfn,3,3],
        ref_poses [rfn,3,4].  Batch of qn <= MAX_BATCH queries (reference forward: refiner.py:249-269 takes [qn,...]): que_img
        [qn,3,h,w], K_in [qn,3,3], pose_in [qn,3,4], ref_imgs [qn,rfn,3,h,w], ref_Ks [qn,rfn,3,3], ref_poses [qn,rfn,3,4] — the
        (rfn+1)*qn crops share the trunk / feature-net launches, the qn volumes the volume-net launches and the FC weight stream."""
        sn = self.cfg["refiner_sample_num"]
        dev = que_img.device
        ops.stats_arena_begin(dev)
        if ref_feats is not None:
            return self._step_from_feats(que_img, K_in, pose_in, ref_feats, ref_Ks, ref_poses)
        batched = ref_imgs.dim() == 5
        qn = ref_imgs.shape[0] if batched else 1
        rfn = ref_imgs.shape[-4]
        h_in, w_in = ref_imgs.shape[-2:]
        if batched:                                    # image order: (refs of query 0, query 0), (refs of query 1, query 1), ...
            imgs = torch.cat([ref_imgs, que_img[:, None]], 1).reshape(qn * (rfn + 1), *ref_imgs.shape[-3:])
        else:
            imgs = torch.cat([ref_imgs, que_img], 0)                                         # query last
        feats = self.run_feature_net(imgs)
        lin = _linspace(sn, dev)
        C = feats.shape[-1]
        lead = (qn,) if batched else ()
        mean_in = torch.empty(lead + (sn ** 3, 2 * C), dtype=torch.float32, device=dev)
        std = torch.empty(lead + (sn ** 3, C), dtype=torch.float32, device=dev)
        feats = feats.contiguous()
        if batched:
            feats = feats.view(qn, rfn + 1, *feats.shape[1:])
        # projections K @ pose and the volume's rotation are formed inside the kernel (reference refiner.py:208-226)
        ops.refiner_volume_kp(feats, ref_Ks.contiguous(), ref_poses.contiguous(), K_in.contiguous(), pose_in.contiguous(),
                              lin, h_in, w_in, mean_in, std)
        return self.run_regressor(self.run_volume_net(mean_in, std, sn))

    def _step_from_feats(self, que_img, K_in, pose_in, ref_feats, ref_Ks, ref_poses):
        """The step with the reference crops' features given (cache hits): only the query crop(s) pass the trunk + feature net.
        Single query: ref_feats [rfn,fh,fw,C]; batch: ref_feats [qn,rfn,fh,fw,C] with the [qn,...] operands of _step_fp."""
        sn = self.cfg["refiner_sample_num"]
        dev = que_img.device
        h_in, w_in = que_img.shape[-2:]
        batched = ref_feats.dim() == 5
        qf = self.run_feature_net(que_img)                                               # [qn,fh,fw,C]
        if batched:
            feats = torch.cat([ref_feats, qf[:, None]], 1).contiguous()                  # [qn,rfn+1,fh,fw,C], query last
            lead = (ref_feats.shape[0],)
        else:
            feats = torch.cat([ref_feats, qf], 0).contiguous()
            lead = ()
        lin = _linspace(sn, dev)
        C = feats.shape[-1]
        mean_in = torch.empty(lead + (sn ** 3, 2 * C), dtype=torch.float32, device=dev)
        std = torch.empty(lead + (sn ** 3, C), dtype=torch.float32, device=dev)
        ops.refiner_volume_kp(feats, ref_Ks.contiguous(), ref_poses.contiguous(), K_in.contiguous(), pose_in.contiguous(),
                              lin, h_in, w_in, mean_in, std)
        return self.run_regressor(self.run_volume_net(mean_in, std, sn))

    def forward(self, data):
        """Same dict contract as the reference (refiner.py:249-269)."""
        is_inference = data["inference"] if "inference" in data else False
        que, ref = data["que_imgs_info"], data["ref_imgs_info"]
        qn_all = que["imgs"].shape[0]
        if qn_all == 1:
            outs = [self._step(que["imgs"], que["Ks_in"][0], que["poses_in"][0], ref["imgs"][0], ref["Ks"][0], ref["poses"][0])]
        else:                                                                  # the queries of a chunk share every launch
            outs = [self._step(que["imgs"][q0:q0 + MAX_BATCH], que["Ks_in"][q0:q0 + MAX_BATCH], que["poses_in"][q0:q0 + MAX_BATCH],
                               ref["imgs"][q0:q0 + MAX_BATCH], ref["Ks"][q0:q0 + MAX_BATCH], ref["poses"][q0:q0 + MAX_BATCH])
                    for q0 in range(0, qn_all, MAX_BATCH)]
        out = {"rotation": ops.cat1([o[0] for o in outs], 0), "offset": ops.cat1([o[1] for o in outs], 0),
               "scale": ops.cat1([o[2] for o in outs], 0)}
        if not is_inference:
            sn = self.cfg["refiner_sample_num"]
            g = torch.linspace(-1, 1, sn, dtype=torch.float32, device=que["imgs"].device)
            V = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, sn ** 3, 3) @ que["poses_in"][:, :3, :3]
            out["grids"] = pose_apply_th(que["poses_in"], V)
        return out

    # ------------------------------------------------------------------ estimator API
    def load_ref_imgs(self, ref_database, ref_ids):
        """reference refiner.py:270-272.  A new database drops whatever an own image cache held for the previous one
        (Gen6DEstimator.build shares and clears its cache itself)."""
        cache = getattr(self, "image_cache", None)
        if ref_database is not self.ref_database and cache is not None and not cache.holds(ref_database):
            cache.clear()
        if ref_database is not self.ref_database:
            self.feat_cache.clear()
        self.ref_database = ref_database
        self.ref_ids = ref_ids

    def refine_que_imgs(self, que_img, que_K, in_pose, size=128, ref_num=6, ref_even=False):
        """One refinement step with the reference's signature (refiner.py:275-341): que_img uint8 [H,W,3] (numpy or a
        device tensor), que_K [3,3], in_pose [3,4] -> refined pose [3,4] float32.  The look-at crop of the query and the
        6 aligned reference crops are device warps (g6d_warp_perspective) of images cached on the GPU; the pose algebra
        stays on the host."""
        from .. import estimator as E
        from .. import geometry as G
        margin, even_num = 0.05, min(128, len(self.ref_ids))
        dev = self.device_()
        cache = getattr(self, "image_cache", None)
        if cache is None:
            cache = self.image_cache = E.DeviceImageCache(dev)
        db = E.NormalizedDatabase(self.ref_database)
        in_pose = G.normalize_pose(np.asarray(in_pose, np.float64), db.scale, db.offset)
        center, diameter = db.object_center, db.object_diameter
        que_K = np.asarray(que_K, np.float64)
        _, new_f = G.let_me_look_at(in_pose, que_K, center)
        in_dist = np.linalg.norm(G.pose_inverse(in_pose)[:, 3] - center)
        scale = size * (1 - margin) / diameter * in_dist / new_f
        position = G.project_points(center[None], in_pose, que_K)[0][0]
        K_warp, pose_warp, pose_rect, H = G.look_at_crop_params(que_K, in_pose, position, 0, scale, size, size)
        if not torch.is_tensor(que_img):
            que_img = torch.from_numpy(np.ascontiguousarray(que_img)).to(dev)
        que_warp = ops.warp_perspective(que_img, H, size, size)
        ref_ids = E.select_reference_img_ids_refinement(db, center, self.ref_ids, pose_warp, ref_num, ref_even, even_num, cache)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        que_t = que_warp.float().div_(255).permute(2, 0, 1)[None].contiguous()
        step = self.angle_step()
        with torch.no_grad():
            if step > 0:
                # alignment angles snapped to the grid: the crop of a view depends on (view, bucket) only -> cached features
                ref_Ks, ref_poses, Hs, buckets = E.reference_view_params(db, ref_ids, size, margin, True, pose_warp, K_warp, angle_step=step)
                keys = [(id(self.ref_database), str(i), int(b), int(size)) for i, b in zip(ref_ids, buckets)]
                make = lambda miss: torch.stack([ops.warp_perspective(cache.get(db, ref_ids[k]), Hs[k], size, size) for k in miss], 0) \
                    .float().div_(255).permute(0, 3, 1, 2).contiguous()
                ref_feats = self.cached_ref_feats(keys, make)
                rot, off, scl = self._step(que_t, f(K_warp), f(pose_warp), None, f(ref_Ks), f(ref_poses), ref_feats=ref_feats)
            else:
                ref_imgs, _, ref_Ks, ref_poses, _ = E.normalize_reference_views(db, ref_ids, size, margin, cache, True, pose_warp, K_warp,
                                                                             with_masks=False)
                rot, off, scl = self._step(que_t, f(K_warp), f(pose_warp), ref_imgs.float().div_(255).permute(0, 3, 1, 2).contiguous(),
                                           f(ref_Ks), f(ref_poses))
            out = torch.cat([rot[0], off[0], scl[0]]).cpu().numpy()
        quat, offset, scale_pr = out[:4], out[4:6], 2 ** out[6]
        pose_sim = G.compose_sim_pose(scale_pr, quat, offset, pose_warp, center)
        pose_pr = G.pose_sim_to_pose_rigid(pose_sim, pose_warp, K_warp, K_warp, center)
        pose_pr = G.pose_compose(pose_pr, G.pose_inverse(pose_rect))
        return G.denormalize_pose(pose_pr, db.scale, db.offset)

    def refine_step_tensors(self, que_img_u8, K_in, pose_in, ref_imgs_u8, ref_Ks, ref_poses):
        """Numpy boundary of one step after the host-side warps: uint8 crops [h,w,3] / [rfn,h,w,3], float32 K/poses
        -> (quat [4], scale (=2**s) [1], offset [2]) numpy, as consumed at reference refiner.py:327-331."""
        dev = self.device_()
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        q = torch.from_numpy(np.ascontiguousarray(que_img_u8)).to(dev).float().div_(255).permute(2, 0, 1)[None].contiguous()
        r = torch.from_numpy(np.ascontiguousarray(ref_imgs_u8)).to(dev).float().div_(255).permute(0, 3, 1, 2).contiguous()
        with torch.no_grad():
            rot, off, scl = self._step(q, f(K_in), f(pose_in), r, f(ref_Ks), f(ref_poses))
        return rot[0].cpu().numpy(), 2 ** scl[0].cpu().numpy(), off[0].cpu().numpy()
