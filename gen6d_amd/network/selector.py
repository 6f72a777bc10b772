"""ViewpointSelector — drop-in for the reference's network/selector.py (same constructor, state_dict keys, methods).

Query-time data flow (reference compute_view_point_feats, selector.py:177-215), D = rfn*an hypotheses, d = r*an + a:
    own Winograd trunk + g6d_l2norm_rows                  query features, 3 levels, channels-last
  per level l:
    g6d_selector_scan          score maps sum_c q*r and the "vps" scalars  (one coalesced pass over the ref cache)
    g6d_selector_prod_affine   InstanceNorm3d(512) statistics of the never-materialised product q*r from R1/R2
    g6d_conv_igemm             first (1,3,3) conv with the product, the norm affine and the zero padding fused into
                               its operand loader; later convs fuse the preceding InstanceNorm(+ReLU) the same way and
                               accumulate the next InstanceNorm's statistics in their epilogue
    g6d_affine_act_pool        only where a MaxPool sits between two convs
  tail: 1x1 convs as GEMMs on the same MFMA kernel, g6d_vps_norm, g6d_max_an_add, g6d_attention, g6d_layernorm,
        g6d_affine_act_add.  AvgPool(1,4,4) is commuted in front of the last 1x1x1 conv (both linear).

Reference-sharded mode (`set_shard(rank, world)`): every rank keeps the cache of a contiguous slice of
the references (all rotations of a reference stay together).  The logits of one reference depend on all references
through the InstanceNorms, so the exchange is: one all-reduce of R1/R2 at build time; per BATCH of <= 32 queries (they share every
collective: the tables are [qn, C, 2]) five all-reduces of the fp64 (sum, sumsq) InstanceNorm tables of the correlation stacks (the
three levels advance in lock-step, a round's tables travel together: <= qn * 3 * 8 KB), one for corr_feats_conv's InstanceNorm, one
all-gather of the vps scalars, one of the per-reference feature rows [qn, rfn/G, 512] before the replicated attention tail, and one
of the per-reference angles: 9 collectives per batch (RCCL over xGMI via torch.distributed, issued on device tensors on the current
stream; messages are KB-sized, i.e. latency-bound).  Results equal the unsharded ones up to fp reassociation of the sums.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import ops, parallel, specs
from .backbone import pack_trunk, trunk_features
from .params import ParamBank, fold_vgg

# per level: (conv index in corr_conv_list[l], InstanceNorm after?, ReLU after?, MaxPool after?)  selector.py:27-69
_CORR = (
    ((1, 1, 1, 0), (4, 1, 0, 1), (7, 1, 1, 0), (10, 1, 0, 1), (13, 1, 1, 0), (16, 0, 0, 0)),
    ((1, 1, 1, 0), (4, 1, 0, 1), (7, 1, 1, 0), (10, 0, 0, 0)),
    ((1, 1, 1, 0), (4, 0, 0, 0)),
)
_K133, _P011 = (1, 3, 3), (0, 1, 1)
PRODUCT16 = True       # round 6: the first conv of every level (query x reference product) on the direct 16-bit convolution — the product written
                       # once as fp16 hi / lo pairs (fp32 path) or 16-bit activations; False (tools / tests): the product prologue of the Winograd /
                       # implicit-GEMM kernels
STACK16 = True         # ... and the levels' other 3x3 convs (InstanceNorm stacks): the previous norm's affine + ReLU as an elementwise pass into the
                       # kernel's format (g6d_affine_split16); False: the operand prologues of the Winograd / implicit-GEMM kernels
MAX_BATCH = 32         # queries that share one set of launches (BASELINE configs[4]: 32 concurrent queries; g6d_selector_levels runs them
                       # in groups of 8 query rows per pass over the reference cache)
FEAT_LD = 516          # 512 corr channels + 3 vps channels + 1 zero pad (16-byte rows)


class ViewpointSelector(ParamBank):
    # lowp_keep_fp32: parts that stay on fp32 matrix-core operands when the reduced-precision mode is on (cfg `math_mode` or an enclosing
    # ops.math_mode): "trunk" = the query's VGG trunk, "product" = the first conv of every level (query x reference product), "stack" =
    # the later convs of the three InstanceNorm stacks, "corr<l>.<i>" = conv i of level l alone, "fuse" = corr_feats_conv, "tail" =
    # score_process / attention / predictors; () = none.  lowp_only (tools/lowp_selector_sensitivity.py): when set, ONLY the named parts
    # run on 16-bit operands.
    # Default ("trunk", "tail"): measured part by part (profiles/r05_lowp_selector_sensitivity.md), the fp16 logit error is made by the
    # query trunk (1.0e-2 alone) and the attention / predictor tail (0.7e-2) — both a few small launches — not by the InstanceNorm stacks
    # that hold the multiply-adds (1-3e-3 each): with these two on fp32 operands the reduced-precision selector keeps its logits within
    # 0.17 of the smallest top-2 margin (bar 1/4) at the all-fp16 speed (4.88 vs 4.79 ms per batch of 8).
    default_cfg = {"selector_angle_num": 5, "lowp_keep_fp32": ("trunk", "tail"), "lowp_only": None}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}
        super().__init__(specs.selector_rows(self.cfg["selector_angle_num"]))
        self.ref_feats_cache = None      # 3 x [D, HW_l, 512] channels-last, d = r*an + a
        self.ref_sums = None             # 3 x (R1, R2) fp64 [HW_l, 512]
        self.ref_pose_embed = None       # [rfn, 512]
        self.rfn = self.an = None
        self.rank, self.world, self.group = 0, 1, None
        self.sharded = False             # True: the collective-carrying path (world > 1, or forced at world 1)
        self.r_begin, self.r_end = 0, None

    # ------------------------------------------------------------------ reference sharding
    def set_shard(self, rank, world, group=None, force_collectives=False):
        """Keep only references [begin, end) = parallel.shard_range(rfn, rank, world) on this rank (call before
        load_ref_imgs / extract_ref_feats). world == 1 restores the unsharded behaviour — unless `force_collectives`: then the
        sharded code path runs with its 9 collectives per batch on a one-rank process group (RCCL accepts every collective at world
        size 1), which is how the path meets ncclAllReduce / ncclAllGather and stream capture on a 1-GPU box."""
        self.rank, self.world, self.group = int(rank), int(world), group
        self.sharded = self.world > 1 or bool(force_collectives)

    def _allreduce(self, tensors):
        """Sum small fp64 statistics tensors over the ranks in ONE collective."""
        if not self.sharded:
            return
        flat = torch.cat([t.reshape(-1) for t in tensors])
        parallel.all_reduce_(flat, "sum", self.group)
        o = 0
        for t in tensors:
            t.copy_(flat[o:o + t.numel()].view_as(t)); o += t.numel()

    def _allgather_rows(self, rows, n_total):
        """[n_local, F] per rank -> [n_total, F] in global reference order."""
        if not self.sharded:
            return rows
        return parallel.all_gather_ragged_rows(rows, n_total, self.world, self.group)

    def _allgather_batch(self, t, qn, n_local, n_total):
        """[qn * n_local, F] (query-major rows of the local references) -> [qn * n_total, F] over all references: ONE all-gather
        for the whole batch (a rank's message = its references' rows of all qn queries)."""
        if not self.sharded:
            return t
        F_ = t.shape[1]
        rows = t.view(qn, n_local, F_).permute(1, 0, 2).reshape(n_local, qn * F_).contiguous()
        return self._allgather_rows(rows, n_total).view(n_total, qn, F_).permute(1, 0, 2).reshape(qn * n_total, F_)

    # ------------------------------------------------------------------ weights
    def _pack(self):
        if self._packed is None:
            an = self.cfg["selector_angle_num"]
            pk = {"vgg": pack_trunk(fold_vgg(self, "backbone.features"))}
            pk["corr"] = [[self.conv_w(f"corr_conv_list.{l}.{i}", wino_kd=1) for i, *_ in layers] for l, layers in enumerate(_CORR)]
            pk["fuse0"] = self.conv_w("corr_feats_conv.0")
            pk["fuse3"] = self.conv_w("corr_feats_conv.3")
            pk["sp0"] = self.conv_w("score_process.0", cin_pad=FEAT_LD)
            pk["sp2"] = self.conv_w("score_process.2")
            pk["att"] = []
            for i in range(2):
                qkv = [self.conv_w(f"atts.{i}.{n}") for n in ("conv_query", "conv_key", "conv_feats")]
                pk["att"].append({
                    "qkv": (torch.cat([w for w, _ in qkv], 0).contiguous(), torch.cat([b for _, b in qkv], 0).contiguous()),
                    "merge": self.conv_w(f"atts.{i}.conv_merge"),
                    "ln": (self.p(f"atts.{i}.norm.norm.weight").contiguous(), self.p(f"atts.{i}.norm.norm.bias").contiguous()),
                    "mlp0": self.conv_w(f"mlps.{i}.0"), "mlp3": self.conv_w(f"mlps.{i}.3")})
            pk["pred0"] = self.conv_w("score_predict.0")
            pk["pred2"] = self.conv_w("score_predict.2")
            # angle head consumes feats.permute(0,1,3,2).reshape(qn,515*an,rfn): channel c*an+a (selector.py:212-214);
            # our per-reference row is [an][516], i.e. index a*516+c -> permute the weight once.
            w, b = self.p("angle_predict.0.weight"), self.p("angle_predict.0.bias")
            w = w.reshape(512, 515, an).permute(0, 2, 1)
            w = F.pad(w, (0, FEAT_LD - 515)).reshape(512, 1, an * FEAT_LD).contiguous()
            pk["ang0"] = (w, b.contiguous())
            pk["ang2"] = self.conv_w("angle_predict.2")
            pk["ang4"] = self.conv_w("angle_predict.4")
            self._packed = pk
        return self._packed

    def _mm(self, *names):
        """math-mode context of one part of the network: fp32 if any of its names is in cfg['lowp_keep_fp32'] (or, with cfg['lowp_only'],
        if none of them is listed there), else whatever the enclosing mode is."""
        only = self.cfg.get("lowp_only")
        keep = self.cfg.get("lowp_keep_fp32") or ()
        fp32 = (not any(n in only for n in names)) if only is not None else any(n in keep for n in names)
        return ops.math_mode("fp32" if fp32 else None, inherit_if_none=True)

    # ------------------------------------------------------------------ features
    def get_feats(self, imgs):
        """imgs [n,3,h,w] in [0,1] -> 3 channels-last, L2-normalised maps [n,1,h_l,w_l,512] (selector.py:113-119)."""
        with self._mm("trunk"):
            return trunk_features(self._pack()["vgg"], imgs, ("c5", "c7_pre", "p7"), True)

    def extract_ref_feats(self, ref_imgs, ref_poses, object_center, object_vert, is_train=False):
        """ref_imgs [an,rfn,3,h,w]; builds the reference cache, its R1/R2 sums and the viewpoint embedding
        (reference selector.py:121-148, eval branch: object_forward = first reference)."""
        if is_train:
            raise NotImplementedError("inference-only implementation (random object_forward is a training augmentation)")
        an, rfn, _, h, w = ref_imgs.shape
        if an != self.cfg["selector_angle_num"]:
            raise ValueError("number of rotations does not match selector_angle_num")
        self.rfn, self.an = rfn, an
        self.r_begin, self.r_end = parallel.shard_range(rfn, self.rank, self.world)
        if self.r_end == self.r_begin:
            raise ValueError("more ranks than reference views")
        ref_imgs = ref_imgs[:, self.r_begin:self.r_end]
        D = (self.r_end - self.r_begin) * an                                 # local hypotheses
        imgs = ref_imgs.permute(1, 0, 2, 3, 4).reshape(D, 3, h, w)          # d = r*an + a
        dev = imgs.device
        cache = [torch.empty((D, 1, h >> s, w >> s, 512), dtype=torch.float32, device=dev) for s in (3, 4, 5)]
        for i0 in range(0, D, 64):
            feats = self.get_feats(imgs[i0:i0 + 64].contiguous())
            for l in range(3):
                cache[l][i0:i0 + 64].copy_(feats[l])
        self.ref_feats_cache = cache
        self.ref_sums = [ops.selector_ref_sums(c.view(D, -1, 512)) for c in cache]
        self._allreduce([t for pair in self.ref_sums for t in pair])         # R1/R2 over ALL references

        # viewpoint embedding: tiny one-time MLP on [rfn,3] (torch ops on the device)
        cam = (-ref_poses[:, :3, :3].transpose(1, 2) @ ref_poses[:, :3, 3:])[..., 0] - object_center[None]
        fwd = cam[0]
        y = torch.linalg.cross(object_vert, fwd)
        x = torch.linalg.cross(y, object_vert)
        R = torch.stack([F.normalize(x, dim=0), F.normalize(y, dim=0), F.normalize(object_vert, dim=0)], 0)
        v = F.normalize(cam @ R.T, dim=1)
        for i in (0, 2, 4):
            v = F.linear(v, self.p(f"view_point_encoder.{i}.weight"), self.p(f"view_point_encoder.{i}.bias"))
            if i != 4: v = F.relu(v)
        self.ref_pose_embed = v.contiguous()

    # ------------------------------------------------------------------ query
    def _level(self, l, q, cat, scale, shift, qn=1):
        """GENERATOR (it runs to completion without yielding when the references are not sharded; in the sharded mode it yields the
        fp64 statistics tensor of every InstanceNorm whose sums still have to be added over the ranks — the caller all-reduces it
        in place, together with those of the other levels, and resumes).  One pyramid level for a batch of qn queries: q [qn,1,h,w,512] query features, (scale, shift) [qn,512] the InstanceNorm
        affine of each query x reference product; writes channels [256l,256l+256) of cat [qn*D,1,4,4,768].  The qn*D hypothesis
        images of the batch go through every launch together: image n = query n // D, hypothesis n % D; the first conv reads the
        reference cache through `in_mod` (shared, not replicated) with the query's own feature map as multiplier, and every
        InstanceNorm keeps one statistics group / affine table per query (`rows_per_group`, `per_n` = D)."""
        pk = self._pack()
        cache = self.ref_feats_cache[l]
        D, _, h, w, _ = cache.shape
        dev = cache.device
        Dg = self.rfn * self.an                                              # global hypothesis count
        grp = D if qn > 1 else 0                                             # images per query group (0: the single-query launches of round 2)
        x, mul, relu = cache, (q.view(qn, h, w, 512) if qn > 1 else q.view(h, w, 512)), False
        first = True
        layers = _CORR[l]
        for li, (idx, has_in, has_relu, has_pool) in enumerate(layers):
            wgt, bias = pk["corr"][l][li]
            wu = pk["corr"][l][li].u
            co = wgt.shape[0]
            last = li == len(layers) - 1
            out = cat[..., 256 * l:256 * l + 256] if last else torch.empty((qn * D, 1, h, w, co), dtype=torch.float32, device=dev)
            stats = ops.new_stats(qn, co, dev) if has_in else None
            # InstanceNorm finalisation inside the producing launch, unless the statistics still have to be summed over ranks
            fin = Dg * h * w if (has_in and not last and not self.sharded) else None
            with self._mm("product" if first else "stack", f"corr{l}.{li}"):
                mode16 = self._product16_mode(first, D * h * w, co) if first else self._stack16_mode(D * h * w, co, wgt.shape[2])
                if mode16 and not first:
                    # round 6: the stack layers too — the InstanceNorm affine + ReLU of the previous layer is applied by one elementwise pass that
                    # writes the map in the direct kernel's format (g6d_affine_split16), the conv adds this layer's sums in its epilogue
                    x16 = ops.affine_split16(x, scale, shift, grp if scale is not None else 0, relu, False, mode16)
                    filt = self._product16_filters(l, li, wgt, mode16)
                    ci = wgt.shape[2]
                    per = max(1, ((1 << 31) - 1) // (D * h * w * ci * (4 if mode16 == 3 else 2)))
                    for q0 in range(0, qn, per):
                        q1 = min(qn, q0 + per)
                        ops.conv16_direct_multi([x16[q0 * D:q1 * D]], filt, bias, relu=False, full=torch.float32, pool=None,
                                                stats=stats[q0:q1] if stats is not None else None, rows_per_group=D * h * w,
                                                out_full=[out[q0 * D:q1 * D, 0]])
                    res, fin = out, None
                elif mode16:
                    # round 6: the product layer on the direct 16-bit convolution (csrc/conv16_direct.hip): the normalised query x reference
                    # product is written once in the kernel's activation format (fp32 path: fp16 hi / lo pairs, fp32-class results), the conv
                    # adds this level's InstanceNorm sums in its epilogue, the affine of that norm comes from one small finalize launch
                    prod = ops.product_split16(cache.view(D, h * w, 512), q.view(qn, h * w, 512), scale, shift, mode16)
                    prod = prod.view(qn * D, h, w, 2, 512) if mode16 == 3 else prod.view(qn * D, h, w, 512)
                    filt = self._product16_filters(l, li, wgt, mode16)
                    # (a launch addresses its input with 32-bit offsets: 2^31 bytes = 8 queries of the 16 x 16 level in pairs)
                    per = max(1, ((1 << 31) - 1) // (D * h * w * 512 * (4 if mode16 == 3 else 2)))
                    for q0 in range(0, qn, per):
                        q1 = min(qn, q0 + per)
                        ops.conv16_direct_multi([prod[q0 * D:q1 * D]], filt, bias, relu=False, full=torch.float32, pool=None, stats=stats[q0:q1],
                                                rows_per_group=D * h * w, out_full=[out[q0 * D:q1 * D]])
                    res, fin = out, None
                else:
                    res = ops.conv(x, wgt, bias, out, ksize=_K133, pad=_P011, mul=mul, in_scale=scale, in_shift=shift, in_relu=relu, stats=stats,
                                   w_wino=wu, finalize=fin, per_n=grp if scale is not None else 0, rows_per_group=grp * h * w,
                                   in_mod=grp if first else 0, mul_group=grp if mul is not None else 0)
            mul, first = None, False
            if last:
                break
            if fin is not None:
                scale, shift = res
            else:
                yield stats                                                  # [qn, co, 2] fp64: summed over the ranks by the caller
                scale, shift = ops.stats_finalize(stats, Dg * h * w)
            if has_pool:
                pooled = torch.empty((qn * D, 1, h // 2, w // 2, co), dtype=torch.float32, device=dev)
                ops.affine_act_pool(out, pooled, scale, shift, per_n=grp, relu=bool(has_relu), pool=1)
                h, w = h // 2, w // 2
                x, scale, shift, relu = pooled, None, None, False
            else:
                x, relu = out, bool(has_relu)

    def _product16_mode(self, first, rows_per_query, co):
        """conv16 math mode of a level's first (product) layer, or 0 = the Winograd / implicit-GEMM kernels with the product prologue: the
        fp32 path takes fp16 hi / lo pairs (3); the reduced-precision modes their own 16-bit type.
        A query's hypothesis images must fill whole 128-pixel tiles (its InstanceNorm sums are taken per tile): true for 64 x 5 views."""
        if not (PRODUCT16 and first and rows_per_query % 128 == 0):
            return 0
        mm = ops.MATH_MODE
        if mm == 0:
            return 3
        # (Cout = 64 — level 0 — stays on the 16-bit Winograd kernel there: on the direct kernel's 32-channel waves it is 1 % faster end to end, but
        # the product rounded once to 16 bits takes the fp16 schemes' logits past their quarter-margin bar: measured, fp16ref32 `ok` true -> false)
        return mm if co % 128 == 0 else 0

    def _stack16_mode(self, rows_per_query, co, ci):
        """conv16 math mode of a stack layer (see _product16_mode), or 0."""
        if not (STACK16 and rows_per_query % 128 == 0):
            return 0
        mm = ops.MATH_MODE
        if mm == 0:
            return 3 if (co % 64 == 0 and ci % 32 == 0) else 0
        return mm if (co % 128 == 0 and ci % 64 == 0) else 0

    def _product16_filters(self, l, li, wgt, mode):
        cache = self.__dict__.setdefault("_prod16", {})
        key = (l, li, mode, wgt.data_ptr())
        if key not in cache:
            cache[key] = ops.conv16_pack(wgt, mode, layout=1)
        return cache[key]

    def _query_batch(self, que_imgs):
        """que_imgs [qn,3,128,128], qn <= MAX_BATCH (32) -> logits [qn,rfn], angles [qn,rfn]; one set of launches for the whole batch
        (the reference cache is streamed once per batch, the qn*D hypothesis images fill the conv grids)."""
        pk = self._pack()
        an = self.an
        qn = que_imgs.shape[0]
        rfn_all, rfn = self.rfn, self.r_end - self.r_begin                  # global / local reference counts
        D, Dg = rfn * an, rfn_all * an
        dev = que_imgs.device
        grp = D if qn > 1 else 0
        ops.stats_arena_begin(dev)
        qf = self.get_feats(que_imgs)
        cat = torch.empty((qn * D, 1, 4, 4, 768), dtype=torch.float32, device=dev)
        # score maps -> viewpoint scores and the product's InstanceNorm statistics of all three levels: one streaming launch
        caches = [c.view(c.shape[0], c.shape[2] * c.shape[3], 512) for c in self.ref_feats_cache]
        vps, psc, psh, _ = ops.selector_levels([qf[l].view(qn, -1, 512) for l in range(3)], caches, self.ref_sums, Dg)  # [qn,3,D], [qn,3,512]
        levels = [self._level(l, qf[l], cat, psc[:, l].contiguous(), psh[:, l].contiguous(), qn) for l in range(3)]
        if not self.sharded:
            ops.fork_join([(lambda g=g: sum(1 for _ in g)) for g in levels], dev)       # nothing is yielded: the levels run side by side
        else:
            # sharded: the three levels advance in lock-step on ONE stream (collectives must be issued in the same order on every
            # rank); the InstanceNorm sums that the levels reach in the same round share one all-reduce: 5 rounds (5 / 3 / 1
            # InstanceNorms per level) instead of 9 collectives, each carrying the [qn, C, 2] tables of the whole query batch
            while levels:
                pending, alive = [], []
                for g in levels:
                    st = next(g, None)
                    if st is not None:
                        pending.append(st); alive.append(g)
                if pending:
                    self._allreduce(pending)
                levels = alive

        with self._mm("fuse"):
            # corr_feats_conv: 1x1x1 768->512, IN3d, ReLU, (AvgPool commuted) 512->512   selector.py:71-77,197-200
            y = torch.empty((qn * D, 1, 4, 4, 512), dtype=torch.float32, device=dev)
            st = ops.new_stats(qn, 512, dev)
            if not self.sharded:
                sc, sh = ops.conv(cat, pk["fuse0"][0], pk["fuse0"][1], y, stats=st, finalize=Dg * 16, rows_per_group=grp * 16)
            else:
                ops.conv(cat, pk["fuse0"][0], pk["fuse0"][1], y, stats=st, rows_per_group=grp * 16)
                self._allreduce([st])
                sc, sh = ops.stats_finalize(st, Dg * 16)
            pooled = torch.empty((qn * D, 1, 1, 1, 512), dtype=torch.float32, device=dev)
            ops.affine_act_pool(y, pooled, sc, sh, per_n=grp, relu=True, pool=2)
            feats = torch.zeros((qn * D, FEAT_LD), dtype=torch.float32, device=dev)
            ops.conv(pooled.view(1, 1, 1, qn * D, 512), pk["fuse3"][0], pk["fuse3"][1], feats.view(1, 1, 1, qn * D, FEAT_LD)[..., :512])
        if not self.sharded:
            ops.vps_norm(vps, feats, 512)                                               # selector.py:201-202
        else:                                                                           # norm over ALL hypotheses
            # one all-gather for the batch: row r = the (an, qn, 3) scalars of local reference r, in global reference order
            rows = vps.reshape(qn, 3, rfn, an).permute(2, 3, 0, 1).reshape(rfn, an * qn * 3).contiguous()
            vall = self._allgather_rows(rows, rfn_all).view(rfn_all, an, qn, 3).permute(2, 3, 0, 1).reshape(qn, 3, Dg).contiguous()
            fall = torch.zeros((qn * Dg, FEAT_LD), dtype=torch.float32, device=dev)
            ops.vps_norm(vall, fall, 512)
            feats.view(qn, D, FEAT_LD)[:, :, 512:515] = fall.view(qn, Dg, FEAT_LD)[:, self.r_begin * an:self.r_end * an, 512:515]

        with self._mm("tail"):
            # score_process + max over rotations + viewpoint embedding                     selector.py:204-205
            t0 = torch.empty((1, 1, 1, qn * D, 512), dtype=torch.float32, device=dev)
            ops.conv(feats.view(1, 1, 1, qn * D, FEAT_LD), pk["sp0"][0], pk["sp0"][1], t0, out_act=1)
            t1 = torch.empty_like(t0)
            ops.conv(t0, pk["sp2"][0], pk["sp2"][1], t1)
            xl = torch.empty((qn * rfn, 512), dtype=torch.float32, device=dev)
            ops.max_an_add(t1.view(qn * D, 512), rfn, an, self.ref_pose_embed[self.r_begin:self.r_end].contiguous(), xl, batch=qn)
            feats_l, rfn_l = feats, rfn
            rfn = rfn_all                                                                   # the tail runs on ALL refs
            xm = torch.empty((qn * rfn, 1024), dtype=torch.float32, device=dev)             # [x | msg]
            xm[:, :512] = self._allgather_batch(xl, qn, rfn_l, rfn_all)

            def tok(t, n_tok=None):      # [qn*n, C] (row-strided) -> conv view [qn,1,1,n,C]: one image per query
                n_tok = rfn if n_tok is None else n_tok
                return t.as_strided((qn, 1, 1, n_tok, t.shape[1]), (n_tok * t.stride(0), 0, 0, t.stride(0), 1), t.storage_offset())

            pn = 1 if qn > 1 else 0                      # InstanceNorm1d tables per query (image of the tok view)
            for i in range(2):                                                              # selector.py:207-209
                a = pk["att"][i]
                qkv = torch.empty((qn * rfn, 1536), dtype=torch.float32, device=dev)
                ops.conv(tok(xm[:, :512]), a["qkv"][0], a["qkv"][1], tok(qkv))
                att = torch.empty((qn * rfn, 512), dtype=torch.float32, device=dev)
                ops.attention(qkv[:, 0:512], qkv[:, 512:1024], qkv[:, 1024:1536], 8, att, batch=qn)
                mrg = torch.empty((qn * rfn, 512), dtype=torch.float32, device=dev)
                ops.conv(tok(att), a["merge"][0], a["merge"][1], tok(mrg))
                ops.layernorm(mrg, a["ln"][0], a["ln"][1], xm[:, 512:])
                y0 = torch.empty((qn * rfn, 512), dtype=torch.float32, device=dev)
                s0 = ops.new_stats(qn, 512, dev)
                sc0, sh0 = ops.conv(tok(xm), a["mlp0"][0], a["mlp0"][1], tok(y0), stats=s0, finalize=rfn, rows_per_group=pn * rfn)
                y1 = torch.empty((qn * rfn, 512), dtype=torch.float32, device=dev)
                s1 = ops.new_stats(qn, 512, dev)
                sc1, sh1 = ops.conv(tok(y0), a["mlp3"][0], a["mlp3"][1], tok(y1), in_scale=sc0, in_shift=sh0, in_relu=True, per_n=pn, stats=s1,
                                    finalize=rfn, rows_per_group=pn * rfn)
                xn = torch.empty((qn * rfn, 1024), dtype=torch.float32, device=dev)
                ops.affine_act_add(y1, xn[:, :512], sc1, sh1, relu=True, residual=xm[:, :512], rows_per_group=pn * rfn)
                xm = xn
            p0 = torch.empty((qn * rfn, 512), dtype=torch.float32, device=dev)
            ops.conv(tok(xm[:, :512]), pk["pred0"][0], pk["pred0"][1], tok(p0), out_act=1)
            logits = torch.empty((qn * rfn, 1), dtype=torch.float32, device=dev)
            ops.conv(tok(p0), pk["pred2"][0], pk["pred2"][1], tok(logits))

            # angle head on the per-reference rows [an*516]                                 selector.py:212-214
            a0 = torch.empty((qn * rfn_l, 512), dtype=torch.float32, device=dev)
            ops.conv(tok(feats_l.view(qn * rfn_l, an * FEAT_LD), rfn_l), pk["ang0"][0], pk["ang0"][1], tok(a0, rfn_l), out_act=1)
            a1 = torch.empty_like(a0)
            ops.conv(tok(a0, rfn_l), pk["ang2"][0], pk["ang2"][1], tok(a1, rfn_l), out_act=1)
            angles = torch.empty((qn * rfn_l, 1), dtype=torch.float32, device=dev)
            ops.conv(tok(a1, rfn_l), pk["ang4"][0], pk["ang4"][1], tok(angles, rfn_l))
        return logits.view(qn, rfn), self._allgather_batch(angles, qn, rfn_l, rfn_all).view(qn, rfn_all)

    def compute_view_point_feats(self, *a, **k):
        """cfg key 'math_mode' ('bf16' / 'fp16'; default fp32) selects the matrix-core operand precision of this network's conv /
        correlation launches; absent, an enclosing `ops.math_mode(...)` context applies."""
        with ops.math_mode(self.cfg.get("math_mode"), inherit_if_none=True):
            return self._compute_view_point_feats_fp(*a, **k)

    def _compute_view_point_feats_fp(self, que_imgs):
        """que_imgs [qn,3,h,w] in [0,1] -> logits [qn,rfn], angles [qn,rfn] (reference selector.py:177-215)."""
        # the queries of a chunk share every launch (and, sharded, every collective); a chunk's largest tensor — the first conv's output
        # [qn * D, h0, w0, 64] — has to stay below the 2^31-byte reach of the kernels' buffer loads (64 x 36 rotations: 14 -> 8 queries)
        # (sharded: sized by the LARGEST shard, so that every rank cuts the same chunks and issues the same collectives — ADVICE r04)
        c0 = self.ref_feats_cache[0]
        d_max = -(-self.rfn // self.world) * self.an
        per_query = d_max * c0.shape[2] * c0.shape[3] * 64 * 4
        step = max(1, min(MAX_BATCH, ((1 << 31) - 1) // per_query))
        if step >= 8:
            step -= step % 8                           # whole groups of 8 for g6d_selector_levels
        outs = [self._query_batch(que_imgs[i:i + step].contiguous()) for i in range(0, que_imgs.shape[0], step)]
        return ops.cat1([o[0] for o in outs], 0), ops.cat1([o[1] for o in outs], 0)

    def forward(self, data):
        self.extract_ref_feats(data["ref_imgs"], data["ref_imgs_info"]["poses"], data["object_center"],
                               data["object_vert"], "eval" not in data)
        logits, angles = self.compute_view_point_feats(data["que_imgs_info"]["imgs"])
        return {"ref_vp_logits": logits, "angles_pr": angles}

    # ------------------------------------------------------------------ numpy API used by Gen6DEstimator
    def load_ref_imgs(self, ref_imgs, ref_poses, object_center, object_vert):
        """uint8 [an,rfn,h,w,3], [rfn,3,4], [3], [3] (reference selector.py:150-163)."""
        dev = self.device_()
        x = torch.from_numpy(np.ascontiguousarray(ref_imgs)).to(dev).float().div_(255).permute(0, 1, 4, 2, 3)
        f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dev)
        with torch.no_grad():
            self.extract_ref_feats(x, f(ref_poses), f(object_center), f(object_vert))

    def select_que_imgs(self, que_imgs):
        """uint8 [qn,h,w,3] -> {'ref_idx','angles','scores'} numpy; the raw network angle at the arg-max reference
        (reference selector.py:165-175)."""
        x = torch.from_numpy(np.ascontiguousarray(que_imgs)).to(self.device_()).float().div_(255).permute(0, 3, 1, 2)
        with torch.no_grad():
            logits, angles = self.compute_view_point_feats(x.contiguous())
            idx = torch.argmax(logits, 1)
            ang = angles[torch.arange(idx.shape[0], device=idx.device), idx]
        return {"ref_idx": idx.cpu().numpy(), "angles": ang.cpu().numpy(), "scores": logits.cpu().numpy()}
