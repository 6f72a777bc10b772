"""VGG-11-BN trunk.  BatchNorm (eval) is folded into the conv weights once.  Output taps follow the reference exactly,
including its quirk that the 1/16 output is the BatchNorm output of conv 25 WITHOUT the final ReLU
(reference network/pretrain_models.py:17-25,66-72,109-111).

`vgg_taps_cl` / `vgg_taps_cl_multi`: g6d_vgg_conv1_pool_nhwc for the 3->64 layer and the Winograd kernels (F(2x2,3x3), or F(4x4,3x3)
for the detector's pyramid, on fp32 MFMA; bias + ReLU + 2x2 max-pool fused) for the other seven, channels-last end to end, so the
features reach the correlation / similarity / volume kernels without any layout pass.  (The PyTorch-ROCm / MIOpen trunk that rounds
1-2 kept beside it for A/B measurements lives in tools/library_trunk.py.)"""
import torch

from .. import ops, specs

_POOL_BEFORE = (1, 2, 4, 6)          # positions (in the list of 8 convs) preceded by a 2x2 max-pool


def winograd_filters(w):
    """[Cout,Cin,3,3] -> U [Cin/8,16,Cout,8] with U[c][4a+b][co][k ^ (4 if co & 8 else 0)] = (G g G^T)[a][b] of filter
    (co, 8c+k): the operand layout of g6d_wino_conv3x3 (a block's slice of one 8-channel chunk is 16 contiguous runs)."""
    co, ci = w.shape[:2]
    if ci % 8 or co % 32:
        raise ValueError("winograd_filters: Cin % 8 == 0 and Cout % 32 == 0 expected")
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
    U = torch.einsum("ai,ocij,bj->ocab", G, w.double(), G).to(w.dtype)               # [co,ci,4,4]
    U = U.reshape(co, ci // 8, 8, 16).permute(1, 3, 0, 2).contiguous()               # [chunk][ab][co][8]
    # the kernel copies a block's rows straight into LDS (lane-linear): rows with co & 8 carry their two 4-channel halves
    # swapped so that the fragment reads (one 16-byte half per lane) spread over all banks
    swap = (torch.arange(co, device=w.device) & 8) != 0
    U[:, :, swap] = torch.cat([U[:, :, swap, 4:], U[:, :, swap, :4]], -1)
    return U


def winograd_filters16(w, dtype):
    """[Cout,Cin,3,3] -> U16 [Cin/16,16,Cout,16] in `dtype` (torch.float16 / torch.bfloat16) for g6d_wino16_conv3x3_multi: the
    Winograd-domain filters (G g G^T, computed in fp64) ROUNDED to the operand type, a chunk = 16 input channels, rows with co & 8
    carry their two 8-channel halves swapped (the same LDS swizzle as winograd_filters)."""
    co, ci = w.shape[:2]
    if ci % 16 or co % 64:
        raise ValueError("winograd_filters16: Cin % 16 == 0 and Cout % 64 == 0 expected")
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
    U = torch.einsum("ai,ocij,bj->ocab", G, w.double(), G)                            # [co,ci,4,4]
    U = U.reshape(co, ci // 16, 16, 16).permute(1, 3, 0, 2).contiguous()              # [chunk][ab][co][16]
    swap = (torch.arange(co, device=w.device) & 8) != 0
    U[:, :, swap] = torch.cat([U[:, :, swap, 8:], U[:, :, swap, :8]], -1)
    return U.to(dtype).contiguous()


def winograd_filters_taps(w_taps, kd=1):
    """[Cout, kd*9, Cin] (the tap-major layout of ParamBank.conv_w) -> [kd*Cin/8, 16, Cout, 8]: one winograd_filters block per
    depth tap, stacked along the chunk axis (the K loop of the kernel walks depth taps outermost)."""
    co, taps, ci = w_taps.shape
    assert taps == kd * 9
    w = w_taps.reshape(co, kd, 3, 3, ci).permute(1, 0, 4, 2, 3)              # [kd, co, ci, 3, 3]
    return torch.cat([winograd_filters(w[k].contiguous()) for k in range(kd)], 0).contiguous()


def winograd_filters16_taps(w_taps, kd, dtype):
    """[Cout, kd*9, Cin] -> [kd*Cin/16, 16, Cout, 16] in `dtype`: one winograd_filters16 block per depth tap (G6dConv.weight_wino16)."""
    co, taps, ci = w_taps.shape
    assert taps == kd * 9
    w = w_taps.reshape(co, kd, 3, 3, ci).permute(1, 0, 4, 2, 3)
    return torch.cat([winograd_filters16(w[k].contiguous(), dtype) for k in range(kd)], 0).contiguous()


def winograd_corr_filters(w_taps, k):
    """Correlation filters [Cout, k*k, Cin] (tap = ky*k + kx), k = 3*kb -> U [Cin/8 * kb*kb, 16, Cout, 8] for g6d_corr2d_wino_multi:
    the k x k filter cut into kb x kb blocks of 3x3 taps, each transformed like a trunk filter (winograd_filters), CHUNK-major
    (row c * kb*kb + b = 8-channel chunk c of block b = kb*bi + bj, which holds taps [3bi, 3bi+3) x [3bj, 3bj+3)): the kernel
    visits the kb*kb shifts of a chunk back to back."""
    co, taps, ci = w_taps.shape
    kb = k // 3
    assert taps == k * k and k == 3 * kb
    w = w_taps.reshape(co, kb, 3, kb, 3, ci).permute(1, 3, 0, 5, 2, 4)      # [bi, bj, co, ci, 3, 3]
    U = torch.stack([winograd_filters(w[bi, bj].contiguous()) for bi in range(kb) for bj in range(kb)], 1)      # [Cin/8, kb*kb, 16, Cout, 8]
    return U.reshape(-1, *U.shape[2:]).contiguous()


# ---- Winograd F(4x4,3x3) with the interpolation points (0, +-3/4, +-3/2, inf) of gen6d_amd/csrc/wino43_conv.hip
W43_A, W43_B = 0.75, 1.5


def winograd43_matrices(dtype=torch.float64, device=None):
    """(B^T [6,6], G [6,3], A^T [4,6]) of F(4,3) at the points (0, a, -a, b, -b, inf), a = 3/4, b = 3/2, normalised so that B^T has the
    monic rows the kernel evaluates (csrc/wino43_conv.hip header) and G carries the 1 / prod_{k != i} (p_i - p_k) factors."""
    a, b = W43_A, W43_B
    a2, b2 = a * a, b * b
    BT = torch.tensor([[a2 * b2, 0, -(a2 + b2), 0, 1, 0],
                       [0, -a * b2, -b2, a, 1, 0],
                       [0, a * b2, -b2, -a, 1, 0],
                       [0, -b * a2, -a2, b, 1, 0],
                       [0, b * a2, -a2, -b, 1, 0],
                       [0, a2 * b2, 0, -(a2 + b2), 0, 1]], dtype=dtype, device=device)
    pts = [0.0, a, -a, b, -b]
    rows = []
    for i, pi in enumerate(pts):
        n = 1.0
        for k, pk in enumerate(pts):
            if k != i:
                n *= pi - pk
        rows.append([1.0 / n, pi / n, pi * pi / n])
    rows.append([0.0, 0.0, 1.0])
    G = torch.tensor(rows, dtype=dtype, device=device)
    AT = torch.tensor([[1, 1, 1, 1, 1, 0],
                       [0, a, -a, b, -b, 0],
                       [0, a2, a2, b2, b2, 0],
                       [0, a2 * a, -a2 * a, b2 * b, -b2 * b, 1]], dtype=dtype, device=device)
    return BT, G, AT


def winograd43_filters(w):
    """[Cout,Cin,3,3] -> U43 [Cin/8, 2, Cout/CB, 18, CB/32, 4, 16, 4] (CB = 64, or 32 when Cout % 64) for g6d_wino43_conv3x3_multi:
    U43[c][b // 3][blk][3 a + b % 3][np][kg][lt][2 par + s] = (G g G^T)[a][b] of filter (co = CB blk + 32 np + 16 par + lt,
    ci = 8 c + 2 kg + s) — per 8-channel chunk the two column halves the kernel stages one after the other, each half of a CB-channel
    block ONE contiguous run in the order of its LDS image (a lane reads the 16 bytes at [position][np][kg][lt]: its B operands for
    the channel tiles 2 np and 2 np + 1).  Computed in fp64."""
    co, ci = w.shape[:2]
    if ci % 8 or co % 32:
        raise ValueError("winograd43_filters: Cin % 8 == 0 and Cout % 32 == 0 expected")
    cb = 32 if co % 64 else 64
    _, G, _ = winograd43_matrices(device=w.device)
    U = torch.einsum("ai,ocij,bj->ocab", G, w.double(), G).to(w.dtype)               # [co,ci,6,6]
    U = U.reshape(co // cb, cb // 32, 2, 16, ci // 8, 4, 2, 6, 2, 3)                 # blk, np, par, lt, c, kg, s, a, half, b3
    U = U.permute(4, 8, 0, 7, 9, 1, 5, 3, 2, 6)                                      # c, half, blk, a, b3, np, kg, lt, par, s
    return U.reshape(ci // 8, 2, co // cb, 18, cb // 32, 4, 16, 4).contiguous()


def winograd43_filters_taps(w_taps, kd=1):
    """[Cout, kd*9, Cin] (ParamBank.conv_w) -> [kd*Cin/8, 2, Cout/CB, 18, CB/32, 4, 16, 4]: one winograd43_filters block per depth tap, depth taps outermost
    (G6dConv.weight_wino43)."""
    co, taps, ci = w_taps.shape
    assert taps == kd * 9
    w = w_taps.reshape(co, kd, 3, 3, ci).permute(1, 0, 4, 2, 3)
    return torch.cat([winograd43_filters(w[k].contiguous()) for k in range(kd)], 0).contiguous()


def winograd43_corr_filters(w_taps, k):
    """Correlation filters [Cout, k*k, Cin], k = 3*kb -> U43 [Cin/8 * kb*kb, 2, Cout/CB, 18, CB/32, 4, 16, 4] for g6d_corr2d_wino43_multi: CHUNK-major
    like winograd_corr_filters (row c * kb*kb + b = 8-channel chunk c of block b = kb*bi + bj)."""
    co, taps, ci = w_taps.shape
    kb = k // 3
    assert taps == k * k and k == 3 * kb
    w = w_taps.reshape(co, kb, 3, kb, 3, ci).permute(1, 3, 0, 5, 2, 4)      # [bi, bj, co, ci, 3, 3]
    U = torch.stack([winograd43_filters(w[bi, bj].contiguous()) for bi in range(kb) for bj in range(kb)], 1)      # [Cin/8, kb*kb, 2, Cout/CB, 18, CB/32, 4, 16, 4]
    return U.reshape(-1, *U.shape[2:]).contiguous()


def winograd43_corr_filters_padded(w_taps, k):
    """Correlation filters [Cout, k*k, Cin] with k NOT a multiple of 3 (the detector's 7x7 level): zero-extended symmetrically to the
    next multiple of 3 (9x9: one tap on every side — the "same" correlation is unchanged) and cut into blocks like
    winograd43_corr_filters.  Returns (U43, kblocks)."""
    co, taps, ci = w_taps.shape
    kb = (k + 2) // 3
    e = 3 * kb - k
    if taps != k * k or e % 2:
        raise ValueError("winograd43_corr_filters_padded: odd k expected")
    w = torch.nn.functional.pad(w_taps.reshape(co, k, k, ci), (0, 0, e // 2, e // 2, e // 2, e // 2))
    return winograd43_corr_filters(w.reshape(co, 9 * kb * kb, ci).contiguous(), 3 * kb), kb


class TrunkLayer(tuple):
    """(U, bias) of a Winograd trunk layer as the fp32 kernel takes them; `.u16(dtype)` = the 16-bit filters of the reduced-precision
    kernel, built from the folded fp32 weights on first use (ops.MATH_MODE 1 / 2)."""

    def __new__(cls, U, b, w):
        t = super().__new__(cls, (U, b))
        t._w, t._u16, t._u43 = w, {}, None
        return t

    def u43(self):
        """The layer's filters for the F(4x4,3x3) kernel (winograd43_filters), built on first use."""
        if self._u43 is None:
            self._u43 = winograd43_filters(self._w)
        return self._u43

    def u16(self, dtype):
        if dtype not in self._u16:
            self._u16[dtype] = winograd_filters16(self._w, dtype)
        return self._u16[dtype]

    def w16(self, mode):
        """The layer's (BatchNorm-folded) filters for the direct kernel on 16-bit activations (ops.conv16_pack, fragment-major): mode 1 / 2 =
        rounded to bf16 / fp16 (the reduced-precision mode), 3 = fp16 hi / lo pairs (the fp32 path's split-precision trunk)."""
        key = ("direct", mode)
        if key not in self._u16:
            co, ci = self._w.shape[:2]
            # bf16 / fp16: both operand tiles through LDS (measured 9 % faster than filters in registers: the register variant pulls every
            # filter fragment once per wave instead of once per block, profiles/r06_conv16_bench.md); pairs: fragment-major filters in registers
            self._u16[key] = ops.conv16_pack(self._w.permute(0, 2, 3, 1).reshape(co, 9, ci).contiguous(), mode, layout=1)
        return self._u16[key]


_LOWP_DTYPE = {1: torch.bfloat16, 2: torch.float16}
LOWP_TRUNK = True        # reduced-precision mode: trunk on the 16-bit Winograd kernel (False, set by tools / tests: stays on the fp32 kernel)
CONV16_TRUNK = True      # reduced-precision mode (round 6): trunk on 16-bit ACTIVATIONS and the direct 16-bit kernel (g6d_conv16_direct_multi:
                         # 1.5-1.8x the 16-bit Winograd kernel per layer); False (tools / tests): the round-5 path above
SPLIT16_TRUNK = True     # fp32 path (round 6): the trunks that ran on the F(4x4,3x3) kernel (detector pyramid, refiner crops) on the SAME direct
                         # kernel with every operand a pair of fp16 values (hi + lo, 22+ significand bits; 3 MFMAs per product on the 16-bit
                         # matrix cores instead of the fp32 ones): fp32-class results — smaller error than F(4x4,3x3)'s — at 1.3-1.6x its speed.
                         # False (tools / tests): the F(4x4,3x3) kernel of rounds 4-5
SPLIT16_ALWAYS = False   # ... and EVERY fp32 trunk call whose map sizes the kernel takes (the selector's query crops, the refiner's crops at
                         # any batch size, the reference-side trunks at build time): the trunk's kernel is then a function of the layer, not
                         # of how many queries share the launch, and its error (2e-6 of range) is below both Winograd kernels'.
                         # False (default): only where F(4x4,3x3) ran — measured neutral on the batched step (270.9 vs 270.2 images/s) and
                         # 0.2 ms slower on a single query (7.84 vs 7.62 ms: a crop's 224 tiles do not fill the chip)


def _wino_layer(xs, layer, relu=True, full=True, pool=False, f43=False):
    """One trunk layer over the segments xs on the kernel of the current math mode: fp32 Winograd — F(2x2,3x3), or with f43 the
    F(4x4,3x3) kernel (1.78x fewer multiplications at ~5x the rounding error: the detector's pyramid, whose parity budget has the
    room) — or (ops.MATH_MODE 1 / 2, Cin % 16 == 0) the 16-bit one."""
    mm = ops.MATH_MODE
    if mm and LOWP_TRUNK and hasattr(layer, "u16") and xs[0].shape[3] % 16 == 0:
        return ops.wino16_conv3x3_multi(xs, layer.u16(_LOWP_DTYPE[mm]), layer[1], relu=relu, full=full, pool=pool)
    if f43 and not mm and layer[1].numel() % 64 == 0:
        return ops.wino43_conv3x3_multi(xs, layer.u43(), layer[1], relu=relu, full=full, pool=pool)
    return ops.wino_conv3x3_multi(xs, layer[0], layer[1], relu=relu, full=full, pool=pool)


def pack_trunk(folded):
    """fold_vgg(...) output -> what the trunk consumes: the first layer's (w, b) and a TrunkLayer per Winograd layer."""
    return [folded[0]] + [TrunkLayer(winograd_filters(w), b, w) for w, b in folded[1:]]


_IMG_NORM = (tuple(specs.IMAGENET_MEAN), tuple(specs.IMAGENET_STD))


def vgg_taps_cl(packed, x, taps, norm=None, f43=False):
    """Own trunk, channels-last: x [n,3,h,w] normalised image (or an image in [0,1] with norm = (mean, std): the first layer
    normalises while it stages its input) -> {'c3': [n,h/4,w/4,256] post-ReLU, 'c5': [n,h/8,w/8,512] post-ReLU,
    'c7_pre': [n,h/16,w/16,512] pre-ReLU, 'p7': max-pool of c7_pre} (only the requested taps + c7_pre)."""
    if (ops.MATH_MODE and LOWP_TRUNK) or f43 or (SPLIT16_TRUNK and SPLIT16_ALWAYS and hasattr(packed[1], "w16") and _conv16_eligible([x], taps)):
        # reduced precision: the multi-segment 16-bit kernel; f43: the F(4x4,3x3) kernel (one segment); fp32 path: the split-precision kernel
        return vgg_taps_cl_multi(packed, [x], taps, norm=norm, f43=f43)[0]
    w0, b0 = packed[0]
    x = ops.vgg_conv1_pool_nhwc(x.contiguous(), w0, b0, norm=norm)              # (normalise +) conv0 + ReLU + pool
    _, x = ops.wino_conv3x3(x, *packed[1], relu=True, full=False, pool=True)    # conv1 + ReLU + pool
    x, _ = ops.wino_conv3x3(x, *packed[2], relu=True)                           # conv2
    c3, x = ops.wino_conv3x3(x, *packed[3], relu=True, full="c3" in taps, pool=True)
    x, _ = ops.wino_conv3x3(x, *packed[4], relu=True)
    c5, x = ops.wino_conv3x3(x, *packed[5], relu=True, full="c5" in taps, pool=True)
    x, _ = ops.wino_conv3x3(x, *packed[6], relu=True)
    c7, p7 = ops.wino_conv3x3(x, *packed[7], relu=False, full=True, pool="p7" in taps)   # BN output WITHOUT the last ReLU
    out = {"c3": c3, "c5": c5, "c7_pre": c7, "p7": p7}
    return {k: v for k, v in out.items() if v is not None and (k in taps or k == "c7_pre")}


def _conv16_eligible(xs, taps):
    """The 16-bit activation path pools whole 2x2 windows only: every pooled layer's map must have even sides (the detector's pyramid
    sizes are multiples of 32, the crops 128: always true there; other sizes keep the round-5 path, whose kernels pool with floor)."""
    need = 32 if "p7" in taps else 16
    return all(x.shape[2] % need == 0 and x.shape[3] % need == 0 for x in xs) and len(xs) <= 4


def _vgg_taps_conv16(packed, xs, taps, norm, mode, taps16=()):
    """The trunk on 16-bit activations (round 6): from the first layer's epilogue on the maps are fp16 / bf16 channels-last (mode 1 / 2:
    the reduced-precision mode) or fp16 hi / lo PAIRS (mode 3: the fp32 path), the seven 3x3 layers run on the direct kernel
    (DMA-staged activation tiles, filters in registers, v_mfma_f32_32x32x16), and only the requested taps are written in fp32 for their
    consumers."""
    w0, b0 = packed[0]
    t16 = "t16"
    f32 = torch.float32
    cur = [ops.vgg_conv1_pool_nhwc16(x.contiguous(), w0, b0, norm=norm, mode=mode) for x in xs]       # conv0 + ReLU + pool, per size

    def layer(i, cur, relu=True, full=None, pool=None):
        return ops.conv16_direct_multi(cur, packed[i].w16(mode), packed[i][1], relu=relu, full=full, pool=pool)

    _, cur = layer(1, cur, pool=t16)
    cur, _ = layer(2, cur, full=t16)
    c3, cur = layer(3, cur, full=f32 if "c3" in taps else None, pool=t16)
    cur, _ = layer(4, cur, full=t16)
    # (taps16: taps handed over in the kernel's own 16-bit / pair format — the detector's 15x15 correlation reads c5 that way)
    c5, cur = layer(5, cur, full=(t16 if "c5" in taps16 else f32) if "c5" in taps else None, pool=t16)
    cur, _ = layer(6, cur, full=t16)
    c7, p7 = layer(7, cur, relu=False, full=t16 if "c7_pre" in taps16 else f32, pool=f32 if "p7" in taps else None)
    outs = []
    for i in range(len(xs)):
        d = {"c3": c3[i], "c5": c5[i], "c7_pre": c7[i], "p7": p7[i]}
        outs.append({k: v for k, v in d.items() if v is not None and (k in taps or k == "c7_pre")})
    return outs


def vgg_taps_cl_multi(packed, xs, taps, norm=None, f43=False, taps16=()):
    """vgg_taps_cl for several image sizes at once (the scales of the detector's pyramid): every Winograd layer is ONE launch
    over all sizes (ops.wino_conv3x3_multi).  xs: list of [1,3,h_i,w_i] images (normalised, or in [0,1] with norm) -> list of
    tap dicts.  f43: the seven Winograd layers on the F(4x4,3x3) kernel (fp32 mode only)."""
    if ops.MATH_MODE and LOWP_TRUNK and CONV16_TRUNK and hasattr(packed[1], "w16") and _conv16_eligible(xs, taps):
        return _vgg_taps_conv16(packed, xs, taps, norm, ops.MATH_MODE, taps16)
    if (f43 or SPLIT16_ALWAYS) and not ops.MATH_MODE and SPLIT16_TRUNK and hasattr(packed[1], "w16") and _conv16_eligible(xs, taps):
        return _vgg_taps_conv16(packed, xs, taps, norm, 3, taps16)
    w0, b0 = packed[0]
    dev = xs[0].device
    cur = ops.alloc_like_segments([(x.shape[0], x.shape[2] // 2, x.shape[3] // 2, w0.shape[0]) for x in xs], dev)
    for x, o in zip(xs, cur):
        ops.vgg_conv1_pool_nhwc(x.contiguous(), w0, b0, out=o, norm=norm)               # conv0 + ReLU + pool, per size
    _, cur = _wino_layer(cur, packed[1], relu=True, full=False, pool=True, f43=f43)
    cur, _ = _wino_layer(cur, packed[2], relu=True, f43=f43)
    c3, cur = _wino_layer(cur, packed[3], relu=True, full="c3" in taps, pool=True, f43=f43)
    cur, _ = _wino_layer(cur, packed[4], relu=True, f43=f43)
    c5, cur = _wino_layer(cur, packed[5], relu=True, full="c5" in taps, pool=True, f43=f43)
    cur, _ = _wino_layer(cur, packed[6], relu=True, f43=f43)
    c7, p7 = _wino_layer(cur, packed[7], relu=False, full=True, pool="p7" in taps, f43=f43)
    outs = []
    for i in range(len(xs)):
        d = {"c3": c3, "c5": c5, "c7_pre": c7, "p7": p7}
        outs.append({k: v[i] for k, v in d.items() if v is not None and (k in taps or k == "c7_pre")})
    return outs


def trunk_features_multi(packed, imgs_list, keys, f43=False, taps16=()):
    """trunk_features (no L2 normalisation) for a list of [1,3,h_i,w_i] images of different sizes -> list of lists of
    [1,1,h_l,w_l,C] maps.  One launch per layer for all sizes (up to 4 per launch).  taps16: keys that may come back in the 16-bit
    activation format of the direct kernel ([n,h,w,C] fp16 / bf16, or [n,h,w,2,C] fp16 pairs on the fp32 path) when the trunk runs on it —
    the caller checks the dtype."""
    if len(imgs_list) > 4:
        return [trunk_features(packed, im, keys, False) for im in imgs_list]
    taps = vgg_taps_cl_multi(packed, imgs_list, set(keys), norm=_IMG_NORM, f43=f43, taps16=taps16)
    return [[(t[k] if t[k].dtype != torch.float32 else t[k].unsqueeze(1)) for k in keys] for t in taps]


def trunk_features(packed, imgs, keys, l2norm, f43=False):
    """Normalised images [n,3,h,w] in [0,1] -> channels-last 5-D feature maps [n,1,h_l,w_l,C] for `keys`, optionally
    L2-normalised over C (F.normalize, reference selector.py:118 / refiner.py:69-71)."""
    t = vgg_taps_cl(packed, imgs, set(keys), norm=_IMG_NORM, f43=f43)
    outs = []
    for k in keys:
        f = t[k]
        if l2norm:
            ops.l2norm_rows(f)                  # in place: a tap is never the input of a later layer
        outs.append(f.unsqueeze(1))
    return outs
