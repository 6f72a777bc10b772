"""Tensor-level wrappers over the C ABI (include/gen6d_hip.h).

Every function takes PyTorch tensors that live on the GPU, checks layout, and enqueues the HIP kernel on torch's
current stream through ctypes.  PyTorch is only the owner of device memory and streams here.  Channel slices of wider
buffers are expressed as ordinary tensor views: an activation is a 5-D view [N, D, H, W, C] whose last stride is 1
and whose other strides are those of a dense channels-last buffer with channel stride ``ld = view.stride(3)``.

The host networks call these through the module (``ops.conv(...)``), which lets the CPU test-suite substitute the
per-op PyTorch references of tests/ref_ops.py to validate host orchestration without a GPU; the product path itself
has no fallback — `lib.load()` raises if libgen6d_hip.so is absent.
"""
import ctypes as C

import torch

from . import lib as _lib

_WS = {}
WORKSPACE_BYTES = 96 << 20
# Matrix-core operand precision of the conv / correlation kernels: 0 = fp32 (default, the parity path), 1 = bf16, 2 = fp16
# (fp32 accumulation, fp32 InstanceNorm statistics).  Opt-in speed mode: set through `math_mode(...)` / the networks'
# cfg key "math_mode"; graded separately from the fp32 path (arg-max equality + reported logit error).
MATH_MODE = 0
_MATH_NAMES = {"fp32": 0, "f32": 0, "bf16": 1, "fp16": 2, "f16": 2, 0: 0, 1: 1, 2: 2, None: 0}


class math_mode:
    """Context manager: `with ops.math_mode("bf16"): ...` runs the enclosed launches (and graph captures) in that mode."""

    def __init__(self, mode, inherit_if_none=False):
        self.mode = None if (mode is None and inherit_if_none) else _MATH_NAMES[mode]

    def __enter__(self):
        global MATH_MODE
        self.prev = MATH_MODE
        if self.mode is not None:
            MATH_MODE = self.mode
        return self

    def __exit__(self, *exc):
        global MATH_MODE
        MATH_MODE = self.prev
        return False
# When set to a list, conv() brackets every g6d_conv_igemm launch with HIP events recorded on the launch stream and
# appends (algorithmic flops, start, end); bench.py turns this into the roofline entry.
PROFILE = None
# Same for the HBM-bound kernels: a dict name -> list of (algorithmic bytes, start, end).
PROFILE_HBM = None


def _timed_hbm(name, nbytes, launch):
    if PROFILE_HBM is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    PROFILE_HBM.setdefault(name, []).append((float(nbytes), e0, e1))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gen6d_amd.ops: tensors must live on the GPU (HIP kernels only, no CPU fallback)")


def marker(i):
    """Empty kernel that brackets a region in a rocprofv3 kernel trace."""
    _lib.check(_lib.load().g6d_marker(int(i), _stream()), "g6d_marker")


def workspace(device):
    """Split scratch, one buffer per (device, stream): launches on different streams may overlap.  Zero-filled once: its head
    holds the per-tile arrival counters of the split launches, which every launch leaves at zero (include/gen6d_hip.h)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in _WS:
        _WS[key] = torch.zeros(WORKSPACE_BYTES // 4, dtype=torch.float32, device=device)
    return _WS[key]


_SIDE = {}
SERIAL = False      # True: fork_join runs its branches back to back on the current stream (per-kernel timing runs)


def fork_join(fns, device):
    """Run independent launch sequences concurrently: fns[0] on the current stream, the others on side streams cached per
    (device, current stream), forked from it and joined back (capturable in a hipGraph as parallel branches).  Independent stages of the path
    (detector scales, selector pyramid levels, refiner feature branches) are small grids that do not fill 256 CUs."""
    if SERIAL:
        return [fn() for fn in fns]
    main = torch.cuda.current_stream(device)
    # side streams (and with them the per-stream split-K workspaces and statistics arenas) belong to ONE launching stream:
    # queries captured / enqueued on different lanes must never share them, or concurrent replays race on the scratch
    streams = _SIDE.setdefault((str(device), main.cuda_stream), [])
    while len(streams) < len(fns) - 1:
        streams.append(torch.cuda.Stream(device=device))
    results = [None] * len(fns)
    for i, fn in enumerate(fns[1:]):
        s = streams[i]
        s.wait_stream(main)
        with torch.cuda.stream(s):
            workspace(device)                    # make sure the per-stream scratch exists
            results[i + 1] = fn()
    results[0] = fns[0]()
    for i in range(len(fns) - 1):
        main.wait_stream(streams[i])
    return results


def _cl5(t, name):
    """Validate a channels-last 5-D view; return (N, D, H, W, C, ld)."""
    if t.dim() != 5 or t.dtype != torch.float32:
        raise ValueError(f"{name}: expected float32 [N,D,H,W,C], got {tuple(t.shape)} {t.dtype}")
    N, D, H, W, Cc = t.shape
    ld = t.stride(3)
    s = t.stride()
    ok = s[4] == 1 and ld >= Cc
    exp = (D * H * W * ld, H * W * ld, W * ld, ld)
    for dim, (got, want) in enumerate(zip(s[:4], exp)):
        if t.shape[dim] > 1 and got != want:
            ok = False
    if not ok:
        raise ValueError(f"{name}: not a channels-last view (shape {tuple(t.shape)}, strides {s})")
    return N, D, H, W, Cc, ld


# InstanceNorm finalisation by the last block of the producing conv (G6dConv.fin_*).  False (set by tools / tests for A/B runs): a
# separate g6d_stats_finalize launch after the conv
FUSED_FINALIZE = True


def conv(x, w, bias, out, ksize=(1, 1, 1), stride=(1, 1, 1), pad=(0, 0, 0), mul=None, in_scale=None, in_shift=None,
         in_relu=False, per_n=False, out_act=0, stats=None, rows_per_group=0, split_k=0, w_wino=None, finalize=None, eps=1e-5,
         in_mod=0, mul_group=0, w_wino43=None):
    """Implicit-GEMM convolution (g6d_conv_igemm). x [N,Di,Hi,Wi,Cin], w [Cout,taps,Cin], out [N,Do,Ho,Wo,Cout] views.
    stats [G,Cout,2] fp64 (zeroed): per-(group, channel) sum / sum of squares of the output are accumulated into it.
    finalize=count: additionally turn the completed statistics into the affine of the following InstanceNorm (count values
    per group) inside the same launch and return (scale, shift) [G,Cout] instead of out.
    per_n: False / 0 = one affine table; True / 1 = a table per image; k = a table per run of k images.
    Query batches (G6dConv.in_image_mod / mul_group_images): in_mod = k — x holds k images and output image n reads x[n % k];
    mul_group = k — mul is [N/k,Hi,Wi,Cin] and image n is multiplied by mul[n // k]."""
    _need_gpu(x, w, out)
    Nx, Di, Hi, Wi, Cin, ld_in = _cl5(x, "conv.x")
    No, Do, Ho, Wo, Cout, ld_out = _cl5(out, "conv.out")
    N = No if in_mod else Nx
    if in_mod and Nx != in_mod:
        raise ValueError("conv: with in_mod the input must hold exactly in_mod images")
    per_n = int(per_n)
    kd, kh, kw = ksize
    if tuple(w.shape) != (Cout, kd * kh * kw, Cin) or not w.is_contiguous():
        raise ValueError(f"conv.w: expected contiguous {(Cout, kd * kh * kw, Cin)}, got {tuple(w.shape)}")
    for i, (di, do, k, s, p) in enumerate(zip((Di, Hi, Wi), (Do, Ho, Wo), ksize, stride, pad)):
        if do != (di + 2 * p - k) // s + 1:
            raise ValueError(f"conv: output extent mismatch on axis {i}: in {di} k {k} s {s} p {p} -> out {do}")
    if No != N:
        raise ValueError("conv: batch mismatch")
    if mul is not None:
        want = ((N + mul_group - 1) // mul_group, Hi, Wi, Cin) if mul_group else (Hi, Wi, Cin)
        if tuple(mul.shape) != want or not mul.is_contiguous():
            raise ValueError(f"conv.mul: expected contiguous {want}")
    if w_wino is not None and (tuple(w_wino.shape) != (kd * (Cin // 8), 16, Cout, 8) or not w_wino.is_contiguous()):
        raise ValueError(f"conv.w_wino: expected contiguous {(kd * (Cin // 8), 16, Cout, 8)}, got {tuple(w_wino.shape)}")
    if w_wino43 is not None and (tuple(w_wino43.shape) != w43_shape(kd * (Cin // 8), Cout) or not w_wino43.is_contiguous()):
        raise ValueError(f"conv.w_wino43: expected contiguous {w43_shape(kd * (Cin // 8), Cout)}, got {tuple(w_wino43.shape)}")
    u16 = None
    if MATH_MODE and w_wino is not None and Cin % 16 == 0 and Cout % 64 == 0:
        # reduced-precision mode: the layer's Winograd filters rounded to the operand type, built once per (layer, type) and kept
        # on the fp32 filter tensor (a new weight pack drops both)
        cache = w_wino.__dict__.setdefault("_g6d_u16", {})
        if MATH_MODE not in cache:
            from .network.backbone import winograd_filters16_taps
            cache[MATH_MODE] = winograd_filters16_taps(w, kd, {1: torch.bfloat16, 2: torch.float16}[MATH_MODE])
        u16 = cache[MATH_MODE]
    ws = workspace(x.device)
    d = _lib.G6dConv(
        in_=x.data_ptr(), mul=mul.data_ptr() if mul is not None else None,
        in_scale=in_scale.data_ptr() if in_scale is not None else None,
        in_shift=in_shift.data_ptr() if in_shift is not None else None,
        weight=w.data_ptr(), bias=bias.data_ptr() if bias is not None else None, out=out.data_ptr(),
        stats=stats.data_ptr() if stats is not None else None, workspace=ws.data_ptr(), workspace_bytes=ws.numel() * 4,
        N=N, Di=Di, Hi=Hi, Wi=Wi, Cin=Cin, ld_in=ld_in, Do=Do, Ho=Ho, Wo=Wo, Cout=Cout, ld_out=ld_out,
        kd=kd, kh=kh, kw=kw, sd=stride[0], sh=stride[1], sw=stride[2], pd=pad[0], ph=pad[1], pw=pad[2],
        in_relu=int(in_relu), in_affine_per_n=int(per_n), out_act=int(out_act),
        stat_rows_per_group=int(rows_per_group), split_k=int(split_k), math_mode=int(MATH_MODE),
        weight_wino=w_wino.data_ptr() if w_wino is not None else None, in_image_mod=int(in_mod), mul_group_images=int(mul_group),
        weight_wino16=u16.data_ptr() if u16 is not None else None,
        weight_wino43=w_wino43.data_ptr() if (w_wino43 is not None and not MATH_MODE) else None)
    fin = None
    if finalize is not None:
        if stats is None:
            raise ValueError("conv: finalize needs stats")
        if FUSED_FINALIZE:
            G = stats.shape[0]
            fin = (torch.empty((G, Cout), dtype=torch.float32, device=x.device), torch.empty((G, Cout), dtype=torch.float32, device=x.device))
            cnt = new_counter(x.device)
            d.fin_scale, d.fin_shift, d.fin_counter = fin[0].data_ptr(), fin[1].data_ptr(), cnt.data_ptr()
            d.fin_count, d.fin_eps, d.fin_groups = float(finalize), float(eps), G
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.load().g6d_conv_igemm(C.byref(d), _stream()), "g6d_conv_igemm")
        e1.record()
        fam = _lib.load().g6d_conv_plan(C.byref(d))
        fl = 2.0 * N * Do * Ho * Wo * Cout * kd * kh * kw * Cin
        if fam == 4:          # narrow-output layer on the vector ALUs: bound by reading its input, booked with the HBM-bound kernels
            if PROFILE_HBM is not None:
                PROFILE_HBM.setdefault("conv_narrow", []).append((4.0 * (N * Di * Hi * Wi * Cin + w.numel() + N * Do * Ho * Wo * Cout), e0, e1))
        else:
            PROFILE.append((fl / 4 if fam == 3 else (fl / 2.25 if fam == 2 else fl), e0, e1,      # Winograd kernels: FLOPs executed in the transform domain
                        ("wino3x3 F43 " if fam == 3 else "wino3x3 " if fam == 2 else "") + f"conv N={N} in={Di}x{Hi}x{Wi}x{Cin} out={Do}x{Ho}x{Wo}x{Cout} k={kd}x{kh}x{kw} s={stride[0]}{stride[1]}{stride[2]}"
                        f"{' mul' if mul is not None else ''}{' aff' if in_scale is not None else ''}{' stats' if stats is not None else ''}",
                        # algorithmic bytes: every operand once (input images, multiplier maps, filters, output)
                        4.0 * ((in_mod or N) * Di * Hi * Wi * Cin + (mul.numel() if mul is not None else 0) + w.numel() + N * Do * Ho * Wo * Cout), fl))
    else:
        _lib.check(_lib.load().g6d_conv_igemm(C.byref(d), _stream()), "g6d_conv_igemm")
    if finalize is not None:
        return fin if fin is not None else stats_finalize(stats, finalize, eps)
    return out


def corr2d_patch(x, w, out, k):
    """Detector correlation with LDS patch reuse: x [1,1,H,W,Cin], w [Cout<=32, k*k, Cin], out [1,1,H,W,Cout]."""
    _need_gpu(x, w, out)
    N, D, H, W, Cin, ld_in = _cl5(x, "corr2d.x")
    _, _, Ho, Wo, Cout, ld_out = _cl5(out, "corr2d.out")
    if N * D != 1 or (Ho, Wo) != (H, W) or tuple(w.shape) != (Cout, k * k, Cin) or not w.is_contiguous():
        raise ValueError("corr2d_patch: shape mismatch")
    ws = workspace(x.device)
    flops = 2.0 * H * W * Cout * k * k * Cin
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().g6d_corr2d_patch(_ptr(x), H, W, Cin, ld_in, _ptr(w), Cout, k, k, _ptr(out), ld_out, _ptr(ws),
                                           ws.numel() * 4, int(MATH_MODE), _stream()), "g6d_corr2d_patch")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((flops, e0, e1, f"corr2d_patch in={H}x{W}x{Cin} out={Cout} k={k}x{k}", 4.0 * (H * W * (Cin + Cout) + w.numel())))
    return out


# reduced-precision mode: correlation on corr16_patch_kernel (False, set by tools / tests: corr_patch_kernel with 16-bit operands, one
# hand-over per tap)
CORR16 = True


def corr_filters16(w, k, dtype):
    """Correlation filters [Cout, k*k, Cin] (tap = ky*k + kx, Cin % 32 == 0, Cout <= 32) -> the unit-major 16-bit layout of
    g6d_corr2d_patch16_multi: [(Cin/32) * k units][k taps kx][32 co][40] (32 channels + 8 zeros per row; rows co >= Cout zero),
    flattened, followed by 1 KB of zeros."""
    Cout, taps, Cin = w.shape
    assert taps == k * k and Cin % 32 == 0 and Cout <= 32
    v = w.reshape(Cout, k, k, Cin // 32, 32).permute(3, 1, 2, 0, 4)                     # [chunk, ky, kx, co, 32]
    out = torch.zeros((Cin // 32, k, k, 32, 40), dtype=dtype, device=w.device)
    out[:, :, :, :Cout, :32] = v.to(dtype)
    return torch.cat([out.reshape(-1), torch.zeros(512, dtype=dtype, device=w.device)]).contiguous()


def corr2d_patch_multi(xs, w, outs, k):
    """corr2d_patch for several map sizes in ONE launch (the scales of the detector's pyramid against the same reference filters):
    xs[i] [N,1,H_i,W_i,Cin] and outs[i] [N,1,H_i,W_i,Cout] dense tensors (N = queries of the batch at that scale), each list cut
    from one buffer (alloc_like_segments)."""
    _need_gpu(w, *xs, *outs)
    if not 1 <= len(xs) <= 4 or len(outs) != len(xs):
        raise ValueError("corr2d_patch_multi: 1..4 maps")
    Cout, _, Cin = w.shape
    segs = (_lib.G6dCorrSeg * len(xs))()
    flops, sizes = 0.0, []
    for i, (x, o) in enumerate(zip(xs, outs)):
        N, D, H, W, Cx, ld_in = _cl5(x, "corr2d_multi.x")
        No, _, Ho, Wo, Co, ld_out = _cl5(o, "corr2d_multi.out")
        if D != 1 or No != N or (Ho, Wo) != (H, W) or Cx != Cin or Co != Cout or (N > 1 and not (x.is_contiguous() and o.is_contiguous())):
            raise ValueError("corr2d_patch_multi: shape mismatch (batched maps must be dense)")
        segs[i] = _lib.G6dCorrSeg(in_=x.data_ptr(), out=o.data_ptr(), H=H, W=W, ld_in=ld_in, ld_out=ld_out, N=N)
        flops += 2.0 * N * H * W * Cout * k * k * Cin
        sizes.append(f"{N}x{H}x{W}" if N > 1 else f"{H}x{W}")
    if tuple(w.shape) != (Cout, k * k, Cin) or not w.is_contiguous():
        raise ValueError("corr2d_patch_multi: filter shape mismatch")
    ws = workspace(w.device)
    # reduced-precision mode: the 16-bit kernel with all kw weight tiles of a unit staged at once (g6d_corr2d_patch16_multi); its
    # host-rounded, unit-major filters are built once per (filter tensor, type) and kept on the fp32 tensor
    w16 = None
    if MATH_MODE and CORR16 and Cin % 32 == 0 and k <= 15:
        cache = w.__dict__.setdefault("_g6d_c16", {})
        if MATH_MODE not in cache:
            cache[MATH_MODE] = corr_filters16(w, k, {1: torch.bfloat16, 2: torch.float16}[MATH_MODE])
        w16 = cache[MATH_MODE]
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if w16 is not None:
        _lib.check(_lib.load().g6d_corr2d_patch16_multi(segs, len(xs), Cin, _ptr(w16), Cout, k, k, _ptr(ws), ws.numel() * 4, int(MATH_MODE),
                                                       _stream()), "g6d_corr2d_patch16_multi")
    else:
        _lib.check(_lib.load().g6d_corr2d_patch_multi(segs, len(xs), Cin, _ptr(w), Cout, k, k, _ptr(ws), ws.numel() * 4, int(MATH_MODE),
                                                     _stream()), "g6d_corr2d_patch_multi")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((flops, e0, e1, f"corr2d_patch{'16' if w16 is not None else ''} multi in={'+'.join(sizes)}x{Cin} out={Cout} k={k}x{k}",
                        4.0 * (sum(x.numel() for x in xs) + sum(o.numel() for o in outs) + w.numel())))
    return outs


def corr2d_wino_multi(xs, U, outs, kblocks=5):
    """The 15x15 correlation level on the Winograd kernel (g6d_corr2d_wino_multi): xs[i] [N,1,H_i,W_i,Cin] and outs[i]
    [N,1,H_i,W_i,Cout] dense tensors cut from one buffer each, U [kblocks^2 * Cin/8, 16, Cout, 8] (backbone.winograd_corr_filters)."""
    _need_gpu(U, *xs, *outs)
    if not 1 <= len(xs) <= 4 or len(outs) != len(xs):
        raise ValueError("corr2d_wino_multi: 1..4 map sizes")
    Cin, Cout = xs[0].shape[4], U.shape[2]
    if tuple(U.shape) != (kblocks * kblocks * (Cin // 8), 16, Cout, 8) or not U.is_contiguous():
        raise ValueError(f"corr2d_wino_multi: U must be contiguous {(kblocks * kblocks * (Cin // 8), 16, Cout, 8)}")
    segs = (_lib.G6dCorrSeg * len(xs))()
    flops, sizes = 0.0, []
    k = 3 * kblocks
    for i, (x, o) in enumerate(zip(xs, outs)):
        N, D, H, W, Cx, ld_in = _cl5(x, "corr2d_wino.x")
        No, _, Ho, Wo, Co, ld_out = _cl5(o, "corr2d_wino.out")
        if D != 1 or No != N or (Ho, Wo) != (H, W) or Cx != Cin or Co != Cout or not (x.is_contiguous() and o.is_contiguous()):
            raise ValueError("corr2d_wino_multi: shape mismatch (maps must be dense)")
        segs[i] = _lib.G6dCorrSeg(in_=x.data_ptr(), out=o.data_ptr(), H=H, W=W, ld_in=ld_in, ld_out=ld_out, N=N)
        flops += 2.0 * N * H * W * Cout * k * k * Cin
        sizes.append(f"{N}x{H}x{W}" if N > 1 else f"{H}x{W}")
    ws = workspace(U.device)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().g6d_corr2d_wino_multi(segs, len(xs), Cin, _ptr(U), Cout, int(kblocks), _ptr(ws), ws.numel() * 4, _stream()),
               "g6d_corr2d_wino_multi")
    if PROFILE is not None:
        e1.record()
        # FLOPs executed in the Winograd domain = direct form / 2.25 (booked in the Winograd family)
        PROFILE.append((flops / 2.25, e0, e1, f"wino3x3 corr multi in={'+'.join(sizes)}x{Cin} out={Cout} k={k}x{k} ({kblocks}x{kblocks} blocks of 3x3)",
                        4.0 * (sum(x.numel() for x in xs) + sum(o.numel() for o in outs) + U.numel())))
    return outs


_ARENA = {}          # (device, stream) -> [buffer, bump offset]
_CUR_ARENA = None    # arena of the query being enqueued (host-side state; set by stats_arena_begin)
USE_ARENA = True     # False (tools / tests): one zero-fill per InstanceNorm statistics buffer instead of one arena per query
ARENA_DOUBLES = 1 << 21      # 16 MB: a batch of 8 queries needs ~140 K accumulators in the refiner step (56 images x 1024 channels x 2), 32 queries 560 K


def stats_arena_begin(device):
    """Start a new query: zero ONE per-(device, stream) arena of fp64 (sum, sumsq) accumulators; new_stats() then hands
    out slices of it instead of launching one zero-fill kernel per InstanceNorm layer (62 per query).  Keyed by the
    launching stream so that queries captured / enqueued on different streams never share accumulators."""
    global _CUR_ARENA
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in _ARENA:
        _ARENA[key] = [torch.zeros(ARENA_DOUBLES, dtype=torch.float64, device=device), 0, 0]
    a = _ARENA[key]
    # Eager launches clear only what has ever been handed out of this arena (a[2] = its all-time high-water mark: everything behind it
    # is still zero from the allocation) — 2-4 MB instead of 16.  A CAPTURED clear keeps the size it had at capture time, and eager work
    # of another shape on the same stream could later dirty the arena beyond it: under capture the whole arena is cleared (four 16 MB
    # fills per batch of 16 queries = 20 us of a 63 ms step buy a replay that is right whatever ran on the stream in between).
    n_clear = ARENA_DOUBLES if (a[0].is_cuda and torch.cuda.is_current_stream_capturing()) else a[2]
    if n_clear:
        _lib.check(_lib.load().g6d_zero_bytes(_ptr(a[0]), n_clear * 8, _stream()), "g6d_zero_bytes")
    a[1] = 0
    _CUR_ARENA = a


def new_stats(groups, channels, device):
    a = _CUR_ARENA if USE_ARENA else None
    n = groups * channels * 2
    if a is None or a[0].device != torch.device(device) or a[1] + n > ARENA_DOUBLES:
        return torch.zeros((groups, channels, 2), dtype=torch.float64, device=device)
    t = a[0][a[1]:a[1] + n].view(groups, channels, 2)
    a[1] += n
    a[2] = max(a[2], a[1])
    return t


def new_counter(device):
    """One zeroed int32 (an arrival counter of a launch that finalises its statistics): a slot of the query's arena."""
    a = _CUR_ARENA if USE_ARENA else None
    if a is None or a[0].device != torch.device(device) or a[1] + 1 > ARENA_DOUBLES:
        return torch.zeros((2,), dtype=torch.int32, device=device)
    t = a[0][a[1]:a[1] + 1].view(torch.int32)
    a[1] += 1
    a[2] = max(a[2], a[1])
    return t


def stats_finalize(stats, count, eps=1e-5):
    """[G,C,2] fp64 (sum, sumsq) -> scale, shift float32 [G,C]."""
    _need_gpu(stats)
    G, Cc, _ = stats.shape
    scale = torch.empty((G, Cc), dtype=torch.float32, device=stats.device)
    shift = torch.empty_like(scale)
    _lib.check(_lib.load().g6d_stats_finalize(_ptr(stats), G * Cc, float(count), float(eps), _ptr(scale), _ptr(shift),
                                             _stream()), "g6d_stats_finalize")
    return scale, shift


def affine_act_pool(x, out, scale=None, shift=None, per_n=False, relu=False, pool=0):
    """x [N,1,H,W,C] view -> out: pool 0 same shape; 1 -> [N,1,H/2,W/2,C]; 2 -> [N,1,1,1,C] (mean over HxW)."""
    _need_gpu(x, out)
    N, D, H, W, Cc, ld_in = _cl5(x, "affine_act_pool.x")
    _, _, _, _, Co, ld_out = _cl5(out, "affine_act_pool.out")
    if Co != Cc:
        raise ValueError("affine_act_pool: channel mismatch")
    _lib.check(_lib.load().g6d_affine_act_pool(_ptr(x), ld_in, _ptr(scale), _ptr(shift), int(per_n) * D, int(relu), int(pool),
                                              N * D, H, W, Cc, _ptr(out), ld_out, _stream()), "g6d_affine_act_pool")
    return out


def upsample_bilinear(x, out, factor, scale=None, shift=None, per_n=False):
    _need_gpu(x, out)
    N, D, H, W, Cc, ld_in = _cl5(x, "upsample.x")
    _, _, Ho, Wo, Co, ld_out = _cl5(out, "upsample.out")
    if (Ho, Wo, Co) != (H * factor, W * factor, Cc):
        raise ValueError("upsample: output shape mismatch")
    _lib.check(_lib.load().g6d_upsample_bilinear(_ptr(x), ld_in, _ptr(scale), _ptr(shift), int(per_n) * D, N * D, H, W, Cc,
                                                int(factor), _ptr(out), ld_out, _stream()), "g6d_upsample_bilinear")
    return out


def bias_relu_pool_nchw(x, bias, relu, pool):
    """x [N,C,H,W] contiguous (conv output without bias) -> maxpool2x2?(relu?(x + bias)) in one pass."""
    _need_gpu(x, bias)
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise ValueError("bias_relu_pool_nchw: x must be contiguous float32 NCHW")
    N, Cc, H, W = x.shape
    out = torch.empty((N, Cc, H // 2, W // 2) if pool else (N, Cc, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().g6d_bias_relu_pool_nchw(_ptr(x), _ptr(bias), N, Cc, H, W, int(relu), int(pool), _ptr(out), _stream()),
               "g6d_bias_relu_pool_nchw")
    return out


def vgg_conv1_pool(x, w_oihw, bias):
    """First trunk layer fused: x [N,3,H,W] contiguous, w [64,3,3,3] (BatchNorm folded), bias [64] ->
    maxpool2x2(relu(conv3x3(x) + bias)) [N,64,H//2,W//2]."""
    _need_gpu(x, w_oihw, bias)
    if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous() or not w_oihw.is_contiguous():
        raise ValueError("vgg_conv1_pool: x and w must be contiguous float32 NCHW / OIHW")
    N, Cin, H, W = x.shape
    Cout = w_oihw.shape[0]
    if tuple(w_oihw.shape) != (Cout, Cin, 3, 3) or bias.numel() != Cout:
        raise ValueError("vgg_conv1_pool: weight / bias shape mismatch")
    out = torch.empty((N, Cout, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().g6d_vgg_conv1_pool(_ptr(x), N, H, W, _ptr(w_oihw), _ptr(bias), Cin, Cout, _ptr(out), _stream()),
               "g6d_vgg_conv1_pool")
    return out


def vgg_conv1_pool_nhwc(x, w_oihw, bias, out=None, norm=None):
    """As vgg_conv1_pool with a channels-last result [N,H//2,W//2,64] (the input layout of wino_conv3x3); `out`: a contiguous
    destination of that shape (a slice of the buffer the scales of a pyramid share); `norm` = (mean, std) 3-tuples: x is an
    image in [0,1] and (x - mean) / std is applied inside the kernel."""
    _need_gpu(x, w_oihw, bias)
    if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous() or not w_oihw.is_contiguous():
        raise ValueError("vgg_conv1_pool_nhwc: x and w must be contiguous float32 NCHW / OIHW")
    N, Cin, H, W = x.shape
    Cout = w_oihw.shape[0]
    if tuple(w_oihw.shape) != (Cout, Cin, 3, 3) or bias.numel() != Cout:
        raise ValueError("vgg_conv1_pool_nhwc: weight / bias shape mismatch")
    if out is None:
        out = torch.empty((N, H // 2, W // 2, Cout), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (N, H // 2, W // 2, Cout) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("vgg_conv1_pool_nhwc: out must be contiguous float32 [N,H/2,W/2,Cout]")
    if norm is not None:
        m, sd = (C.c_float * 3)(*norm[0]), (C.c_float * 3)(*norm[1])
        _lib.check(_lib.load().g6d_vgg_conv1_pool_nhwc_norm(_ptr(x), N, H, W, _ptr(w_oihw), _ptr(bias), Cin, Cout, m, sd, _ptr(out),
                                                            _stream()), "g6d_vgg_conv1_pool_nhwc_norm")
        return out
    _lib.check(_lib.load().g6d_vgg_conv1_pool_nhwc(_ptr(x), N, H, W, _ptr(w_oihw), _ptr(bias), Cin, Cout, _ptr(out), _stream()),
               "g6d_vgg_conv1_pool_nhwc")
    return out


def wino_conv3x3(x, U, bias, relu=True, full=True, pool=False):
    """Trunk layer on the Winograd/MFMA kernel.  x [N,H,W,Cin] channels-last (last stride 1, dense rows of ld = x.stride(2)),
    U [Cin/8,16,Cout,8] (backbone.winograd_filters), bias [Cout] -> (y [N,H,W,Cout] or None, maxpool2x2(y) or None)."""
    _need_gpu(x, U, bias)
    if x.dim() != 4 or x.dtype != torch.float32 or x.stride(3) != 1:
        raise ValueError("wino_conv3x3: x must be a float32 channels-last [N,H,W,C] view")
    N, H, W, Cin = x.shape
    ld_in = x.stride(2)
    if (N > 1 and x.stride(0) != H * W * ld_in) or (H > 1 and x.stride(1) != W * ld_in):
        raise ValueError("wino_conv3x3: x rows must be dense")
    Cout = U.shape[2]
    if tuple(U.shape) != (Cin // 8, 16, Cout, 8) or not U.is_contiguous() or bias.numel() != Cout:
        raise ValueError(f"wino_conv3x3: U must be contiguous {(Cin // 8, 16, Cout, 8)}")
    y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device) if full else None
    yp = torch.empty((N, H // 2, W // 2, Cout), dtype=torch.float32, device=x.device) if pool else None
    flops = 2.0 * N * H * W * Cout * 9 * Cin
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = workspace(x.device)
    _lib.check(_lib.load().g6d_wino_conv3x3(_ptr(x), N, H, W, Cin, ld_in, _ptr(U), _ptr(bias), Cout, int(relu), _ptr(y), Cout,
                                           _ptr(yp), Cout, _ptr(ws), ws.numel() * 4, _stream()), "g6d_wino_conv3x3")
    if PROFILE is not None:
        e1.record()
        # direct-form FLOPs / 2.25 = multiplications actually executed in the Winograd domain (what the matrix cores do)
        PROFILE.append((flops / 2.25, e0, e1, f"wino3x3 N={N} in={H}x{W}x{Cin} out={Cout}{' full' if full else ''}{' pool' if pool else ''}",
                        4.0 * (x.numel() + U.numel() + (y.numel() if full else 0) + (yp.numel() if pool else 0))))
    return y, yp


def alloc_like_segments(shapes, device):
    """One flat float32 buffer cut into contiguous tensors of `shapes` (16-byte aligned starts): the segments of a
    wino_conv3x3_multi launch are addressed with 32-bit offsets from a common base."""
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    starts, tot = [], 0
    for n in sizes:
        starts.append(tot)
        tot += (n + 3) // 4 * 4
    buf = torch.empty((max(tot, 1),), dtype=torch.float32, device=device)
    return [buf[st:st + n].view(sh) for st, n, sh in zip(starts, sizes, shapes)]


def wino_conv3x3_multi(xs, U, bias, relu=True, full=True, pool=False):
    """One trunk layer over several map sizes in ONE launch (the scales of the detector's image pyramid): xs = dense
    channels-last [N_i,H_i,W_i,Cin] tensors cut from one buffer (alloc_like_segments / the outputs of the previous layer).
    Returns (list of y_i or None, list of maxpool2x2(y_i) or None)."""
    _need_gpu(U, bias, *xs)
    if not 1 <= len(xs) <= 4:
        raise ValueError("wino_conv3x3_multi: 1..4 segments")
    Cin, Cout = xs[0].shape[3], U.shape[2]
    if tuple(U.shape) != (Cin // 8, 16, Cout, 8) or not U.is_contiguous() or bias.numel() != Cout:
        raise ValueError(f"wino_conv3x3_multi: U must be contiguous {(Cin // 8, 16, Cout, 8)}")
    for x in xs:
        if x.dim() != 4 or x.dtype != torch.float32 or not x.is_contiguous() or x.shape[3] != Cin:
            raise ValueError("wino_conv3x3_multi: segments must be contiguous float32 [N,H,W,Cin]")
    dev = xs[0].device
    ys = alloc_like_segments([(x.shape[0], x.shape[1], x.shape[2], Cout) for x in xs], dev) if full else None
    yps = alloc_like_segments([(x.shape[0], x.shape[1] // 2, x.shape[2] // 2, Cout) for x in xs], dev) if pool else None
    segs = (_lib.G6dWinoSeg * len(xs))()
    flops = 0.0
    for i, x in enumerate(xs):
        N, H, W, _ = x.shape
        segs[i] = _lib.G6dWinoSeg(in_=x.data_ptr(), out_full=ys[i].data_ptr() if full else None,
                                  out_pool=yps[i].data_ptr() if pool else None, N=N, H=H, W=W, ld_in=Cin, ld_full=Cout, ld_pool=Cout)
        flops += 2.0 * N * H * W * Cout * 9 * Cin
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = workspace(dev)
    _lib.check(_lib.load().g6d_wino_conv3x3_multi(segs, len(xs), Cin, _ptr(U), _ptr(bias), Cout, int(relu), _ptr(ws), ws.numel() * 4,
                                                 _stream()), "g6d_wino_conv3x3_multi")
    if PROFILE is not None:
        e1.record()
        sizes = "+".join(f"{x.shape[0]}x{x.shape[1]}x{x.shape[2]}" for x in xs)
        PROFILE.append((flops / 2.25, e0, e1, f"wino3x3 multi in={sizes}x{Cin} out={Cout}{' full' if full else ''}{' pool' if pool else ''}",
                        4.0 * (sum(x.numel() for x in xs) + U.numel() + (sum(t.numel() for t in ys) if full else 0) + (sum(t.numel() for t in yps) if pool else 0))))
    return ys, yps


def w43_shape(chunks, Cout):
    """Shape of the F(4x4,3x3) filter tensor (backbone.winograd43_filters) for `chunks` 8-channel chunks and Cout output channels."""
    cb = 32 if Cout % 64 else 64
    return (chunks, 2, Cout // cb, 18, cb // 32, 4, 16, 4)


def wino43_conv3x3_multi(xs, U43, bias, relu=True, full=True, pool=False):
    """wino_conv3x3_multi on the Winograd F(4x4,3x3) kernel (g6d_wino43_conv3x3_multi): 4x fewer multiplications than the direct form
    (1.78x fewer than F(2x2,3x3)) at ~5x the fp32 error — for the layers whose parity budget has the room (the detector's pyramid).
    U43 = backbone.winograd43_filters(w) (shape w43_shape(Cin/8, Cout)), Cout % 64 == 0."""
    _need_gpu(U43, bias, *xs)
    if not 1 <= len(xs) <= 4:
        raise ValueError("wino43_conv3x3_multi: 1..4 segments")
    Cin, Cout = xs[0].shape[3], bias.numel()
    if tuple(U43.shape) != w43_shape(Cin // 8, Cout) or not U43.is_contiguous():
        raise ValueError(f"wino43_conv3x3_multi: U43 must be contiguous {w43_shape(Cin // 8, Cout)}")
    for x in xs:
        if x.dim() != 4 or x.dtype != torch.float32 or not x.is_contiguous() or x.shape[3] != Cin:
            raise ValueError("wino43_conv3x3_multi: segments must be contiguous float32 [N,H,W,Cin]")
    dev = xs[0].device
    ys = alloc_like_segments([(x.shape[0], x.shape[1], x.shape[2], Cout) for x in xs], dev) if full else None
    yps = alloc_like_segments([(x.shape[0], x.shape[1] // 2, x.shape[2] // 2, Cout) for x in xs], dev) if pool else None
    segs = (_lib.G6dWinoSeg * len(xs))()
    flops = 0.0
    for i, x in enumerate(xs):
        N, H, W, _ = x.shape
        segs[i] = _lib.G6dWinoSeg(in_=x.data_ptr(), out_full=ys[i].data_ptr() if full else None,
                                  out_pool=yps[i].data_ptr() if pool else None, N=N, H=H, W=W, ld_in=Cin, ld_full=Cout, ld_pool=Cout)
        flops += 2.0 * N * H * W * Cout * 9 * Cin
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = workspace(dev)
    _lib.check(_lib.load().g6d_wino43_conv3x3_multi(segs, len(xs), Cin, _ptr(U43), _ptr(bias), Cout, int(relu), _ptr(ws), ws.numel() * 4,
                                                   _stream()), "g6d_wino43_conv3x3_multi")
    if PROFILE is not None:
        e1.record()
        sizes = "+".join(f"{x.shape[0]}x{x.shape[1]}x{x.shape[2]}" for x in xs)
        # direct-form FLOPs / 4 = multiplications executed in the F(4x4,3x3) domain (36 per 16 outputs instead of 144)
        PROFILE.append((flops / 4, e0, e1, f"wino3x3 F43 multi in={sizes}x{Cin} out={Cout}{' full' if full else ''}{' pool' if pool else ''}",
                        4.0 * (sum(x.numel() for x in xs) + U43.numel() + (sum(t.numel() for t in ys) if full else 0) + (sum(t.numel() for t in yps) if pool else 0)),
                        flops))
    return ys, yps


def corr2d_wino43_multi(xs, U43, outs, kblocks=5, k_true=None):
    """corr2d_wino_multi on the F(4x4,3x3) kernel (g6d_corr2d_wino43_multi): U43 = backbone.winograd43_corr_filters(w, 15)
    (shape w43_shape(kblocks^2 * Cin/8, Cout)), Cout % 32 == 0.  kblocks = 3 with k_true = 7: the 7x7 level on zero-extended 9x9
    filters (backbone.winograd43_corr_filters_padded; k_true only enters the profile's direct-form FLOP count)."""
    _need_gpu(U43, *xs, *outs)
    if not 1 <= len(xs) <= 4 or len(outs) != len(xs):
        raise ValueError("corr2d_wino43_multi: 1..4 map sizes")
    Cin, Cout = xs[0].shape[4], outs[0].shape[4]
    if tuple(U43.shape) != w43_shape(kblocks * kblocks * (Cin // 8), Cout) or not U43.is_contiguous():
        raise ValueError(f"corr2d_wino43_multi: U43 must be contiguous {w43_shape(kblocks * kblocks * (Cin // 8), Cout)}")
    segs = (_lib.G6dCorrSeg * len(xs))()
    flops, sizes = 0.0, []
    k = 3 * kblocks
    for i, (x, o) in enumerate(zip(xs, outs)):
        N, D, H, W, Cx, ld_in = _cl5(x, "corr2d_wino43.x")
        No, _, Ho, Wo, Co, ld_out = _cl5(o, "corr2d_wino43.out")
        if D != 1 or No != N or (Ho, Wo) != (H, W) or Cx != Cin or Co != Cout or not (x.is_contiguous() and o.is_contiguous()):
            raise ValueError("corr2d_wino43_multi: shape mismatch (maps must be dense)")
        segs[i] = _lib.G6dCorrSeg(in_=x.data_ptr(), out=o.data_ptr(), H=H, W=W, ld_in=ld_in, ld_out=ld_out, N=N)
        flops += 2.0 * N * H * W * Cout * k * k * Cin
        sizes.append(f"{N}x{H}x{W}" if N > 1 else f"{H}x{W}")
    kt = k_true or k
    ws = workspace(U43.device)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().g6d_corr2d_wino43_multi(segs, len(xs), Cin, _ptr(U43), Cout, int(kblocks), _ptr(ws), ws.numel() * 4, _stream()),
               "g6d_corr2d_wino43_multi")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((flops / 4, e0, e1, f"wino3x3 F43 corr multi in={'+'.join(sizes)}x{Cin} out={Cout} k={kt}x{kt} ({kblocks}x{kblocks} blocks of 3x3)",
                        4.0 * (sum(x.numel() for x in xs) + sum(o.numel() for o in outs) + U43.numel()), flops * kt * kt / (k * k)))
    return outs


def wino16_conv3x3_multi(xs, U16, bias, relu=True, full=True, pool=False):
    """wino_conv3x3_multi with 16-bit matrix-core operands (MATH_MODE 1 = bf16, 2 = fp16; fp32 activations in and out): U16
    [Cin/16,16,Cout,16] in torch.bfloat16 / torch.float16 (backbone.winograd_filters16).  xs as wino_conv3x3_multi."""
    _need_gpu(U16, bias, *xs)
    if not 1 <= len(xs) <= 4:
        raise ValueError("wino16_conv3x3_multi: 1..4 segments")
    mm = {torch.bfloat16: 1, torch.float16: 2}.get(U16.dtype)
    Cin, Cout = xs[0].shape[3], U16.shape[2]
    if mm is None or tuple(U16.shape) != (Cin // 16, 16, Cout, 16) or not U16.is_contiguous() or bias.numel() != Cout:
        raise ValueError(f"wino16_conv3x3_multi: U16 must be contiguous bfloat16 / float16 {(Cin // 16, 16, Cout, 16)}")
    for x in xs:
        if x.dim() != 4 or x.dtype != torch.float32 or not x.is_contiguous() or x.shape[3] != Cin:
            raise ValueError("wino16_conv3x3_multi: segments must be contiguous float32 [N,H,W,Cin]")
    dev = xs[0].device
    ys = alloc_like_segments([(x.shape[0], x.shape[1], x.shape[2], Cout) for x in xs], dev) if full else None
    yps = alloc_like_segments([(x.shape[0], x.shape[1] // 2, x.shape[2] // 2, Cout) for x in xs], dev) if pool else None
    segs = (_lib.G6dWinoSeg * len(xs))()
    flops = 0.0
    for i, x in enumerate(xs):
        N, H, W, _ = x.shape
        segs[i] = _lib.G6dWinoSeg(in_=x.data_ptr(), out_full=ys[i].data_ptr() if full else None,
                                  out_pool=yps[i].data_ptr() if pool else None, N=N, H=H, W=W, ld_in=Cin, ld_full=Cout, ld_pool=Cout)
        flops += 2.0 * N * H * W * Cout * 9 * Cin
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = workspace(dev)
    _lib.check(_lib.load().g6d_wino16_conv3x3_multi(segs, len(xs), Cin, _ptr(U16), _ptr(bias), Cout, int(relu), mm, _ptr(ws), ws.numel() * 4,
                                                   _stream()), "g6d_wino16_conv3x3_multi")
    if PROFILE is not None:
        e1.record()
        sizes = "+".join(f"{x.shape[0]}x{x.shape[1]}x{x.shape[2]}" for x in xs)
        PROFILE.append((flops / 2.25, e0, e1, f"wino3x3 {'bf16' if mm == 1 else 'fp16'} multi in={sizes}x{Cin} out={Cout}{' full' if full else ''}{' pool' if pool else ''}",
                        4.0 * (sum(x.numel() for x in xs) + U16.numel() / 2 + (sum(t.numel() for t in ys) if full else 0) + (sum(t.numel() for t in yps) if pool else 0))))
    return ys, yps


_T16 = {1: torch.bfloat16, 2: torch.float16, 3: torch.float16}


class Conv16Filters:
    """Filters of g6d_conv16_direct_multi: `data` (16-bit, flat), `layout` (0 = [Cout][taps][Cin] rows, 1 = fragment-major), `mode`
    (1 bf16, 2 fp16, 3 fp16 hi / lo pairs), `acc_scale` (1 / the power-of-two scale the filters carry), and the layer's shape."""

    def __init__(self, data, layout, mode, acc_scale, Cout, taps, Cin):
        self.data, self.layout, self.mode, self.acc_scale, self.Cout, self.taps, self.Cin = data, layout, mode, acc_scale, Cout, taps, Cin


def conv16_pack(w_taps, mode, layout=1):
    """[Cout, taps, Cin] fp32 filters (tap = (kz*3 + ky)*3 + kx) -> Conv16Filters for `mode` (1 bf16, 2 fp16, 3 fp16 hi / lo pairs).
    layout 1 (fragment-major, include/gen6d_hip.h): [Cout/128][Cin/BK][taps][BK/16][planes][4][64 lanes][8], BK = 64 (pairs: 32), lane l
    of group j holds filter co = 128 tile + 32 j + (l & 31), ci = BK slice + 16 ks + 8 (l >> 5) + e.  Mode 3: the filters are scaled by an
    exact power of two S (their lo parts stay normal fp16 numbers; the kernel multiplies the accumulators by 1 / S) and split in fp64."""
    co, taps, ci = w_taps.shape
    bk = 32 if mode == 3 else 64
    co_true = co
    if co == 64 and layout == 1 and taps == 9:
        # the selector's first product layer: one 128-channel tile whose upper half is zero (the kernel launches waves for 64 channels only)
        w_taps = torch.cat([w_taps, torch.zeros_like(w_taps)], 0)
        co = 128
    if co % 128 or ci % bk:
        raise ValueError("conv16_pack: Cout % 128 == 0 (pairs, 2-D: also 64) and Cin % 64 (pairs: 32) == 0 expected")
    acc_scale = 1.0
    if mode == 3:
        import math
        amax = float(w_taps.abs().max())
        S = 2.0 ** min(14, math.floor(math.log2(2048.0 / max(amax, 1e-30))))
        w = w_taps.double() * S
        hi = w.to(torch.float16)
        lo = (w - hi.double()).to(torch.float16)
        planes = torch.stack([hi, lo], 0)
        acc_scale = 1.0 / S
    else:
        planes = w_taps.to(_T16[mode])[None]
    if layout == 0:
        if mode == 3:
            raise ValueError("conv16_pack: pairs need the fragment-major layout")
        return Conv16Filters(planes[0].contiguous(), 0, mode, 1.0, co, taps, ci)
    P, ks = planes.shape[0], bk // 16
    x = planes.reshape(P, co // 128, 4, 32, taps, ci // bk, ks, 2, 8)           # P, tile, j, l31, tap, slice, ks, half, e
    x = x.permute(1, 5, 4, 6, 0, 2, 7, 3, 8).contiguous()                        # tile, slice, tap, ks, P, j, half, l31, e
    return Conv16Filters(x.reshape(-1), 1, mode, acc_scale, co_true, taps, ci)


def product_split16(ref, que, scale, shift, mode):
    """The selector's query x reference product in the activation format of conv16_direct_multi (g6d_product_split16): ref [D,P,C], que
    [qn,P,C], scale / shift [qn,C] fp32 -> [qn*D, P, C] (mode 1 / 2) or [qn*D, P, 2, C] fp16 pairs (mode 3)."""
    _need_gpu(ref, que, scale, shift)
    D, P, Cc = ref.shape
    qn = que.shape[0]
    if tuple(que.shape) != (qn, P, Cc) or tuple(scale.shape) != (qn, Cc) or tuple(shift.shape) != (qn, Cc) or not all(
            t.is_contiguous() and t.dtype == torch.float32 for t in (ref, que, scale, shift)):
        raise ValueError("product_split16: dense fp32 ref [D,P,C], que [qn,P,C], scale / shift [qn,C] expected")
    out = torch.empty((qn * D, P, 2, Cc) if mode == 3 else (qn * D, P, Cc), dtype=_T16[mode], device=ref.device)
    nbytes = 4.0 * (ref.numel() + que.numel()) + 2.0 * out.numel()
    _timed_hbm("product_split16", nbytes, lambda: _lib.check(_lib.load().g6d_product_split16(
        _ptr(ref), _ptr(que), _ptr(scale), _ptr(shift), _ptr(out), qn, D, P, Cc, int(mode), _stream()), "g6d_product_split16"))
    return out


def affine_split16(x, scale, shift, per_n, relu, pool, mode):
    """affine_act_pool (pool False / True = 2x2 max) with the result in the activation format of conv16_direct_multi (g6d_affine_split16):
    x [N,1,H,W,C] fp32 view -> [N,Ho,Wo,C] (mode 1 / 2) or [N,Ho,Wo,2,C] fp16 pairs (mode 3)."""
    _need_gpu(x)
    N, D, H, W, Cc, ld_in = _cl5(x, "affine_split16.x")
    if D != 1:
        raise ValueError("affine_split16: 2-D maps expected")
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.empty((N, Ho, Wo, 2, Cc) if mode == 3 else (N, Ho, Wo, Cc), dtype=_T16[mode], device=x.device)
    _timed_hbm("affine_split16", 4.0 * N * H * W * Cc + 2.0 * out.numel(), lambda: _lib.check(_lib.load().g6d_affine_split16(
        _ptr(x), ld_in, _ptr(scale), _ptr(shift), int(per_n), int(bool(relu)), int(bool(pool)), N, H, W, Cc, _ptr(out), int(mode), _stream()),
        "g6d_affine_split16"))
    return out


def vgg_conv1_pool_nhwc16(x, w_oihw, bias, out=None, norm=None, mode=None):
    """vgg_conv1_pool_nhwc with a 16-bit channels-last result (g6d_vgg_conv1_pool_nhwc16): mode 1 / 2 (default: the current math mode) =
    bf16 / fp16 [N,H/2,W/2,64], the first layer of the reduced-precision mode's 16-bit activation path; mode 3 = fp16 hi / lo pairs
    [N,H/2,W/2,2,64], the first layer of the fp32 path's split-precision trunk."""
    _need_gpu(x, w_oihw, bias)
    mode = MATH_MODE if mode is None else mode
    if mode not in _T16:
        raise RuntimeError("vgg_conv1_pool_nhwc16: mode 1 (bf16), 2 (fp16) or 3 (fp16 pairs)")
    N, Cin, H, W = x.shape
    Cout = w_oihw.shape[0]
    shape = (N, H // 2, W // 2, 2, Cout) if mode == 3 else (N, H // 2, W // 2, Cout)
    if out is None:
        out = torch.empty(shape, dtype=_T16[mode], device=x.device)
    if out.dtype != _T16[mode] or not out.is_contiguous() or tuple(out.shape) != shape:
        raise ValueError(f"vgg_conv1_pool_nhwc16: out must be a dense {_T16[mode]} tensor of shape {shape}")
    mean = std = None
    if norm is not None:
        mean, std = (C.c_float * 3)(*norm[0]), (C.c_float * 3)(*norm[1])
    _lib.check(_lib.load().g6d_vgg_conv1_pool_nhwc16(_ptr(x.contiguous()), N, H, W, _ptr(w_oihw.contiguous()), _ptr(bias), Cin, Cout,
                                                     mean, std, _ptr(out), int(mode), _stream()), "g6d_vgg_conv1_pool_nhwc16")
    return out


def conv16_direct_multi(xs, filt, bias, relu=True, full=None, pool=None, kd=1, stats=None, rows_per_group=0, out_full=None):
    """Direct 3x3 / 3x3x3 convolution on 16-bit activations (g6d_conv16_direct_multi).  filt: Conv16Filters (conv16_pack); its mode
    decides the arithmetic: 1 / 2 = bf16 / fp16 operands (the reduced-precision mode), 3 = fp16 hi / lo pairs (fp32-class results: the
    fp32 path's trunk).  xs: 1..4 dense channels-last tensors of the mode's 16-bit type, [N,H,W,Cin] (kd = 1) or [N,D,H,W,Cin] (kd = 3);
    pairs carry an extra plane axis in front of the channels: [N,H,W,2,Cin].
    full / pool: None = not produced, torch.float32, or "t16" = the mode's 16-bit format (pairs for mode 3) -> lists of dense outputs
    (None where not produced).  stats [G,Cout,2] fp64 (zeroed): sum / sum of squares of the fp32 results are added.  out_full: optional list
    of caller-provided dense output tensors."""
    _need_gpu(filt.data, *xs)
    mode = filt.mode
    t16 = _T16[mode]
    Cout, taps, Cin = filt.Cout, filt.taps, filt.Cin
    if taps != 9 * kd:
        raise ValueError("conv16_direct_multi: filters / kd mismatch")
    pair = mode == 3
    code = {None: 0, "t16": 3 if pair else 1, torch.float32: 2}
    if full not in code or pool not in code:
        raise ValueError('conv16_direct_multi: output types are None, torch.float32 or "t16"')
    segs = (_lib.G6dConv16Seg * len(xs))()
    fulls, pools, flops, sizes, nbytes = [], [], 0.0, [], 2.0 * filt.data.numel()
    nd = (4 if kd == 1 else 5) + (1 if pair else 0)
    for i, x in enumerate(xs):
        if x.dtype != t16 or not x.is_contiguous() or x.dim() != nd or x.shape[-1] != Cin or (pair and x.shape[-2] != 2):
            raise ValueError(f"conv16_direct_multi: input {i} must be a dense {t16} channels-last tensor with {Cin} channels{' in hi / lo planes' if pair else ''}")
        N, D, H, W = (x.shape[0], 1, x.shape[1], x.shape[2]) if kd == 1 else tuple(x.shape[:4])
        lead = (N, H, W) if kd == 1 else (N, D, H, W)

        def alloc(kind, lead_):
            if kind is None:
                return None
            if kind == "t16":
                return torch.empty(lead_ + ((2, Cout) if pair else (Cout,)), dtype=t16, device=x.device)
            return torch.empty(lead_ + (Cout,), dtype=torch.float32, device=x.device)
        q = alloc(pool, (N, H // 2, W // 2))
        ld_f = None
        if out_full is not None:                          # caller-provided outputs (slices of one buffer; fp32: also a channel slice of wider rows)
            f = out_full[i]
            want_t = torch.float32 if full is torch.float32 else t16
            nel = N * D * H * W * Cout * (2 if (pair and full == "t16") else 1)
            if full is None or f.dtype != want_t or f.numel() != nel or f.stride(-1) != 1:
                raise ValueError("conv16_direct_multi: out_full must hold the outputs' type and size")
            if not f.is_contiguous():
                ld_f = f.stride(-2)                       # [.., W, Cout] view of rows of ld_f elements
                if full is not torch.float32 or f.shape[-1] != Cout or any(f.stride(d) != f.stride(d + 1) * f.shape[d + 1] for d in range(f.dim() - 2)):
                    raise ValueError("conv16_direct_multi: a strided out_full must be an fp32 channel slice of dense rows")
        else:
            f = alloc(full, lead)
        fulls.append(f); pools.append(q)
        ld = lambda t_: (2 * Cout if t_.dtype != torch.float32 and pair else Cout)
        segs[i] = _lib.G6dConv16Seg(in_=x.data_ptr(), out_full=f.data_ptr() if f is not None else None,
                                    out_pool=q.data_ptr() if q is not None else None, N=N, D=D, H=H, W=W, ld_in=(2 if pair else 1) * Cin,
                                    ld_full=(ld_f or ld(f)) if f is not None else 0, ld_pool=ld(q) if q is not None else 0)
        flops += 2.0 * N * D * H * W * Cout * taps * Cin
        nbytes += x.numel() * 2.0 + sum(t.numel() * t.element_size() for t in (f, q) if t is not None)
        sizes.append("x".join(str(v) for v in lead))
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().g6d_conv16_direct_multi(segs, len(xs), Cin, _ptr(filt.data), int(filt.layout), float(filt.acc_scale), _ptr(bias), Cout, int(kd),
                                                  int(bool(relu)), code[full], code[pool], int(mode), _ptr(stats), int(rows_per_group), _stream()),
               "g6d_conv16_direct_multi")
    if PROFILE is not None:
        e1.record()
        # mode 3 executes three 16-bit MFMAs per product: booked as "conv16x3" with the DIRECT-FORM flops (what the fp32 kernels it replaces are booked with / 4)
        PROFILE.append((flops, e0, e1, f"{'conv16x3' if pair else 'conv16'} direct in={'+'.join(sizes)}x{Cin} out={Cout} k={'3x' if kd == 3 else ''}3x3"
                        f"{' full' if full is not None else ''}{' pool' if pool is not None else ''}{' stats' if stats is not None else ''}", nbytes, flops))
    return fulls, pools


def corr16_pack(w_taps, mode):
    """[32, k*k, Cin] fp32 reference-centre features (tap = ky*k + kx) -> the filters of g6d_corr16_multi for `mode` (1 bf16, 2 fp16, 3 fp16
    hi / lo pairs): [Cin/S][k*k][2][64 lanes][8] with S = 32 and fragment f = the slice's 16-channel group (modes 1 / 2), S = 16 and f = hi / lo
    plane (mode 3); lane l holds reference l & 31, channels S c + 16 f (modes 1 / 2) + 8 (l >> 5) + e.  Mode 3 scales by an exact power of two."""
    co, taps, ci = w_taps.shape
    if co != 32 or ci % 32:
        raise ValueError("corr16_pack: 32 references and Cin % 32 == 0 expected")
    acc_scale = 1.0
    if mode == 3:
        import math
        amax = float(w_taps.abs().max())
        S = 2.0 ** min(14, math.floor(math.log2(2048.0 / max(amax, 1e-30))))
        w = w_taps.double() * S
        hi = w.to(torch.float16)
        lo = (w - hi.double()).to(torch.float16)
        acc_scale = 1.0 / S
        x = torch.stack([hi, lo], 0).reshape(2, 32, taps, ci // 16, 2, 8)           # plane, r, tap, slice, half, e
        x = x.permute(3, 2, 0, 4, 1, 5).contiguous()                                 # slice, tap, plane, half, r, e
    else:
        x = w_taps.to(_T16[mode]).reshape(32, taps, ci // 32, 2, 2, 8)               # r, tap, slice, group, half, e
        x = x.permute(2, 1, 3, 4, 0, 5).contiguous()                                 # slice, tap, group, half, r, e
    k = int(round(taps ** 0.5))
    f = Conv16Filters(x.reshape(-1), 1, mode, acc_scale, co, taps, ci)
    f.k = k
    return f


def corr16_multi(xs, filt, outs):
    """The detector's k x k correlation on 16-bit activations (g6d_corr16_multi).  xs: 1..4 dense channels-last maps [N,H,W,Cin] of the
    mode's 16-bit type (pairs: [N,H,W,2,Cin]); outs: fp32 [N,1,H,W,32] (or [N,H,W,32]) dense; filt: corr16_pack(...)."""
    _need_gpu(filt.data, *xs, *outs)
    mode, Cin = filt.mode, filt.Cin
    pair = mode == 3
    segs = (_lib.G6dConv16Seg * len(xs))()
    flops, sizes, nbytes = 0.0, [], 2.0 * filt.data.numel()
    for i, (x, o) in enumerate(zip(xs, outs)):
        if x.dtype != _T16[mode] or not x.is_contiguous() or x.dim() != (5 if pair else 4) or x.shape[-1] != Cin or not o.is_contiguous() \
                or o.dtype != torch.float32 or o.shape[-1] != 32:
            raise ValueError("corr16_multi: dense 16-bit channels-last inputs and dense fp32 outputs with 32 channels expected")
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
        if o.numel() != N * H * W * 32:
            raise ValueError("corr16_multi: output shape mismatch")
        segs[i] = _lib.G6dConv16Seg(in_=x.data_ptr(), out_full=o.data_ptr(), out_pool=None, N=N, D=1, H=H, W=W, ld_in=(2 if pair else 1) * Cin,
                                    ld_full=32, ld_pool=0)
        flops += 2.0 * N * H * W * 32 * filt.taps * Cin
        nbytes += x.numel() * 2.0 + o.numel() * 4.0
        sizes.append(f"{N}x{H}x{W}")
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().g6d_corr16_multi(segs, len(xs), Cin, _ptr(filt.data), float(filt.acc_scale), 32, int(filt.k), int(mode), _stream()),
               "g6d_corr16_multi")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((flops, e0, e1, f"{'conv16x3' if pair else 'conv16'} corr in={'+'.join(sizes)}x{Cin} out=32 k={filt.k}x{filt.k}", nbytes, flops))
    return outs


def l2norm_rows(x):
    """In-place F.normalize over the last axis of a channels-last tensor whose rows are dense (ld = C), or of a 2-D row-strided
    view [rows, C] (ld = x.stride(0))."""
    _need_gpu(x)
    Cc = x.shape[-1]
    if x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and not x.is_contiguous():
        _lib.check(_lib.load().g6d_l2norm_rows(_ptr(x), x.shape[0], Cc, x.stride(0), _stream()), "g6d_l2norm_rows")
        return x
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("l2norm_rows: contiguous float32 expected")
    _lib.check(_lib.load().g6d_l2norm_rows(_ptr(x), x.numel() // Cc, Cc, Cc, _stream()), "g6d_l2norm_rows")
    return x


def nchw_to_nhwc(x, out, l2norm):
    """x [N,C,H,W] contiguous -> out [N,1,H,W,C] view, optionally L2-normalised over C."""
    _need_gpu(x, out)
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise ValueError("nchw_to_nhwc: x must be contiguous float32 NCHW")
    N, Cc, H, W = x.shape
    _, _, Ho, Wo, Co, ld_out = _cl5(out, "nchw_to_nhwc.out")
    if (Ho, Wo, Co) != (H, W, Cc):
        raise ValueError("nchw_to_nhwc: shape mismatch")
    _lib.check(_lib.load().g6d_nchw_to_nhwc(_ptr(x), N, Cc, H, W, int(l2norm), _ptr(out), ld_out, _stream()), "g6d_nchw_to_nhwc")
    return out


def selector_ref_sums(refs):
    """refs [D,HW,C] contiguous -> r1, r2 fp64 [HW,C]."""
    _need_gpu(refs)
    D, HW, Cc = refs.shape
    r1 = torch.empty((HW, Cc), dtype=torch.float64, device=refs.device)
    r2 = torch.empty_like(r1)
    _lib.check(_lib.load().g6d_selector_ref_sums(_ptr(refs.contiguous()), D, HW, Cc, _ptr(r1), _ptr(r2), _stream()),
               "g6d_selector_ref_sums")
    return r1, r2


def selector_prod_affine(que, r1, r2, D, eps=1e-5):
    """que [HW,C] -> InstanceNorm affine (scale, shift) [1,C] of the product que*refs over (D,HW)."""
    _need_gpu(que, r1, r2)
    HW, Cc = que.shape
    scale = torch.empty((1, Cc), dtype=torch.float32, device=que.device)
    shift = torch.empty_like(scale)
    _lib.check(_lib.load().g6d_selector_prod_affine(_ptr(que), _ptr(r1), _ptr(r2), D, HW, Cc, float(eps), _ptr(scale),
                                                   _ptr(shift), _stream()), "g6d_selector_prod_affine")
    return scale, shift


def selector_scan(que, refs):
    """que [HW,C], refs [D,HW,C] -> score_map [D,HW], vps [D]."""
    _need_gpu(que, refs)
    D, HW, Cc = refs.shape
    if not (que.is_contiguous() and refs.is_contiguous()):
        raise ValueError("selector_scan: operands must be contiguous")
    smap = torch.empty((D, HW), dtype=torch.float32, device=que.device)
    vps = torch.empty((D,), dtype=torch.float32, device=que.device)
    # algorithmic bytes: the reference cache and the query features read once, the score map written and re-read for vps
    _timed_hbm("selector_scan", 4.0 * (D * HW * Cc + HW * Cc + 2 * D * HW + D),
               lambda: _lib.check(_lib.load().g6d_selector_scan(_ptr(que), _ptr(refs), D, HW, Cc, _ptr(smap), _ptr(vps),
                                                                _stream()), "g6d_selector_scan"))
    return smap, vps


def selector_levels(ques, refs, sums, Dg, eps=1e-5, want_maps=False, _out=None):
    """All pyramid levels of a batch of queries in one launch: ques[l] [qn,HW_l,C] (or [HW_l,C] for one query), refs[l] [D,HW_l,C],
    sums[l] = (r1, r2) of selector_ref_sums -> vps [qn,L,D], scale [qn,L,C], shift [qn,L,C] (the InstanceNorm affine of the product
    over Dg*HW_l values; a single 2-D query gives [L,D] / [L,C]) and, if want_maps, the score maps [qn,D,HW_l]."""
    L = len(ques)
    _need_gpu(*ques, *refs)
    single = ques[0].dim() == 2
    if single:
        ques = [q.unsqueeze(0) for q in ques]
    qn = ques[0].shape[0]
    D, _, Cc = refs[0].shape
    dev = ques[0].device
    if qn > 8:                                           # the kernel keeps <= 8 query rows in registers: groups of 8 (one cache pass each),
        vps = torch.empty((qn, L, D), dtype=torch.float32, device=dev)                 # every group writes its rows of the results
        scale = torch.empty((qn, L, Cc), dtype=torch.float32, device=dev)
        shift = torch.empty_like(scale)
        maps = [torch.empty((qn, D, r.shape[1]), dtype=torch.float32, device=dev) for r in refs] if want_maps else None
        for i in range(0, qn, 8):
            selector_levels([q[i:i + 8] for q in ques], refs, sums, Dg, eps, want_maps,
                            _out=(vps[i:i + 8], scale[i:i + 8], shift[i:i + 8], [m[i:i + 8] for m in maps] if want_maps else None))
        return vps, scale, shift, maps
    for q, r in zip(ques, refs):
        if not (q.is_contiguous() and r.is_contiguous()) or r.shape[0] != D or r.shape[2] != Cc or tuple(q.shape) != (qn,) + tuple(r.shape[1:]):
            raise ValueError("selector_levels: operands must be contiguous [qn,HW,C] / [D,HW,C]")
    if _out is not None:
        vps, scale, shift, maps = _out
    else:
        vps = torch.empty((qn, L, D), dtype=torch.float32, device=dev)
        scale = torch.empty((qn, L, Cc), dtype=torch.float32, device=dev)
        shift = torch.empty_like(scale)
        maps = None
    if maps is None:
        maps = [torch.empty((qn, D, r.shape[1]), dtype=torch.float32, device=dev) for r in refs]
    arr = lambda ts: (C.c_void_p * L)(*[t.data_ptr() if t is not None else None for t in ts])
    hw = (C.c_int * L)(*[r.shape[1] for r in refs])
    # algorithmic bytes: the reference cache once per BATCH, query rows, score maps written and re-read, r1/r2 (fp64) per query
    nbytes = sum(4.0 * (D * r.shape[1] * Cc + qn * (r.shape[1] * Cc + 2 * D * r.shape[1] + D)) + 16.0 * qn * r.shape[1] * Cc for r in refs)
    _timed_hbm("selector_levels", nbytes,
               lambda: _lib.check(_lib.load().g6d_selector_levels(L, qn, arr(ques), arr(refs), arr([s_[0] for s_ in sums]), arr([s_[1] for s_ in sums]),
                                                                  hw, D, int(Dg), Cc, float(eps), arr(maps), _ptr(vps),
                                                                  _ptr(scale), _ptr(shift), _stream()), "g6d_selector_levels"))
    if single:
        return vps[0], scale[0], shift[0], ([m[0] for m in maps] if want_maps else None)
    return vps, scale, shift, (maps if want_maps else None)


def refiner_volume(feats, projs, rot_in, lin, h_in, w_in, mean_in, std):
    """feats [rfn+1,fh,fw,C], projs [rfn+1,3,4], rot_in [3,3], lin [sn] -> mean_in [sn^3,2C], std [sn^3,C] (written)."""
    _need_gpu(feats, projs, rot_in, lin, mean_in, std)
    V, fh, fw, Cc = feats.shape
    sn = lin.numel()
    for t in (feats, projs, rot_in, lin, mean_in, std):
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError("refiner_volume: operands must be contiguous float32")
    if tuple(mean_in.shape) != (sn ** 3, 2 * Cc) or tuple(std.shape) != (sn ** 3, Cc) or tuple(projs.shape) != (V, 3, 4):
        raise ValueError("refiner_volume: shape mismatch")
    # algorithmic bytes: feature maps read once (L2 resident afterwards), the three volumes written
    _timed_hbm("refiner_volume", 4.0 * (V * fh * fw * Cc + 3 * sn ** 3 * Cc),
               lambda: _lib.check(_lib.load().g6d_refiner_volume(_ptr(feats), _ptr(projs), _ptr(rot_in), _ptr(lin), V - 1, fh,
                                                                 fw, Cc, int(h_in), int(w_in), sn, _ptr(mean_in), _ptr(std),
                                                                 _stream()), "g6d_refiner_volume"))
    return mean_in, std


def refiner_volume_kp(feats, ref_Ks, ref_poses, K_in, pose_in, lin, h_in, w_in, mean_in, std):
    """refiner_volume with the projections formed inside the kernel: feats [rfn+1,fh,fw,C] (query last), ref_Ks [rfn,3,3],
    ref_poses [rfn,3,4], K_in [3,3], pose_in [3,4] (also the volume's rotation) -> mean_in [sn^3,2C], std [sn^3,C]; or a batch of
    queries in one launch: every operand and both outputs with a leading [B] axis."""
    _need_gpu(feats, ref_Ks, ref_poses, K_in, pose_in, lin, mean_in, std)
    batched = feats.dim() == 5
    B = feats.shape[0] if batched else 1
    V, fh, fw, Cc = feats.shape[-4:]
    sn = lin.numel()
    for t in (feats, ref_Ks, ref_poses, K_in, pose_in, lin, mean_in, std):
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError("refiner_volume_kp: operands must be contiguous float32")
    lead = (B,) if batched else ()
    if (tuple(mean_in.shape) != lead + (sn ** 3, 2 * Cc) or tuple(std.shape) != lead + (sn ** 3, Cc) or tuple(ref_Ks.shape) != lead + (V - 1, 3, 3) or
            tuple(ref_poses.shape) != lead + (V - 1, 3, 4) or tuple(K_in.shape) != lead + (3, 3) or tuple(pose_in.shape) != lead + (3, 4)):
        raise ValueError("refiner_volume_kp: shape mismatch")
    _timed_hbm("refiner_volume", 4.0 * B * (V * fh * fw * Cc + 3 * sn ** 3 * Cc),
               lambda: _lib.check(_lib.load().g6d_refiner_volume_kp(_ptr(feats), _ptr(ref_Ks), _ptr(ref_poses), _ptr(K_in), _ptr(pose_in),
                                                                    _ptr(lin), V - 1, fh, fw, Cc, int(h_in), int(w_in), sn, _ptr(mean_in),
                                                                    _ptr(std), B, _stream()), "g6d_refiner_volume_kp"))
    return mean_in, std


def cat1(ts, dim=0):
    """torch.cat that hands a single tensor through (no copy launch for calls that fit one chunk)."""
    return ts[0] if len(ts) == 1 else torch.cat(ts, dim)


def resize_bilinear_pyramid(imgs, sizes):
    """F.interpolate(imgs, size=s, mode='bilinear') (align_corners False) for every size of `sizes` in ONE launch
    (g6d_resize_bilinear_pyramid; reference network/detector.py:236-241): imgs [N,3,H,W] contiguous -> list of [N,3,h,w], cut from one
    buffer; a size equal to the image's own returns the image itself."""
    _need_gpu(imgs)
    if imgs.dim() != 4 or imgs.dtype != torch.float32 or not imgs.is_contiguous():
        raise ValueError("resize_bilinear_pyramid: imgs must be contiguous float32 [N,C,H,W]")
    N, Cc, H, W = imgs.shape
    todo = [i for i, (h, w) in enumerate(sizes) if (h, w) != (H, W)]
    if len(todo) > 4:
        raise ValueError("resize_bilinear_pyramid: at most 4 destination sizes")
    outs = [imgs] * len(sizes)
    if not todo:
        return outs
    buf = torch.empty(sum(N * Cc * sizes[i][0] * sizes[i][1] for i in todo), dtype=torch.float32, device=imgs.device)
    o = 0
    dst = []
    for i in todo:
        h, w = sizes[i]
        dst.append(buf[o:o + N * Cc * h * w].view(N, Cc, h, w)); o += N * Cc * h * w
        outs[i] = dst[-1]
    hs = (C.c_int * len(todo))(*[sizes[i][0] for i in todo])
    ws_ = (C.c_int * len(todo))(*[sizes[i][1] for i in todo])
    ptrs = (C.c_void_p * len(todo))(*[d.data_ptr() for d in dst])
    _lib.check(_lib.load().g6d_resize_bilinear_pyramid(_ptr(imgs), N * Cc, H, W, len(todo), hs, ws_, ptrs, _stream()), "g6d_resize_bilinear_pyramid")
    return outs


def detector_assemble(s0, s1, s2, hc, wc, mu_sigma, clip, hs, ws, scale_idx, stacked, batch=1):
    """s_l [batch*h_l*w_l, rfn] raw correlation maps of one scale (the maps of the queries one after the other) -> channels
    3*scale_idx.. of stacked [batch*hs*ws, rfn, nch]."""
    _need_gpu(s0, s1, s2, stacked)
    rfn = s0.shape[1]
    P, rfn2, nch = stacked.shape
    if P != batch * hs * ws or rfn2 != rfn or not stacked.is_contiguous() or not (s0.is_contiguous() and s1.is_contiguous() and s2.is_contiguous()):
        raise ValueError("detector_assemble: stacked shape mismatch")
    if s0.shape[0] != batch * hc * wc or s1.shape[0] != batch * (hc // 2) * (wc // 2) or s2.shape[0] != batch * (hc // 4) * (wc // 4):
        raise ValueError("detector_assemble: level map sizes do not match")
    ms = (C.c_float * 6)(*[float(v) for pair in mu_sigma for v in pair])
    _lib.check(_lib.load().g6d_detector_assemble(_ptr(s0), _ptr(s1), _ptr(s2), hc, wc, rfn, ms, float(clip), hs, ws,
                                                int(scale_idx), nch, _ptr(stacked), int(batch), _stream()), "g6d_detector_assemble")
    return stacked


def detector_score_mlp_max(stacked, w0, b0, w1, b1):
    _need_gpu(stacked, w0, b0, w1, b1)
    P, rfn, nch = stacked.shape
    out = torch.empty((P, 64), dtype=torch.float32, device=stacked.device)
    _lib.check(_lib.load().g6d_detector_score_mlp_max(_ptr(stacked), P, rfn, nch, _ptr(w0), _ptr(b0), _ptr(w1), _ptr(b1),
                                                     _ptr(out), _stream()), "g6d_detector_score_mlp_max")
    return out


def detector_decode(scores, offset, scale, hs, ws, pool_ratio, batch=1):
    """scores [P,1], offset [P,2], scale [P,1] (row-strided views allowed), P = batch*hs*ws -> result [5] ([batch,5] for batch > 1)."""
    _need_gpu(scores, offset, scale)
    if scores.shape[0] != batch * hs * ws:
        raise ValueError("detector_decode: row count mismatch")
    res = torch.empty((batch, 5), dtype=torch.float32, device=scores.device)
    _lib.check(_lib.load().g6d_detector_decode(_ptr(scores), scores.stride(0), _ptr(offset), offset.stride(0), _ptr(scale),
                                              scale.stride(0), hs, ws, int(pool_ratio), _ptr(res), int(batch), _stream()),
               "g6d_detector_decode")
    return res[0] if batch == 1 else res


def vps_norm(vps, feats, c_off):
    """vps [3,D] (or [qn,3,D]) -> InstanceNorm over D -> feats[:, c_off:c_off+3] (feats [qn*D,ld] contiguous)."""
    _need_gpu(vps, feats)
    batch = vps.shape[0] if vps.dim() == 3 else 1
    D = vps.shape[-1]
    if feats.shape[0] != batch * D:
        raise ValueError("vps_norm: feats rows != batch * D")
    _lib.check(_lib.load().g6d_vps_norm(_ptr(vps.contiguous()), D, _ptr(feats), feats.stride(0), int(c_off), batch, _stream()),
               "g6d_vps_norm")
    return feats


def max_an_add(x, rfn, an, embed, out, batch=1):
    """x [batch*rfn*an, C] (row stride allowed) -> out[b*rfn+r] = max_a x[(b*rfn+r)*an+a] + embed[r]."""
    _need_gpu(x, embed, out)
    Cc = x.shape[1]
    if x.shape[0] != batch * rfn * an or out.shape[0] != batch * rfn:
        raise ValueError("max_an_add: row count mismatch")
    _lib.check(_lib.load().g6d_max_an_add(_ptr(x), x.stride(0), rfn, an, Cc, _ptr(embed.contiguous()), _ptr(out),
                                         out.stride(0), int(batch), _stream()), "g6d_max_an_add")
    return out


def attention(q, k, v, heads, out, batch=1):
    """q, k, v [batch*n, C] -> out [batch*n, C]: attention among the n tokens of each query of the batch."""
    _need_gpu(q, k, v, out)
    rows, Cc = q.shape
    if not (q.stride(0) == k.stride(0) == v.stride(0)) or rows % batch:
        raise ValueError("attention: q/k/v must share the row stride; rows = batch * n")
    _lib.check(_lib.load().g6d_attention(_ptr(q), _ptr(k), _ptr(v), q.stride(0), rows // batch, Cc, heads, _ptr(out), out.stride(0),
                                        int(batch), _stream()), "g6d_attention")
    return out


def layernorm(x, gamma, beta, out, eps=1e-5):
    _need_gpu(x, gamma, beta, out)
    n, Cc = x.shape
    _lib.check(_lib.load().g6d_layernorm(_ptr(x), x.stride(0), n, Cc, _ptr(gamma), _ptr(beta), float(eps), _ptr(out),
                                        out.stride(0), _stream()), "g6d_layernorm")
    return out


def affine_act_add(x, out, scale=None, shift=None, relu=False, residual=None, rows_per_group=0):
    """out = relu?(x*scale+shift) (+ residual) on [n,C] rows; rows_per_group = k: row r uses table r // k of scale / shift [n/k,C]."""
    _need_gpu(x, out)
    n, Cc = x.shape
    _lib.check(_lib.load().g6d_affine_act_add(_ptr(x), x.stride(0), _ptr(scale), _ptr(shift), int(relu), _ptr(residual),
                                             residual.stride(0) if residual is not None else 0, n, Cc, _ptr(out),
                                             out.stride(0), int(rows_per_group), _stream()), "g6d_affine_act_add")
    return out


def linear_gemv(x, W, bias, act=0):
    """x [B,K] contiguous, W [O,K] contiguous -> [B,O] (g6d_linear_gemv_batch: any B, groups of 8 per weight pass)."""
    _need_gpu(x, W)
    B, K = x.shape
    O = W.shape[0]
    out = torch.empty((B, O), dtype=torch.float32, device=x.device)
    x = x.contiguous()
    ws = workspace(x.device)
    _timed_hbm("linear_gemv" if O * K >= (1 << 22) else "linear_gemv_small", 4.0 * (O * K + B * K + B * O),
               lambda: _lib.check(_lib.load().g6d_linear_gemv_batch(_ptr(x), B, K, _ptr(W), _ptr(bias), O, int(act), _ptr(out), _ptr(ws),
                                                                    ws.numel() * 4, _stream()), "g6d_linear_gemv_batch"))
    return out


def warp_perspective(src_u8, H, dh, dw, out_float=False):
    """src_u8 uint8 [sh,sw,ch] on the GPU; H = 3x3 source->destination pixel homography (numpy, as cv2.warpPerspective
    takes it; a 2x3 affine as cv2.warpAffine takes it is accepted too) -> [dh,dw,ch] uint8 or float32 in [0,1]."""
    import numpy as np
    _need_gpu(src_u8)
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 3 or not src_u8.is_contiguous():
        raise ValueError("warp_perspective: src must be a contiguous uint8 [H,W,C] tensor")
    H = np.asarray(H, dtype=np.float64)
    if H.shape == (2, 3):
        H = np.concatenate([H, [[0.0, 0.0, 1.0]]], 0)
    hinv = np.linalg.inv(H).astype(np.float32).reshape(-1)
    sh, sw, ch = src_u8.shape
    dst = torch.empty((dh, dw, ch), dtype=torch.float32 if out_float else torch.uint8, device=src_u8.device)
    _lib.check(_lib.load().g6d_warp_perspective(_ptr(src_u8), sh, sw, ch, (C.c_float * 9)(*hinv.tolist()), _ptr(dst), dh, dw,
                                               int(out_float), 1.0 / 255.0, _stream()), "g6d_warp_perspective")
    return dst


# ------------------------------------------------------------------------------------------------ device-resident chain
def _f32c(*ts):
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("chain ops: contiguous float32 device tensors expected")


def chain_crop_from_detection(det, size):
    """det [5] (g6d_detector_decode result) or [B,5] -> hinv [B,9] of the selector crops."""
    _need_gpu(det); _f32c(det)
    B = det.shape[0] if det.dim() == 2 else 1
    hinv = torch.empty((B, 9), dtype=torch.float32, device=det.device)
    _lib.check(_lib.load().g6d_chain_crop_from_detection(_ptr(det), float(size), _ptr(hinv), B, _stream()), "g6d_chain_crop_from_detection")
    return hinv


def chain_pose_from_selection(det, logits, angles, ref_poses, ref_Ks, que_K, center):
    """-> pose [3,4], sel [2] = (reference index, in-plane angle), all on the device; batch: det [B,5], logits / angles [B,rfn],
    que_K [B,9] -> pose [B,3,4], sel [B,2]."""
    _need_gpu(det, logits, angles, ref_poses, ref_Ks, que_K, center); _f32c(det, logits, angles, ref_poses, ref_Ks, que_K, center)
    batched = det.dim() == 2
    B = det.shape[0] if batched else 1
    rfn = logits.shape[-1]
    if logits.numel() != B * rfn or angles.numel() != B * rfn or que_K.numel() != 9 * B:
        raise ValueError("chain_pose_from_selection: per-query operands must share the batch size")
    pose = torch.empty((B, 3, 4), dtype=torch.float32, device=det.device)
    sel = torch.empty((B, 2), dtype=torch.float32, device=det.device)
    _lib.check(_lib.load().g6d_chain_pose_from_selection(_ptr(det), _ptr(logits), _ptr(angles), rfn, _ptr(ref_poses),
                                                        _ptr(ref_Ks), _ptr(que_K), _ptr(center), _ptr(pose), _ptr(sel), B, _stream()),
               "g6d_chain_pose_from_selection")
    return (pose, sel) if batched else (pose[0], sel[0])


def chain_refine_prepare(pose_in, que_K, norm, size, margin, sub_poses, sub_Ks, ref_num, angle_step=0.0):
    """-> geo [42 + 30*ref_num] float32 (see include/gen6d_hip.h), ref_idx [ref_num] int32; with angle_step > 0 (radians) the
    alignment angles are snapped to its multiples and the third result is their buckets [ref_num] int32 (cache keys).
    Batch: pose_in [B,12], que_K [B,9] -> geo [B,42+30*ref_num], ref_idx [B,ref_num] (, buckets [B,ref_num])."""
    _need_gpu(pose_in, que_K, norm, sub_poses, sub_Ks); _f32c(pose_in, que_K, norm, sub_poses, sub_Ks)
    batched = pose_in.dim() == 2
    B = pose_in.shape[0] if batched else 1
    if pose_in.numel() != 12 * B or que_K.numel() != 9 * B:
        raise ValueError("chain_refine_prepare: pose_in [B,12] and que_K [B,9] expected")
    geo = torch.empty((B, 42 + 30 * ref_num), dtype=torch.float32, device=pose_in.device)
    idx = torch.empty((B, ref_num), dtype=torch.int32, device=pose_in.device)
    bucket = torch.empty((B, ref_num), dtype=torch.int32, device=pose_in.device) if angle_step > 0 else None
    _lib.check(_lib.load().g6d_chain_refine_prepare(_ptr(pose_in), _ptr(que_K), _ptr(norm), float(size), float(margin), _ptr(sub_poses),
                                                   _ptr(sub_Ks), sub_poses.shape[0], int(ref_num), _ptr(geo), _ptr(idx),
                                                   float(angle_step), _ptr(bucket), B, _stream()),
               "g6d_chain_refine_prepare")
    if not batched:
        geo, idx, bucket = geo[0], idx[0], (bucket[0] if bucket is not None else None)
    return (geo, idx, bucket) if angle_step > 0 else (geo, idx)


def chain_refine_update(rot, off, scl, geo, norm):
    """rot [4] / off [2] / scl [1] + geo record -> pose [3,4]; batch: rot [B,4], off [B,2], scl [B,1], geo [B,G] -> pose [B,3,4]."""
    _need_gpu(rot, off, scl, geo, norm); _f32c(rot, off, scl, geo, norm)
    batched = geo.dim() == 2
    B = geo.shape[0] if batched else 1
    if rot.numel() != 4 * B or off.numel() != 2 * B or scl.numel() != B:
        raise ValueError("chain_refine_update: refiner outputs must share the batch size of geo")
    pose = torch.empty((B, 3, 4), dtype=torch.float32, device=geo.device)
    _lib.check(_lib.load().g6d_chain_refine_update(_ptr(rot), _ptr(off), _ptr(scl), _ptr(geo), geo.shape[-1], _ptr(norm), _ptr(pose), B,
                                                  _stream()), "g6d_chain_refine_update")
    return pose if batched else pose[0]


def warp_batch(stack, single, idx, hinv, dh, dw, out=None):
    """stack uint8 [n,sh,sw,ch] or None, single uint8 [sh,sw,ch] or None, idx int32 [B] or None, hinv float32 [B,9]
    -> float32 [B,ch,dh,dw] in [0,1] (uint8-rounded); `out`: optional contiguous destination (e.g. rows of a larger batch)."""
    src = stack if stack is not None else single
    _need_gpu(src, hinv)
    if src.dtype != torch.uint8 or not src.is_contiguous() or (single is not None and (single.dtype != torch.uint8 or not single.is_contiguous())):
        raise ValueError("warp_batch: contiguous uint8 sources expected")
    if stack is not None and single is not None and tuple(stack.shape[1:]) != tuple(single.shape):
        raise ValueError("warp_batch: stack and single image sizes differ")
    sh, sw, ch = src.shape[-3:]
    B = hinv.shape[0]
    _f32c(hinv)
    if idx is not None and (idx.dtype != torch.int32 or idx.numel() != B):
        raise ValueError("warp_batch: idx must be int32 [B]")
    dst = out if out is not None else torch.empty((B, ch, dh, dw), dtype=torch.float32, device=src.device)
    if tuple(dst.shape) != (B, ch, dh, dw) or dst.dtype != torch.float32 or not dst.is_contiguous():
        raise ValueError("warp_batch: out must be a contiguous float32 [B,ch,dh,dw] tensor")
    _lib.check(_lib.load().g6d_warp_batch(_ptr(stack), _ptr(single), _ptr(idx), B, sh, sw, ch, _ptr(hinv), _ptr(dst), dh, dw, _stream()),
               "g6d_warp_batch")
    return dst
