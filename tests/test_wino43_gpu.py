"""The Winograd F(4x4,3x3) kernel (csrc/wino43_conv.hip, ABI v8) on a real MI355X: every entry point through gen6d_amd.ops -> ctypes
against the fp64 convolution on the CPU.  Bars: 1e-5 of the output range (the kernel's fp32 transform constants put it at ~1.5e-6;
F(2x2,3x3) sits at ~3e-7), statistics 2e-5."""
import pytest
import torch
import torch.nn.functional as F

import ref_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from gen6d_amd import lib, ops as _ops
    lib.load()
    assert torch.cuda.is_available()
    return _ops


def _rand(g, *shape, scale=1.0):
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _d(t):
    return t.detach().cpu().double() if t is not None else None


def _err(got, want):
    got, want = got.detach().cpu().double(), want.double()
    return (got - want).abs().max().item() / max(want.abs().max().item(), 1e-30)


def _where(got, want):
    """Diagnostic: coordinates of the worst element (N, H, W, C) and the error pattern per (h % 8, w % 8)."""
    d = (got.detach().cpu().double() - want.double()).abs()
    idx = torch.nonzero(d == d.max())[0].tolist()
    pat = d.amax(dim=(0, 3))
    H, W = pat.shape
    rows = [max(pat[i::8].max().item() if i < H else 0.0 for _ in [0]) for i in range(8)]
    cols = [pat[:, j::8].max().item() if j < W else 0.0 for j in range(8)]
    chan = d.amax(dim=(0, 1, 2))
    return f"worst at {idx}; by row%8 {['%.1e' % r for r in rows]}; by col%8 {['%.1e' % c for c in cols]}; by channel%16 {['%.1e' % chan[i::16].max().item() for i in range(16)]}"


W43_MULTI_CASES = [
    # segment sizes (N, H, W), Cin, Cout, relu, full, pool
    ([(1, 8, 8)], 8, 64, False, True, False),                                                # one quarter, one chunk
    ([(1, 16, 16)], 16, 64, False, True, True),                                              # 4 quarters, two chunks
    ([(2, 9, 7), (1, 8, 8), (3, 5, 13)], 128, 64, True, True, False),                        # ragged sizes, several images per segment
    ([(1, 22, 30), (1, 16, 20)], 64, 128, True, False, True),                                # pooled output only, blocks straddle segments
    ([(1, 44, 58), (1, 30, 40), (1, 22, 30), (1, 16, 20)], 512, 512, False, True, True),     # detector pyramid, 1/16 level, c7_pre + p7
    ([(1, 88, 116), (1, 60, 80), (1, 44, 60), (1, 32, 40)], 256, 512, True, True, False),    # 1/8 level: un-split
    ([(2, 64, 64)], 64, 128, True, True, True),                                              # crops: batch of images
]


@pytest.mark.parametrize("sizes,Cin,Cout,relu,full,pool", W43_MULTI_CASES)
def test_wino43_conv3x3_multi(ops, sizes, Cin, Cout, relu, full, pool):
    """One launch over several map sizes against F.conv2d per segment in float64; twice, so the split counters are left re-armed."""
    from gen6d_amd.network.backbone import winograd43_filters
    g = torch.Generator().manual_seed(4343 + Cin + len(sizes))
    w = _rand(g, Cout, Cin, 3, 3, scale=(2.0 / (9 * Cin)) ** 0.5 * 1.7)
    b = _rand(g, Cout, scale=0.3)
    xs_cpu = [_rand(g, n, h, ww, Cin) for n, h, ww in sizes]
    xs = ops.alloc_like_segments([tuple(x.shape) for x in xs_cpu], torch.device("cuda"))
    for d, x in zip(xs, xs_cpu):
        d.copy_(x)
    U = winograd43_filters(w).cuda()
    for rep in range(2):
        ys, yps = ops.wino43_conv3x3_multi(xs, U, b.cuda(), relu=relu, full=full, pool=pool)
        torch.cuda.synchronize()
        assert (ys is not None) == full and (yps is not None) == pool
        for i, x in enumerate(xs_cpu):
            ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1)
            if relu:
                ref = F.relu(ref)
            if full:
                e = _err(ys[i].permute(0, 3, 1, 2), ref)
                assert e <= 1e-5, f"full seg {i} rep {rep}: {e:.3e}; {_where(ys[i], ref.permute(0, 2, 3, 1))}"
            if pool:
                pr = F.max_pool2d(ref, 2, 2)
                e = _err(yps[i].permute(0, 3, 1, 2), pr)
                assert e <= 1e-5, f"pool seg {i} rep {rep}: {e:.3e}; {_where(yps[i], pr.permute(0, 2, 3, 1))}"


def test_wino43_split_grid(ops):
    """A grid of one block column: the chunk list is split over gridDim.z and the partial output tiles are added in-kernel."""
    from gen6d_amd.network.backbone import winograd43_filters
    g = torch.Generator().manual_seed(7)
    Cin, Cout = 512, 64
    w = _rand(g, Cout, Cin, 3, 3, scale=(2.0 / (9 * Cin)) ** 0.5)
    b = _rand(g, Cout, scale=0.3)
    x = _rand(g, 1, 24, 16, Cin)
    U = winograd43_filters(w).cuda()
    ref = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1))
    for rep in range(3):
        ys, _ = ops.wino43_conv3x3_multi([x.cuda()], U, b.cuda(), relu=True, full=True, pool=False)
        e = _err(ys[0].permute(0, 3, 1, 2), ref)
        assert e <= 1e-5, f"rep {rep}: {e:.3e}; {_where(ys[0], ref.permute(0, 2, 3, 1))}"


@pytest.mark.parametrize("N,sizes,Cin,Cout", [(1, [(16, 16)], 8, 32), (3, [(22, 30), (9, 13)], 64, 32), (2, [(16, 16)], 128, 64),
                                              (1, [(88, 116), (60, 80), (44, 60), (32, 40)], 512, 32)])
def test_corr2d_wino43_multi(ops, N, sizes, Cin, Cout):
    """15x15 correlation as 5x5 blocks of 3x3 sub-filters accumulated in the F(4x4,3x3) domain against the fp64 direct correlation."""
    from gen6d_amd.network.backbone import winograd43_corr_filters
    g = torch.Generator().manual_seed(900 + Cin)
    k = 15
    w = _rand(g, Cout, k * k, Cin, scale=(1.0 / (k * k * Cin)) ** 0.5)
    U = winograd43_corr_filters(w, k).cuda()
    xs_cpu = [_rand(g, N, 1, h, ww, Cin) for h, ww in sizes]
    dev = torch.device("cuda")
    xs = ops.alloc_like_segments([tuple(x.shape) for x in xs_cpu], dev)
    for d_, x in zip(xs, xs_cpu):
        d_.copy_(x)
    outs = ops.alloc_like_segments([(N, 1, h, ww, Cout) for h, ww in sizes], dev)
    for rep in range(2):
        for o in outs:
            o.fill_(-3.0)
        ops.corr2d_wino43_multi(xs, U, outs, 5)
        for o, xc in zip(outs, xs_cpu):
            ref = torch.empty(tuple(o.shape), dtype=torch.float64)
            ref_ops.corr2d_patch(_d(xc), _d(w), ref, k)
            e = _err(o, ref)
            assert e <= 2e-5, f"rep {rep}: {e:.3e}; {_where(o[:, 0], ref[:, 0])}"


@pytest.mark.parametrize("N,sizes,Cin,Cout", [(1, [(16, 16)], 8, 32), (3, [(22, 30), (9, 13)], 64, 32),
                                              (2, [(44, 58), (30, 40), (22, 30), (16, 20)], 512, 32)])
def test_corr2d_wino43_multi_7x7_on_9x9_blocks(ops, N, sizes, Cin, Cout):
    """The 7x7 "same" correlation as 3x3 blocks of 3x3 on zero-extended 9x9 filters (kblocks = 3) against the fp64 direct 7x7 correlation."""
    from gen6d_amd.network.backbone import winograd43_corr_filters_padded
    g = torch.Generator().manual_seed(700 + Cin)
    k = 7
    w = _rand(g, Cout, k * k, Cin, scale=(1.0 / (k * k * Cin)) ** 0.5)
    U, kb = winograd43_corr_filters_padded(w, k)
    assert kb == 3
    U = U.cuda()
    xs_cpu = [_rand(g, N, 1, h, ww, Cin) for h, ww in sizes]
    dev = torch.device("cuda")
    xs = ops.alloc_like_segments([tuple(x.shape) for x in xs_cpu], dev)
    for d_, x in zip(xs, xs_cpu):
        d_.copy_(x)
    outs = ops.alloc_like_segments([(N, 1, h, ww, Cout) for h, ww in sizes], dev)
    for rep in range(2):
        for o in outs:
            o.fill_(-3.0)
        ops.corr2d_wino43_multi(xs, U, outs, 3, k_true=7)
        for o, xc in zip(outs, xs_cpu):
            ref = torch.empty(tuple(o.shape), dtype=torch.float64)
            ref_ops.corr2d_patch(_d(xc), _d(w), ref, k)
            e = _err(o, ref)
            assert e <= 2e-5, f"rep {rep}: {e:.3e}; {_where(o[:, 0], ref[:, 0])}"


W43_CONV_CASES = [
    dict(N=1, D=16, H=16, W=16, Cin=128, Cout=64, stats=True),                                  # 3-D: depth taps folded into K
    dict(N=1, D=8, H=8, W=8, Cin=64, Cout=64, aff=True, relu=True, stats=True),
    dict(N=2, D=16, H=16, W=16, Cin=64, Cout=64, aff=True, relu=True, per_n=1, stats=True, rpg=4096, ld_out=128),   # volumes of 2 queries, channel slice
    dict(N=2, D=5, H=9, W=11, Cin=16, Cout=64, act=1, stats=True, rpg=495),                      # odd sizes
    dict(N=7, D=1, H=32, W=32, Cin=192, Cout=128, stats=True, rpg=1024, ld_in=256, kd=1),       # 2-D layer, per-image statistics
    dict(N=7, D=1, H=16, W=16, Cin=256, Cout=64, aff=True, relu=True, per_n=1, stats=True, rpg=256, kd=1),
]


@pytest.mark.parametrize("w43_map", [2, 1, 0], ids=lambda m: f"map{m}")
@pytest.mark.parametrize("c", W43_CONV_CASES, ids=lambda c: f"{c['N']}x{c['D']}x{c['H']}_{c['Cin']}x{c['Cout']}")
def test_conv_on_wino43_kernel(ops, c, w43_map, knob):
    """ops.conv with G6dConv.weight_wino43: prologue (InstanceNorm affine, per-image tables), depth fold, statistics and their fused
    finalisation on the F(4x4,3x3) kernel against the fp64 reference — under every block-id decode of the kernel (knob w43_map: 2 = slices
    fastest (product default), 1 = XCD groups with a short last group, 0 = plain grid; the split-K tile counters and the finalize count
    depend on that decode: ADVICE r05)."""
    from gen6d_amd.network.backbone import winograd43_filters_taps
    knob("w43_map", w43_map)
    g = torch.Generator().manual_seed(43)
    N, D, H, W, Cin, Cout = c["N"], c["D"], c["H"], c["W"], c["Cin"], c["Cout"]
    kd = c.get("kd", 3)
    k, s, p = (kd, 3, 3), (1, 1, 1), (kd // 2, 1, 1)
    ld_in, ld_out = c.get("ld_in", Cin), c.get("ld_out", Cout)
    per_n = int(c.get("per_n", 0))
    xbuf = _rand(g, N, D, H, W, ld_in)
    off = 64 if ld_in >= Cin + 64 else 0
    T = kd * 9
    w = _rand(g, Cout, T, Cin, scale=(1.0 / (T * Cin)) ** 0.5)
    bias = _rand(g, Cout, scale=0.1)
    groups_in = (N + per_n - 1) // per_n if per_n else 1
    sc = (0.5 + torch.rand((groups_in, Cin), generator=g)) if c.get("aff") else None
    sh = _rand(g, groups_in, Cin, scale=0.3) if c.get("aff") else None
    M = N * D * H * W
    rpg = c.get("rpg", 0)
    G = M // rpg if rpg else 1
    dev = "cuda"
    xg = xbuf.to(dev)
    xv = xg[..., off:off + Cin]
    outbuf = torch.full((N, D, H, W, ld_out), -777.0, device=dev)
    out = outbuf[..., :Cout]
    count = float(rpg if rpg else M)
    ops.PROFILE = []
    try:
        stats = ops.new_stats(G, Cout, dev) if c.get("stats") else None
        res = ops.conv(xv, w.to(dev), bias.to(dev), out, ksize=k, stride=s, pad=p, in_scale=sc.to(dev) if sc is not None else None,
                       in_shift=sh.to(dev) if sh is not None else None, in_relu=bool(c.get("relu")), per_n=per_n, out_act=c.get("act", 0),
                       stats=stats, rows_per_group=rpg, w_wino43=winograd43_filters_taps(w, kd).to(dev),
                       finalize=count if stats is not None else None)
        torch.cuda.synchronize()
        assert ops.PROFILE[0][3].startswith("wino3x3 F43"), "descriptor was not routed to the F(4x4,3x3) kernel"
    finally:
        ops.PROFILE = None
    ref = torch.empty((N, D, H, W, Cout), dtype=torch.float64)
    rstats = torch.zeros((G, Cout, 2), dtype=torch.float64) if c.get("stats") else None
    ref_ops.conv(_d(xbuf[..., off:off + Cin]), _d(w), _d(bias), ref, ksize=k, stride=s, pad=p, in_scale=_d(sc), in_shift=_d(sh),
                 in_relu=bool(c.get("relu")), per_n=per_n, out_act=c.get("act", 0), stats=rstats, rows_per_group=rpg)
    e = _err(out, ref)
    assert e <= 1e-5, f"conv out {e:.3e}; {_where(out.reshape(N * D, H, W, Cout), ref.reshape(N * D, H, W, Cout))}"
    if ld_out != Cout:
        assert (outbuf[..., Cout:] == -777.0).all(), "conv wrote outside its channel slice"
    if stats is not None:
        assert _err(stats[..., 0], rstats[..., 0]) <= 2e-5 and _err(stats[..., 1], rstats[..., 1]) <= 2e-5
        sc_f, sh_f = res
        rsc, rsh = ref_ops.stats_finalize(rstats, count)
        assert _err(sc_f, rsc.double()) <= 1e-4 and _err(sh_f, rsh.double()) <= 1e-4
