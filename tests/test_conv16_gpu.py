"""g6d_conv16_direct_multi / g6d_vgg_conv1_pool_nhwc16 (ABI v11, round 6): the reduced-precision mode's direct convolution on 16-bit
activations, against the float64 convolution of the SAME rounded operands (the kernel's own arithmetic: exact products of 16-bit values,
fp32 accumulation — bar 2e-5 of the output range, plus half an ulp of the output type where the output is 16-bit) on ragged multi-map
launches, small maps that pack several images into a tile, pooled / full / fp32 / 16-bit outputs, 3x3x3 layers and statistics."""
import pytest
import torch
import torch.nn.functional as F

from parity_log import record

pytestmark = pytest.mark.gpu

T16 = {"bf16": torch.bfloat16, "fp16": torch.float16, "pairs": torch.float16}
ULP = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11, "pairs": 2.0 ** -21}
MODE = {"bf16": 1, "fp16": 2, "pairs": 3}


def _split(x):
    """fp32 tensor [..., C] -> fp16 hi / lo pairs [..., 2, C] (the format of math_mode 3)."""
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    return torch.stack([hi, lo], -2).contiguous()


def _join(p):
    return p[..., 0, :].double() + p[..., 1, :].double()


def _rand(g, *shape, scale=1.0):
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


CASES = [
    # segments (N, H, W) or (N, D, H, W); Cin; Cout; relu; full type; pool type; kd
    dict(segs=[(2, 16, 24)], Cin=64, Cout=128, relu=True, full="t16", pool="t16"),
    dict(segs=[(2, 22, 30), (1, 16, 20), (3, 6, 10)], Cin=128, Cout=256, relu=True, full="f32", pool="f32"),     # ragged widths, three maps
    dict(segs=[(5, 8, 8)], Cin=512, Cout=512, relu=False, full="t16", pool=None),                                 # two images per tile
    dict(segs=[(1, 44, 58), (2, 30, 40)], Cin=256, Cout=128, relu=True, full=None, pool="t16"),                   # pooled output only
    dict(segs=[(3, 4, 4)], Cin=64, Cout=128, relu=True, full="f32", pool="t16"),                                  # 4x4 maps: 8 images per tile
    dict(segs=[(2, 30, 34), (1, 6, 36)], Cin=64, Cout=128, relu=True, full="t16", pool="f32"),                    # heights that are no multiple of the tile
    dict(segs=[(3, 16, 16)], Cin=128, Cout=128, relu=False, full="f32", pool=None, stats=True, rpg=256),           # 2-D statistics per image
    dict(segs=[(6, 16, 16), (4, 8, 8)], Cin=64, Cout=64, relu=False, full="f32", pool=None, halo_only=True),        # Cout = 64: 32-channel waves
    dict(segs=[(2, 8, 8, 8)], Cin=64, Cout=128, relu=False, full="f32", pool=None, kd=3, stats=True),             # 3x3x3 with statistics
    dict(segs=[(1, 16, 16, 16)], Cin=128, Cout=128, relu=False, full="t16", pool=None, kd=3, stats=True),
]


@pytest.mark.parametrize("mode,layout,halo", [("fp16", 1, 1), ("bf16", 1, 1), ("pairs", 1, 1), ("fp16", 1, 0), ("pairs", 1, 0), ("fp16", 0, 0)],
                         ids=["fp16-halo", "bf16-halo", "pairs-halo", "fp16-regB", "pairs-regB", "fp16-ldsB"])
@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_conv16_direct_multi(mode, layout, halo, case, knob):
    """mode "pairs" = math_mode 3: every operand an fp16 hi / lo pair, fp32-CLASS results (the fp32 path's trunk kernel): the bar is 2e-6
    of the output range against the float64 convolution of the fp32 operands themselves (the fp32 Winograd kernels it replaces are held
    to 2e-5 / 4e-5)."""
    from gen6d_amd import ops
    knob("conv16_halo", halo)         # fragment-major filters: the halo-patch kernel (2-D layers) or the per-tap kernel
    if case.get("halo_only") and not (halo and layout == 1):
        pytest.skip("Cout = 64 runs on the halo-patch kernel only")
    c = case
    kd = c.get("kd", 1)
    t16 = T16[mode]
    pairs = mode == "pairs"
    g = torch.Generator().manual_seed(31 + c["Cin"] + len(c["segs"]))
    taps = 9 * kd
    w = _rand(g, c["Cout"], taps, c["Cin"], scale=(1.0 / (taps * c["Cin"])) ** 0.5 * 3)
    w = w if pairs else w.to(t16).float()
    b = _rand(g, c["Cout"], scale=0.2)
    xs = [_rand(g, *s, c["Cin"]) for s in c["segs"]]
    xs = xs if pairs else [x.to(t16).float() for x in xs]
    ty = {"t16": "t16", "f32": torch.float32, None: None}
    groups = sum(s[0] for s in c["segs"])
    stats = torch.zeros((c["segs"][0][0], c["Cout"], 2), dtype=torch.float64, device="cuda") if c.get("stats") else None
    rpg = 0
    if stats is not None:
        s0 = c["segs"][0]
        rpg = c.get("rpg") or s0[1] * s0[2] * s0[3]
    filt = ops.conv16_pack(w.cuda(), MODE[mode], layout)
    xin = [(_split(x) if pairs else x.to(t16)).cuda() for x in xs]
    fulls, pools = ops.conv16_direct_multi(xin, filt, b.cuda(), relu=c["relu"], full=ty[c["full"]], pool=ty[c["pool"]], kd=kd, stats=stats,
                                           rows_per_group=rpg)
    torch.cuda.synchronize()
    if pairs:
        fulls = [None if f is None else (_join(f.cpu()) if f.dtype != torch.float32 else f) for f in fulls]
        pools = [None if q is None else (_join(q.cpu()) if q.dtype != torch.float32 else q) for q in pools]
    base = 2e-6 if pairs else 2e-5
    worst = 0.0
    for i, x in enumerate(xs):
        xd = x.double()
        if kd == 1:
            w4 = w.double().reshape(c["Cout"], 3, 3, c["Cin"]).permute(0, 3, 1, 2)
            ref = F.conv2d(xd.permute(0, 3, 1, 2), w4, b.double(), padding=1).permute(0, 2, 3, 1)
        else:
            w5 = w.double().reshape(c["Cout"], 3, 3, 3, c["Cin"]).permute(0, 4, 1, 2, 3)
            ref = F.conv3d(xd.permute(0, 4, 1, 2, 3), w5, b.double(), padding=1).permute(0, 2, 3, 4, 1)
        if stats is not None and i == 0:
            s1 = ref.reshape(ref.shape[0], -1, c["Cout"]).sum(1)
            s2 = (ref * ref).reshape(ref.shape[0], -1, c["Cout"]).sum(1)
            got = stats.cpu()
            n = ref[0].numel() / c["Cout"]
            assert (got[:, :, 0] - s1).abs().max() / n <= base * ref.abs().max(), "statistics: sum"
            assert (got[:, :, 1] - s2).abs().max() / n <= 2 * base * ref.abs().max() ** 2, "statistics: sum of squares"
        if c["relu"]:
            ref = F.relu(ref)
        rng = float(ref.abs().max())
        if fulls[i] is not None:
            tol = base + (ULP[mode] if fulls[i].dtype != torch.float32 else 0.0)
            e = float((fulls[i].cpu().double() - ref).abs().max()) / rng
            worst = max(worst, e / tol)
            assert e <= tol, (i, "full", e, tol)
        if pools[i] is not None:
            pr = F.max_pool2d(ref.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
            tol = base + (ULP[mode] if pools[i].dtype != torch.float32 else 0.0)
            e = float((pools[i].cpu().double() - pr).abs().max()) / rng
            worst = max(worst, e / tol)
            assert e <= tol, (i, "pool", e, tol)
    record("test_conv16_direct_multi", f"{mode} layout {layout} {c['segs']} x{c['Cin']} -> {c['Cout']} kd={kd} (error / bar)", worst, 1.0,
           note="vs fp64 conv of the fp32 operands, bar 2e-6 of range" if pairs else "vs fp64 conv of the rounded operands")
    assert groups > 0


CORR_CASES = [
    dict(N=1, sizes=[(16, 16)], Cin=64, k=15),
    dict(N=3, sizes=[(22, 30), (9, 13)], Cin=64, k=15),                                   # ragged maps, two sizes
    dict(N=2, sizes=[(11, 15), (32, 40)], Cin=128, k=7),
    dict(N=1, sizes=[(88, 116), (60, 80), (44, 60), (32, 40)], Cin=512, k=15),           # the headline's pyramid, one query
]


@pytest.mark.parametrize("mode", ["fp16", "bf16", "pairs"])
@pytest.mark.parametrize("case", CORR_CASES, ids=lambda c: f"{c['N']}x{len(c['sizes'])}maps_{c['Cin']}_k{c['k']}")
def test_corr16_multi(mode, case):
    """The detector's k x k correlation on 16-bit activations against the fp64 direct correlation of the operands' own values
    (pairs: of the fp32 operands — fp32-class results, bar 2e-6 of range)."""
    from gen6d_amd import ops
    c = case
    pairs = mode == "pairs"
    t16 = T16[mode]
    g = torch.Generator().manual_seed(77 + c["Cin"] + c["k"])
    T = c["k"] * c["k"]
    w = _rand(g, 32, T, c["Cin"], scale=(3.0 / (T * c["Cin"])) ** 0.5)
    w = w if pairs else w.to(t16).float()
    xs = [_rand(g, c["N"], h, ww, c["Cin"]) for h, ww in c["sizes"]]
    xs = xs if pairs else [x.to(t16).float() for x in xs]
    filt = ops.corr16_pack(w.cuda(), MODE[mode])
    xin = [(_split(x) if pairs else x.to(t16)).cuda() for x in xs]
    outs = [torch.full((c["N"], 1, h, ww, 32), -5.0, device="cuda") for h, ww in c["sizes"]]
    ops.corr16_multi(xin, filt, outs)
    torch.cuda.synchronize()
    base = 2e-6 if pairs else 2e-5
    worst = 0.0
    for x, o in zip(xs, outs):
        w4 = w.double().reshape(32, c["k"], c["k"], c["Cin"]).permute(0, 3, 1, 2)
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w4, None, padding=c["k"] // 2).permute(0, 2, 3, 1)
        e = float((o[:, 0].cpu().double() - ref).abs().max()) / float(ref.abs().max())
        worst = max(worst, e / base)
        assert e <= base, (tuple(x.shape), e, base)
    record("test_corr16_multi", f"{mode} {c['sizes']} x{c['Cin']} k={c['k']} (error / bar)", worst, 1.0)


@pytest.mark.parametrize("mode", ["fp16", "bf16", "pairs"])
def test_vgg_conv1_pool_nhwc16(mode):
    """The first trunk layer with a 16-bit result equals the fp32 kernel's result rounded once (pairs: split once)."""
    from gen6d_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.rand((2, 3, 44, 60), generator=g).cuda()
    w = _rand(g, 64, 3, 3, 3, scale=0.3).cuda()
    b = _rand(g, 64, scale=0.1).cuda()
    norm = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    ref = ops.vgg_conv1_pool_nhwc(x, w, b, norm=norm)
    got = ops.vgg_conv1_pool_nhwc16(x, w, b, norm=norm, mode=MODE[mode])
    assert got.dtype == T16[mode]
    assert torch.equal(got, _split(ref) if mode == "pairs" else ref.to(T16[mode]))


def test_product_split16_and_cout64():
    """The selector's product layer on the direct kernel: g6d_product_split16 against its definition (pairs rebuild the fp32 product to
    2^-22), and a Cout = 64 pair convolution with per-query statistics against the fp64 convolution of the product."""
    from gen6d_amd import ops
    g = torch.Generator().manual_seed(5)
    qn, D, h, w, C, Cout = 2, 16, 8, 8, 64, 64
    ref, que = _rand(g, D, h * w, C), _rand(g, qn, h * w, C)
    sc, sh = 0.5 + torch.rand((qn, C), generator=g), _rand(g, qn, C, scale=0.3)
    want = ((ref[None] * que[:, None]) * sc[:, None, None] + sh[:, None, None]).reshape(qn * D, h * w, C)
    for mode, tol in ((3, 3e-7), (2, 1e-3)):
        got = ops.product_split16(ref.cuda(), que.cuda(), sc.cuda(), sh.cuda(), mode).cpu()
        val = _join(got) if mode == 3 else got.double()
        assert float((val - want.double()).abs().max()) <= tol * float(want.abs().max()), mode
    wt = _rand(g, Cout, 9, C, scale=(3.0 / (9 * C)) ** 0.5)
    b = _rand(g, Cout, scale=0.2)
    prod = ops.product_split16(ref.cuda(), que.cuda(), sc.cuda(), sh.cuda(), 3).view(qn * D, h, w, 2, C)
    stats = torch.zeros((qn, Cout, 2), dtype=torch.float64, device="cuda")
    fulls, _ = ops.conv16_direct_multi([prod], ops.conv16_pack(wt.cuda(), 3, 1), b.cuda(), relu=False, full=torch.float32, pool=None, stats=stats,
                                       rows_per_group=D * h * w)
    refc = F.conv2d(want.double().reshape(qn * D, h, w, C).permute(0, 3, 1, 2), wt.double().reshape(Cout, 3, 3, C).permute(0, 3, 1, 2), b.double(),
                    padding=1).permute(0, 2, 3, 1)
    e = float((fulls[0].cpu().double() - refc).abs().max()) / float(refc.abs().max())
    assert e <= 2e-6, e
    s1 = refc.reshape(qn, -1, Cout).sum(1)
    assert float((stats[:, :, 0].cpu() - s1).abs().max()) / (D * h * w) <= 2e-6 * float(refc.abs().max())
    record("test_product_split16_and_cout64", "pairs Cout=64 conv of the product (error / bar 2e-6)", e / 2e-6, 1.0)


@pytest.mark.parametrize("mode", ["pairs", "fp16"])
@pytest.mark.parametrize("pool", [False, True])
def test_affine_split16(mode, pool):
    """g6d_affine_split16 = affine_act_pool (per-group tables, ReLU, optional 2x2 max-pool) written in the conv16 activation format."""
    from gen6d_amd import ops
    import ref_ops
    g = torch.Generator().manual_seed(9)
    N, H, W, C, ld = 6, 8, 12, 64, 96
    buf = _rand(g, N, 1, H, W, ld)
    x = buf[..., 16:16 + C]
    sc, sh = 0.5 + torch.rand((3, C), generator=g), _rand(g, 3, C, scale=0.4)
    want = torch.empty((N, 1, H // 2, W // 2, C) if pool else (N, 1, H, W, C))
    ref_ops.affine_act_pool(x, want, sc, sh, per_n=2, relu=True, pool=1 if pool else 0)
    got = ops.affine_split16(buf.cuda()[..., 16:16 + C], sc.cuda(), sh.cuda(), 2, True, pool, MODE[mode]).cpu()
    val = _join(got) if mode == "pairs" else got.double()
    tol = 3e-7 if mode == "pairs" else 1e-3
    assert float((val - want[:, 0].double()).abs().max()) <= tol * float(want.abs().max())
