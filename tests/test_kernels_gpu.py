"""Per-kernel parity on a real MI355X: every C-ABI entry point (called through gen6d_amd.ops -> ctypes ->
libgen6d_hip.so) against its plain-PyTorch reference in tests/ref_ops.py evaluated in float64 on the CPU.

Tolerances: the kernels compute in fp32 (MFMA fp32 = k-ordered fmaf chain); the reference is fp64, so the bound is
fp32 accumulation error: |err| <= tol * (|a| . |b|) style magnitudes, written per test."""
import numpy as np
import pytest
import torch

import ref_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from gen6d_amd import lib, ops as _ops
    lib.load()                      # fails loudly if the HIP library is missing
    assert torch.cuda.is_available()
    return _ops


def _rand(g, *shape, scale=1.0):
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _d(t):
    return t.detach().cpu().double() if t is not None else None


def _check(got, want, tol, what=""):
    got, want = got.detach().cpu().double(), want.double()
    denom = max(want.abs().max().item(), 1e-30)
    err = (got - want).abs().max().item() / denom
    assert err <= tol, f"{what}: rel-to-max error {err:.3e} > {tol}"


CONV_CASES = [
    # N, D, H, W, Cin, Cout, k, stride, pad, extras
    dict(N=2, D=1, H=16, W=16, Cin=64, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1)),
    dict(N=3, D=1, H=9, W=7, Cin=32, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, stats=True),
    dict(N=1, D=8, H=8, W=8, Cin=128, Cout=64, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), aff=True, relu=True, stats=True),
    dict(N=1, D=8, H=8, W=8, Cin=64, Cout=256, k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1), stats=True),
    dict(N=1, D=4, H=4, W=4, Cin=256, Cout=512, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), aff=True, relu=True),       # M=64 tile
    dict(N=1, D=1, H=22, W=30, Cin=512, Cout=32, k=(1, 15, 15), s=(1, 1, 1), p=(0, 7, 7)),                          # detector level 0
    dict(N=1, D=1, H=11, W=15, Cin=512, Cout=8, k=(1, 7, 7), s=(1, 1, 1), p=(0, 3, 3)),                             # rfn=8
    dict(N=1, D=1, H=6, W=8, Cin=512, Cout=32, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1)),
    dict(N=40, D=1, H=8, W=8, Cin=512, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), mul=True, aff=True, stats=True),  # selector first layer
    dict(N=1, D=1, H=1, W=40, Cin=516, Cout=512, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), act=1),                     # Cin tail chunk
    dict(N=1, D=1, H=1, W=8, Cin=2580, Cout=512, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), act=1),
    dict(N=1, D=1, H=1, W=8, Cin=512, Cout=1, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0)),
    dict(N=1, D=1, H=16, W=16, Cin=64, Cout=2, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), ld_out=4, ld_in=192),
    dict(N=7, D=1, H=8, W=8, Cin=512, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), stats=True, rpg=64),         # groups straddle tiles
    dict(N=7, D=1, H=8, W=8, Cin=256, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=True, stats=True, rpg=64),
    dict(N=6, D=1, H=32, W=32, Cin=64, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=True, stats=True, rpg=1024),
    dict(N=1, D=1, H=1, W=64, Cin=1024, Cout=512, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), stats=True, act=0),
    dict(N=1, D=8, H=8, W=8, Cin=256, Cout=256, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), stats=True, split=1),        # forced no split
    dict(N=1, D=8, H=8, W=8, Cin=256, Cout=256, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), stats=True, split=7, act=2),
    dict(N=2, D=1, H=5, W=5, Cin=20, Cout=40, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), act=2),                        # odd sizes
    # split launches finish inside the kernel at every split count (the last block of a tile adds the partials)
    dict(N=1, D=1, H=10, W=10, Cin=512, Cout=96, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), stats=True, split=5, act=1),     # ragged tile, in-kernel
    dict(N=3, D=1, H=11, W=13, Cin=256, Cout=200, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), stats=True, rpg=143, split=16), # groups straddle tiles
    dict(N=1, D=4, H=4, W=4, Cin=256, Cout=512, k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1), stats=True, split=24),            # 24 splits, in-kernel
    dict(N=1, D=8, H=8, W=8, Cin=128, Cout=32, k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1), aff=True, relu=True, split=9),     # 128x32 tiles
    # shapes that take the LDS-patch kernel (conv_patch.hip): stride 1, 3x3(x3), Cout <= 64, >= 128 tiles
    dict(N=1, D=16, H=16, W=16, Cin=64, Cout=64, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), aff=True, relu=True, stats=True),
    dict(N=1, D=16, H=24, W=16, Cin=100, Cout=48, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), act=1, stats=True, ld_out=64),
    dict(N=1, D=15, H=17, W=16, Cin=32, Cout=64, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1)),                            # partial tiles
    dict(N=80, D=1, H=16, W=16, Cin=64, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), mul=True, aff=True, stats=True),
    dict(N=70, D=1, H=16, W=16, Cin=96, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, stats=True, rpg=256),
    dict(N=66, D=1, H=15, W=16, Cin=32, Cout=20, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), act=2),
    dict(N=7, D=1, H=32, W=32, Cin=128, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=True, stats=True, rpg=1024),  # per-image tables, chunk split
    dict(N=7, D=1, H=32, W=32, Cin=256, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), stats=True, rpg=1024),
    dict(N=321, D=1, H=8, W=8, Cin=64, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, stats=True),   # 2 images / tile, odd N
    dict(N=320, D=1, H=8, W=8, Cin=512, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), mul=True, aff=True, stats=True),
    dict(N=323, D=1, H=4, W=4, Cin=128, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True),             # 4x4 maps stay on the generic kernel
    dict(N=320, D=1, H=4, W=4, Cin=512, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), mul=True, aff=True, stats=True),
    # query batches (round 3): N = qn * k hypothesis images share the k input images (in_mod), every query has its own multiplier
    # map (mul_group), InstanceNorm table (per_n = k) and statistics group (rpg = k * rows per image)
    dict(N=120, D=1, H=4, W=4, Cin=512, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), mul=True, aff=True, per_n=40, in_mod=40, mul_group=40,
         stats=True, rpg=40 * 16),                                                                                       # generic kernel, MODE 4
    dict(N=46, D=1, H=4, W=4, Cin=128, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=23, stats=True, rpg=23 * 16),
    dict(N=80, D=1, H=16, W=16, Cin=64, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=20, stats=True, rpg=20 * 256),  # LDS-patch kernel
    dict(N=66, D=1, H=8, W=8, Cin=64, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=22, stats=True, rpg=22 * 64),    # 2 images / tile
    dict(N=63, D=1, H=8, W=8, Cin=64, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=21, stats=True, rpg=21 * 64),    # odd group: 1 image / tile
    dict(N=2, D=16, H=16, W=16, Cin=64, Cout=64, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), aff=True, relu=True, per_n=1, stats=True, rpg=4096),       # batch of volumes
    dict(N=3, D=8, H=8, W=8, Cin=64, Cout=128, k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1), aff=True, relu=True, per_n=1, stats=True, rpg=64),
    dict(N=3, D=1, H=1, W=64, Cin=512, Cout=512, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), aff=True, relu=True, per_n=1, stats=True, rpg=64),        # selector tail, 3 queries
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"{c['Cin']}x{c['Cout']}k{c['k'][0]}{c['k'][1]}{c['k'][2]}")
def test_conv_igemm(ops, case):
    _run_conv_case(ops, case, False)


# Layers with at most four output channels (the detector heads' merged last conv, reference network/detector.py:164-184) run on the
# vector-ALU dot-product kernel (conv_narrow_kernel, round 6): odd widths (partial pixel runs), Cin with idle lanes, a channel slice of
# a wider input, every activation, and the matrix-core fall-back under the knob.
NARROW_CASES = [
    dict(N=3, D=1, H=9, W=13, Cin=192, Cout=4, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1)),
    dict(N=2, D=1, H=5, W=8, Cin=64, Cout=1, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), act=1),
    dict(N=1, D=1, H=16, W=21, Cin=256, Cout=3, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), act=2, ld_in=320, ld_out=8),
    dict(N=16, D=1, H=60, W=80, Cin=192, Cout=4, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1)),                 # the headline's head layer
]


@pytest.mark.parametrize("narrow", [1, 0], ids=["vector-ALU", "matrix-core"])
@pytest.mark.parametrize("case", NARROW_CASES, ids=lambda c: f"{c['N']}x{c['H']}x{c['W']}_{c['Cin']}x{c['Cout']}")
def test_conv_narrow(ops, case, narrow, knob):
    knob("conv_narrow", narrow)
    _run_conv_case(ops, case, False)


# Position-major tiles of conv_igemm (round 5): small 2-D maps with >= 4 tiles of images — a tile is one output position of 128 (64)
# consecutive images and the K loop skips the taps that fall into the zero padding.  Ragged image counts (partial last tile, partial
# last group of 8 tiles), every operand prologue, statistics groups that straddle tiles, and the forced fall-backs.
PM_CASES = [
    dict(N=600, D=1, H=4, W=4, Cin=128, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=200, stats=True, rpg=200 * 16),
    dict(N=1040, D=1, H=4, W=4, Cin=512, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), mul=True, aff=True, per_n=260, in_mod=260, mul_group=260,
         stats=True, rpg=260 * 16),                                                                    # 9 tiles: a full group of 8 + a tail of 1
    dict(N=1285, D=1, H=4, W=4, Cin=256, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=257),   # odd N, no statistics
    dict(N=520, D=1, H=8, W=8, Cin=64, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), stats=True, rpg=130 * 64),
    dict(N=515, D=1, H=5, W=3, Cin=20, Cout=40, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), act=2),                               # odd map, Cin tail chunk
    dict(N=512, D=1, H=2, W=6, Cin=64, Cout=96, k=(1, 1, 5), s=(1, 1, 1), p=(0, 0, 2), aff=True, relu=True, stats=True),     # 1x5 taps, one table
    dict(N=512, D=1, H=4, W=4, Cin=128, Cout=32, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), ld_out=48),                          # 128x32 tiles, channel slice
]


@pytest.mark.parametrize("pm", [2, 0], ids=["position-major", "row-order"])
@pytest.mark.parametrize("case", PM_CASES, ids=lambda c: f"{c['N']}x{c['H']}x{c['W']}_{c['Cin']}x{c['Cout']}")
def test_conv_igemm_position_major(ops, case, pm, knob):
    knob("conv_patch", 0)            # (8x8 maps with Cout <= 64-channel tiles would take the LDS-patch kernel)
    knob("conv_pm", pm)
    _run_conv_case(ops, case, False)


# Layers that g6d_conv_igemm routes to the Winograd kernel when G6dConv.weight_wino is given (every prologue / epilogue variant the
# selector, the refiner feature net and the volume net use), plus shapes that must fall back to the direct kernels.
WINO_CONV_CASES = [
    dict(N=20, D=1, H=16, W=16, Cin=512, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), mul=True, aff=True, stats=True),      # selector product layer
    dict(N=21, D=1, H=16, W=16, Cin=64, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, stats=True),
    dict(N=13, D=1, H=8, W=8, Cin=64, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), stats=True),                           # one quarter per image
    dict(N=320, D=1, H=8, W=8, Cin=128, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, stats=True),
    dict(N=7, D=1, H=32, W=32, Cin=256, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), stats=True, rpg=1024),                # per-image statistics
    dict(N=7, D=1, H=16, W=16, Cin=256, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=True, stats=True, rpg=256),
    dict(N=7, D=1, H=8, W=8, Cin=512, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), stats=True, rpg=64),                    # groups inside a block
    dict(N=7, D=1, H=32, W=32, Cin=192, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), stats=True, rpg=1024, ld_in=256),
    dict(N=1, D=16, H=16, W=16, Cin=128, Cout=64, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), stats=True),                          # 3-D: depth taps folded into K
    dict(N=1, D=8, H=8, W=8, Cin=256, Cout=256, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), aff=True, relu=True, stats=True),
    dict(N=1, D=32, H=32, W=32, Cin=64, Cout=64, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), aff=True, relu=True, ld_out=128),     # channel slice of the concat buffer
    dict(N=2, D=5, H=9, W=11, Cin=16, Cout=32, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), act=1, stats=True, rpg=495),             # odd sizes, 32 output channels
    dict(N=3, D=1, H=4, W=4, Cin=128, Cout=256, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True),                   # not eligible: 4x4 map
    dict(N=1, D=8, H=8, W=8, Cin=64, Cout=128, k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1), stats=True),                             # not eligible: stride 2
    # query batches on the Winograd kernel (tables of the block's four quarters in LDS; blocks straddle query groups)
    dict(N=60, D=1, H=16, W=16, Cin=512, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), mul=True, aff=True, per_n=20, in_mod=20, mul_group=20,
         stats=True, rpg=20 * 256),
    dict(N=26, D=1, H=8, W=8, Cin=512, Cout=128, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), mul=True, aff=True, per_n=13, in_mod=13, mul_group=13,
         stats=True, rpg=13 * 64),
    dict(N=63, D=1, H=16, W=16, Cin=64, Cout=64, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), aff=True, relu=True, per_n=21, stats=True, rpg=21 * 256),
    dict(N=2, D=16, H=16, W=16, Cin=128, Cout=64, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), aff=True, relu=True, per_n=1, stats=True, rpg=4096),  # volumes of 2 queries
    dict(N=3, D=8, H=8, W=8, Cin=256, Cout=256, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), aff=True, relu=True, per_n=1, stats=True, rpg=512),
]


@pytest.mark.parametrize("case", WINO_CONV_CASES, ids=lambda c: f"{c['N']}x{c['D']}x{c['H']}_{c['Cin']}x{c['Cout']}k{c['k'][0]}")
def test_conv_on_winograd_kernel(ops, case, knob):
    """Knob wino_min_work = 0 lifts the library's profitability rule so that the small test shapes take the Winograd path; the
    stage-level parity tests run with the production rule."""
    knob("wino_min_work", 0)
    ops.PROFILE = []
    try:
        _run_conv_case(ops, case, True)
        routed = ops.PROFILE[0][3].startswith("wino3x3")
    finally:
        ops.PROFILE = None
    assert routed == (case["H"] >= 6 and case["s"] == (1, 1, 1)), "unexpected kernel family for this descriptor"


def _run_conv_case(ops, case, wino):
    g = torch.Generator().manual_seed(17)
    c = case
    N, D, H, W, Cin, Cout = c["N"], c["D"], c["H"], c["W"], c["Cin"], c["Cout"]
    k, s, p = c["k"], c["s"], c["p"]
    ld_in, ld_out = c.get("ld_in", Cin), c.get("ld_out", Cout)
    Do, Ho, Wo = [(i + 2 * pp - kk) // ss + 1 for i, kk, ss, pp in zip((D, H, W), k, s, p)]
    in_mod, mul_group, per_n = c.get("in_mod", 0), c.get("mul_group", 0), int(c.get("per_n", 0))
    xbuf = _rand(g, in_mod or N, D, H, W, ld_in)
    off = 64 if ld_in >= Cin + 64 else 0
    x = xbuf[..., off:off + Cin]
    T = k[0] * k[1] * k[2]
    w = _rand(g, Cout, T, Cin, scale=(1.0 / (T * Cin)) ** 0.5)
    bias = _rand(g, Cout, scale=0.1)
    mul = (_rand(g, (N + mul_group - 1) // mul_group, H, W, Cin) if mul_group else _rand(g, H, W, Cin)) if c.get("mul") else None
    groups_in = (N + per_n - 1) // per_n if per_n else 1
    sc = (0.5 + torch.rand((groups_in, Cin), generator=g)) if c.get("aff") else None
    sh = _rand(g, groups_in, Cin, scale=0.3) if c.get("aff") else None
    M = N * Do * Ho * Wo
    rpg = c.get("rpg", 0)
    G = M // rpg if rpg else 1
    dev = "cuda"
    xg = xbuf.to(dev)
    xv = xg[..., off:off + Cin]
    outbuf = torch.full((N, Do, Ho, Wo, ld_out), -777.0, device=dev)
    out = outbuf[..., :Cout]
    stats = ops.new_stats(G, Cout, dev) if c.get("stats") else None
    count = float(rpg if rpg else M)
    res = ops.conv(xv, w.to(dev), bias.to(dev), out, ksize=k, stride=s, pad=p, mul=mul.to(dev) if mul is not None else None,
                   in_scale=sc.to(dev) if sc is not None else None, in_shift=sh.to(dev) if sh is not None else None,
                   in_relu=bool(c.get("relu")), per_n=per_n, out_act=c.get("act", 0), stats=stats,
                   rows_per_group=rpg, split_k=c.get("split", 0), w_wino=_wino_u(w, k).to(dev) if wino else None,
                   finalize=count if stats is not None else None,          # InstanceNorm affine from the launch's last block
                   in_mod=in_mod, mul_group=mul_group)
    torch.cuda.synchronize()
    ref = torch.empty((N, Do, Ho, Wo, Cout), dtype=torch.float64)
    rstats = torch.zeros((G, Cout, 2), dtype=torch.float64) if c.get("stats") else None
    ref_ops.conv(_d(x), _d(w), _d(bias), ref, ksize=k, stride=s, pad=p, mul=_d(mul), in_scale=_d(sc), in_shift=_d(sh),
                 in_relu=bool(c.get("relu")), per_n=per_n, out_act=c.get("act", 0), stats=rstats,
                 rows_per_group=rpg, in_mod=in_mod, mul_group=mul_group)
    _check(out, ref, 4e-5 if wino else 2e-5, "conv out")
    if ld_out != Cout:
        assert (outbuf[..., Cout:] == -777.0).all(), "conv wrote outside its channel slice"
    if stats is not None:
        _check(stats[..., 0], rstats[..., 0], 4e-5 if wino else 2e-5, "stats sum")
        _check(stats[..., 1], rstats[..., 1], 4e-5 if wino else 2e-5, "stats sumsq")
        # the fused finalisation equals the stand-alone kernel on the same table, and the reference formula
        sc_f, sh_f = res
        sc_k, sh_k = ops.stats_finalize(stats, count)
        assert torch.equal(sc_f, sc_k) and torch.equal(sh_f, sh_k), "fused InstanceNorm finalisation differs from g6d_stats_finalize"
        rsc, rsh = ref_ops.stats_finalize(rstats, count)
        _check(sc_f, rsc.double(), 1e-4, "finalised scale"); _check(sh_f, rsh.double(), 1e-4, "finalised shift")


def _wino_u(w_taps, k):
    from gen6d_amd.network.backbone import winograd_filters_taps
    return winograd_filters_taps(w_taps, k[0])


@pytest.mark.parametrize("N,H,W", [(1, 128, 128), (2, 50, 70), (1, 33, 67), (3, 2, 2), (1, 96, 160)])
def test_vgg_conv1_pool(ops, N, H, W):
    """Fused first trunk layer against conv2d + bias + ReLU + max_pool2d in fp64 (odd sizes pool with floor)."""
    g = torch.Generator().manual_seed(5)
    x = _rand(g, N, 3, H, W)
    w = _rand(g, 64, 3, 3, 3, scale=0.3)
    b = _rand(g, 64, scale=0.2)
    out = ops.vgg_conv1_pool(x.cuda(), w.cuda(), b.cuda())
    assert tuple(out.shape) == (N, 64, H // 2, W // 2)
    _check(out, ref_ops.vgg_conv1_pool(_d(x), _d(w), _d(b)), 1e-5, "vgg_conv1_pool")
    with pytest.raises(RuntimeError, match="G6D_EINVAL"):
        ops.vgg_conv1_pool(torch.zeros((1, 4, 8, 8), device="cuda"), torch.zeros((64, 4, 3, 3), device="cuda"), b.cuda())


def test_conv_rejects_bad_args(ops):
    x = torch.zeros((1, 1, 4, 4, 6), device="cuda")          # Cin % 4 != 0
    w = torch.zeros((8, 1, 6), device="cuda")
    out = torch.zeros((1, 1, 4, 4, 8), device="cuda")
    with pytest.raises(RuntimeError, match="G6D_EINVAL"):
        ops.conv(x, w, None, out)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.conv(x.cpu(), w, None, out)


def test_stats_finalize(ops):
    g = torch.Generator().manual_seed(3)
    st = torch.rand((3, 40, 2), generator=g, dtype=torch.float64)
    st[..., 0] = (st[..., 0] - 0.5) * 200
    st[..., 1] = st[..., 0] ** 2 / 100 + st[..., 1] * 50 + 1
    sc, sh = ops.stats_finalize(st.cuda(), 100.0)
    rsc, rsh = ref_ops.stats_finalize(st, 100.0)
    _check(sc, rsc, 1e-6); _check(sh, rsh, 1e-6)


@pytest.mark.parametrize("pool,per_n,relu", [(0, False, True), (1, False, False), (1, True, True), (2, False, True), (0, True, False)])
def test_affine_act_pool(ops, pool, per_n, relu):
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 6, 8, 8, 64
    x = _rand(g, N, 1, H, W, C)
    sc = 0.5 + torch.rand((N if per_n else 1, C), generator=g)
    sh = _rand(g, N if per_n else 1, C)
    shape = {0: (N, 1, H, W, C), 1: (N, 1, H // 2, W // 2, C), 2: (N, 1, 1, 1, C)}[pool]
    obuf = torch.zeros(shape[:-1] + (C + 8,), device="cuda")
    out = obuf[..., 4:4 + C]
    ops.affine_act_pool(x.cuda(), out, sc.cuda(), sh.cuda(), per_n=per_n, relu=relu, pool=pool)
    ref = torch.empty(shape, dtype=torch.float64)
    ref_ops.affine_act_pool(_d(x), ref, _d(sc), _d(sh), per_n=per_n, relu=relu, pool=pool)
    _check(out, ref, 1e-6)
    # identity affine
    out2 = torch.empty((N, 1, H, W, C), device="cuda")
    ops.affine_act_pool(x.cuda(), out2)
    _check(out2, x, 0)
    if per_n:                                   # a table per run of 3 images (query groups of a batch)
        sc3, sh3 = 0.5 + torch.rand((2, C), generator=g), _rand(g, 2, C)
        ops.affine_act_pool(x.cuda(), out, sc3.cuda(), sh3.cuda(), per_n=3, relu=relu, pool=pool)
        ref_ops.affine_act_pool(_d(x), ref, _d(sc3), _d(sh3), per_n=3, relu=relu, pool=pool)
        _check(out, ref, 1e-6)


@pytest.mark.parametrize("factor", [2, 4])
def test_upsample_bilinear(ops, factor):
    g = torch.Generator().manual_seed(6)
    N, H, W, C = 7, 8, 8, 64
    x = _rand(g, N, 1, H, W, C)
    sc, sh = 0.5 + torch.rand((N, C), generator=g), _rand(g, N, C)
    obuf = torch.zeros((N, 1, H * factor, W * factor, 192), device="cuda")
    out = obuf[..., 64:128]
    ops.upsample_bilinear(x.cuda(), out, factor, sc.cuda(), sh.cuda(), per_n=True)
    ref = torch.empty((N, 1, H * factor, W * factor, C), dtype=torch.float64)
    ref_ops.upsample_bilinear(_d(x), ref, factor, _d(sc), _d(sh), per_n=True)
    _check(out, ref, 1e-6)
    assert (obuf[..., :64] == 0).all() and (obuf[..., 128:] == 0).all()


@pytest.mark.parametrize("l2", [0, 1])
def test_nchw_to_nhwc(ops, l2):
    g = torch.Generator().manual_seed(7)
    x = _rand(g, 3, 512, 5, 7)
    out = torch.empty((3, 1, 5, 7, 512), device="cuda")
    ops.nchw_to_nhwc(x.cuda(), out, bool(l2))
    ref = torch.empty((3, 1, 5, 7, 512), dtype=torch.float64)
    ref_ops.nchw_to_nhwc(_d(x), ref, bool(l2))
    _check(out, ref, 1e-6)


def test_selector_similarity(ops):
    g = torch.Generator().manual_seed(8)
    D, HW, C = 40, 64, 512
    refs = torch.nn.functional.normalize(_rand(g, D, HW, C), dim=2)
    que = torch.nn.functional.normalize(_rand(g, HW, C) + 0.3, dim=1)
    r1, r2 = ops.selector_ref_sums(refs.cuda())
    rr1, rr2 = ref_ops.selector_ref_sums(_d(refs))
    _check(r1, rr1, 1e-12); _check(r2, rr2, 1e-12)
    sc, sh = ops.selector_prod_affine(que.cuda(), r1, r2, D)
    # ground truth: statistics of the materialised product
    prod = _d(refs) * _d(que)[None]
    mean, var = prod.mean((0, 1)), prod.var((0, 1), unbiased=False)
    _check(sc[0], 1 / torch.sqrt(var + 1e-5), 1e-5)
    _check(sh[0], -mean / torch.sqrt(var + 1e-5), 1e-5)
    smap, vps = ops.selector_scan(que.cuda(), refs.cuda())
    rsmap, rvps = ref_ops.selector_scan(_d(que), _d(refs))
    _check(smap, rsmap, 2e-6); _check(vps, rvps, 1e-5)


@pytest.mark.parametrize("want_maps", [False, True])
def test_selector_levels(ops, want_maps):
    """All three pyramid levels in one launch (scores, viewpoint scores, product statistics) against the materialised product."""
    g = torch.Generator().manual_seed(18)
    D, C, Dg = 45, 512, 45
    hws = [256, 64, 16]
    refs = [torch.nn.functional.normalize(_rand(g, D, hw, C), dim=2) for hw in hws]
    ques = [torch.nn.functional.normalize(_rand(g, hw, C) + 0.3, dim=1) for hw in hws]
    sums = [ops.selector_ref_sums(r.cuda()) for r in refs]
    vps, sc, sh, maps = ops.selector_levels([q.cuda() for q in ques], [r.cuda() for r in refs], sums, Dg, want_maps=want_maps)
    assert (maps is not None) == want_maps
    for l, (q, r) in enumerate(zip(ques, refs)):
        prod = _d(r) * _d(q)[None]
        mean, var = prod.mean((0, 1)), prod.var((0, 1), unbiased=False)
        _check(sc[l], 1 / torch.sqrt(var + 1e-5), 1e-5, f"scale level {l}")
        _check(sh[l], -mean / torch.sqrt(var + 1e-5), 1e-5, f"shift level {l}")
        rsmap, rvps = ref_ops.selector_scan(_d(q), _d(r))
        _check(vps[l], rvps, 1e-5, f"vps level {l}")
        if want_maps:
            _check(maps[l], rsmap, 2e-6, f"score map level {l}")
        # same numbers as the per-level entry points (the reduction over HW runs in another kernel: float reassociation)
        smap1, vps1 = ops.selector_scan(q.cuda(), r.cuda())
        _check(vps1, vps[l].double().cpu(), 2e-6, "per-level entry point")


@pytest.mark.parametrize("qn,D", [(2, 45), (4, 33), (7, 10)])
def test_selector_levels_query_batch(ops, qn, D):
    """qn queries against one streamed pass over the reference cache: every query's vps / product statistics / score maps equal
    those of its own single-query call."""
    g = torch.Generator().manual_seed(180 + qn)
    C, Dg = 512, D
    hws = [256, 64, 16]
    refs = [torch.nn.functional.normalize(_rand(g, D, hw, C), dim=2).cuda() for hw in hws]
    ques = [torch.nn.functional.normalize(_rand(g, qn, hw, C) + 0.3, dim=2).cuda() for hw in hws]
    sums = [ops.selector_ref_sums(r) for r in refs]
    vps, sc, sh, maps = ops.selector_levels(ques, refs, sums, Dg, want_maps=True)
    assert vps.shape == (qn, 3, D) and sc.shape == (qn, 3, C) and maps[0].shape == (qn, D, 256)
    for q in range(qn):
        v1, sc1, sh1, m1 = ops.selector_levels([t[q].contiguous() for t in ques], refs, sums, Dg, want_maps=True)
        # (not bit-equal: the batch-width template instantiations order the per-row FMA chains differently)
        _check(vps[q], v1.double().cpu(), 2e-6, "vps vs single"); _check(sc[q], sc1.double().cpu(), 1e-6, "scale vs single")
        _check(sh[q], sh1.double().cpu(), 1e-6, "shift vs single")
        for l in range(3):
            _check(maps[l][q], m1[l].double().cpu(), 2e-6, "score map vs single")
            rsmap, rvps = ref_ops.selector_scan(_d(ques[l][q]), _d(refs[l]))
            _check(vps[q, l], rvps, 1e-5, f"vps q{q} l{l}")


@pytest.mark.parametrize("rfn", [6, 1, 8])
def test_refiner_volume(ops, rfn):
    from gen6d_amd import synth
    g = torch.Generator().manual_seed(9)
    case = synth.refiner_case(rfn=rfn)
    sn, C, fh, fw = 16, 128, 32, 32
    feats = _rand(g, rfn + 1, fh, fw, C)
    projs = torch.cat([case["ref_Ks"][0] @ case["ref_poses"][0], case["Ks_in"] @ case["poses_in"]], 0).contiguous()
    rot = case["poses_in"][0, :, :3].contiguous()
    lin = torch.linspace(-1, 1, sn)
    mean_in = torch.empty((sn ** 3, 2 * C), device="cuda"); std = torch.empty((sn ** 3, C), device="cuda")
    ops.refiner_volume(feats.cuda(), projs.cuda(), rot.cuda(), lin.cuda(), 128, 128, mean_in, std)
    rm = torch.empty((sn ** 3, 2 * C), dtype=torch.float64); rs = torch.empty((sn ** 3, C), dtype=torch.float64)
    ref_ops.refiner_volume(_d(feats), _d(projs), _d(rot), _d(lin), 128, 128, rm, rs)
    # bilinear weights are computed from fp32 coordinates; 1e-4 of the feature range covers that
    _check(mean_in, rm, 2e-4, "mean|query")
    _check(std, rs, 2e-4, "std")
    # the variant that forms K @ pose inside the kernel and reads the rotation from pose_in
    m2 = torch.empty_like(mean_in); s2 = torch.empty_like(std)
    ops.refiner_volume_kp(feats.cuda(), case["ref_Ks"][0].contiguous().cuda(), case["ref_poses"][0].contiguous().cuda(),
                          case["Ks_in"][0].contiguous().cuda(), case["poses_in"][0].contiguous().cuda(), lin.cuda(), 128, 128, m2, s2)
    _check(m2, rm, 2e-4, "mean|query (kp)")
    _check(s2, rs, 2e-4, "std (kp)")
    # a batch of 3 queries (own views, cameras and input poses) in one launch equals the three single launches
    B = 3
    fb = torch.stack([feats, feats.flip(0), feats * 0.5], 0).contiguous().cuda()
    Kb = torch.stack([case["ref_Ks"][0]] * B, 0).contiguous().cuda()
    Pb = torch.stack([case["ref_poses"][0], case["ref_poses"][0].flip(0), case["ref_poses"][0]], 0).contiguous().cuda()
    Kin = torch.stack([case["Ks_in"][0]] * B, 0).contiguous().cuda()
    pin = torch.stack([case["poses_in"][0], torch.from_numpy(synth.perturb_pose(case["poses_in"][0].numpy(), 3.0, 0.02)), case["poses_in"][0]], 0).contiguous().cuda()
    mb = torch.empty((B, sn ** 3, 2 * C), device="cuda"); sb = torch.empty((B, sn ** 3, C), device="cuda")
    ops.refiner_volume_kp(fb, Kb, Pb, Kin, pin, lin.cuda(), 128, 128, mb, sb)
    for b in range(B):
        ops.refiner_volume_kp(fb[b].contiguous(), Kb[b].contiguous(), Pb[b].contiguous(), Kin[b].contiguous(), pin[b].contiguous(), lin.cuda(), 128, 128, m2, s2)
        assert torch.equal(mb[b], m2) and torch.equal(sb[b], s2), f"batched volume {b} differs from its single launch"


def test_detector_glue(ops):
    g = torch.Generator().manual_seed(10)
    hc, wc, rfn, hs, ws = 12, 20, 8, 16, 16
    stats = [[36.264317, 13.151907], [13910.291, 5345.965], [829.70807, 387.98788]]
    s = [_rand(g, (hc >> l) * (wc >> l), rfn, scale=3 * stats[l][1]) + stats[l][0] for l in range(3)]
    stacked = torch.zeros((hs * ws, rfn, 12), device="cuda")
    rstacked = torch.zeros((hs * ws, rfn, 12), dtype=torch.float64)
    for si in range(4):
        ops.detector_assemble(s[0].cuda(), s[1].cuda(), s[2].cuda(), hc, wc, stats, 10.0, hs, ws, si, stacked)
        ref_ops.detector_assemble(_d(s[0]), _d(s[1]), _d(s[2]), hc, wc, stats, 10.0, hs, ws, si, rstacked)
    _check(stacked, rstacked, 1e-5, "assemble")
    assert (stacked.abs().max() <= 10.0 + 1e-6)
    w0, b0, w1, b1 = _rand(g, 64, 12, scale=0.3), _rand(g, 64, scale=0.1), _rand(g, 64, 64, scale=0.2), _rand(g, 64, scale=0.1)
    out = ops.detector_score_mlp_max(stacked, w0.cuda(), b0.cuda(), w1.cuda(), b1.cuda())
    ref = ref_ops.detector_score_mlp_max(rstacked, _d(w0), _d(b0), _d(w1), _d(b1))
    _check(out, ref, 1e-5, "mlp max")
    # decode incl. first-maximum tie rule
    o4 = _rand(g, hs * ws, 4)
    o4[37, 0] = 5.0; o4[101, 0] = 5.0
    res = ops.detector_decode(o4.cuda()[:, 0:1], o4.cuda()[:, 2:4], o4.cuda()[:, 1:2], hs, ws, 8)
    rres = ref_ops.detector_decode(_d(o4)[:, 0:1], _d(o4)[:, 2:4], _d(o4)[:, 1:2], hs, ws, 8)
    assert int(res[3].item()) == 37 % ws and int(res[4].item()) == 37 // ws
    _check(res, rres, 1e-6, "decode")
    # a batch of 3 queries: level maps / stacked slabs / decode rows of the queries one after the other
    B = 3
    sb = [_rand(g, B * (hc >> l) * (wc >> l), rfn, scale=3 * stats[l][1]) + stats[l][0] for l in range(3)]
    stb = torch.zeros((B * hs * ws, rfn, 12), device="cuda"); rstb = torch.zeros((B * hs * ws, rfn, 12), dtype=torch.float64)
    for si in range(4):
        ops.detector_assemble(sb[0].cuda(), sb[1].cuda(), sb[2].cuda(), hc, wc, stats, 10.0, hs, ws, si, stb, batch=B)
        ref_ops.detector_assemble(_d(sb[0]), _d(sb[1]), _d(sb[2]), hc, wc, stats, 10.0, hs, ws, si, rstb, batch=B)
    _check(stb, rstb, 1e-5, "assemble batch")
    o4b = _rand(g, B * hs * ws, 4)
    resb = ops.detector_decode(o4b.cuda()[:, 0:1], o4b.cuda()[:, 2:4], o4b.cuda()[:, 1:2], hs, ws, 8, batch=B)
    rresb = ref_ops.detector_decode(_d(o4b)[:, 0:1], _d(o4b)[:, 2:4], _d(o4b)[:, 1:2], hs, ws, 8, batch=B)
    assert resb.shape == (B, 5)
    _check(resb, rresb, 1e-6, "decode batch")


def test_selector_tail_ops(ops):
    g = torch.Generator().manual_seed(11)
    rfn, an, C = 16, 5, 512
    D = rfn * an
    vps = _rand(g, 3, D, scale=20) + 30
    feats = torch.zeros((D, 516), device="cuda"); rfeats = torch.zeros((D, 516), dtype=torch.float64)
    ops.vps_norm(vps.cuda(), feats, 512); ref_ops.vps_norm(_d(vps), rfeats, 512)
    _check(feats, rfeats, 1e-5, "vps_norm")
    x, emb = _rand(g, D, C), _rand(g, rfn, C)
    out = torch.zeros((rfn, 1024), device="cuda"); rout = torch.zeros((rfn, C), dtype=torch.float64)
    ops.max_an_add(x.cuda(), rfn, an, emb.cuda(), out[:, :512]); ref_ops.max_an_add(_d(x), rfn, an, _d(emb), rout)
    _check(out[:, :512], rout, 1e-6, "max_an_add")
    qkv = _rand(g, rfn, 1536)
    att = torch.empty((rfn, C), device="cuda"); ratt = torch.empty((rfn, C), dtype=torch.float64)
    qg = qkv.cuda()
    ops.attention(qg[:, :512], qg[:, 512:1024], qg[:, 1024:], 8, att)
    qd = _d(qkv)
    ref_ops.attention(qd[:, :512], qd[:, 512:1024], qd[:, 1024:], 8, ratt)
    _check(att, ratt, 1e-5, "attention")
    gam, bet = 0.5 + torch.rand(C, generator=g), _rand(g, C)
    ln = torch.empty((rfn, C), device="cuda"); rln = torch.empty((rfn, C), dtype=torch.float64)
    ops.layernorm(x[:rfn].cuda(), gam.cuda(), bet.cuda(), ln); ref_ops.layernorm(_d(x[:rfn]), _d(gam), _d(bet), rln)
    _check(ln, rln, 1e-5, "layernorm")
    sc, sh = 0.5 + torch.rand((1, C), generator=g), _rand(g, 1, C)
    o = torch.empty((rfn, C), device="cuda"); ro = torch.empty((rfn, C), dtype=torch.float64)
    ops.affine_act_add(x[:rfn].cuda(), o, sc.cuda(), sh.cuda(), relu=True, residual=emb.cuda())
    ref_ops.affine_act_add(_d(x[:rfn]), ro, _d(sc), _d(sh), relu=True, residual=_d(emb))
    _check(o, ro, 1e-6, "affine_act_add")
    # ---- the same helpers on a batch of B queries (row blocks of the queries one after the other)
    B = 3
    vb = _rand(g, B, 3, D, scale=20) + 30
    fb = torch.zeros((B * D, 516), device="cuda"); rfb = torch.zeros((B * D, 516), dtype=torch.float64)
    ops.vps_norm(vb.cuda(), fb, 512); ref_ops.vps_norm(_d(vb), rfb, 512)
    _check(fb, rfb, 1e-5, "vps_norm batch")
    xb = _rand(g, B * D, C)
    ob = torch.zeros((B * rfn, 1024), device="cuda"); rob = torch.zeros((B * rfn, C), dtype=torch.float64)
    ops.max_an_add(xb.cuda(), rfn, an, emb.cuda(), ob[:, :512], batch=B); ref_ops.max_an_add(_d(xb), rfn, an, _d(emb), rob, batch=B)
    _check(ob[:, :512], rob, 1e-6, "max_an_add batch")
    qb = _rand(g, B * rfn, 1536).cuda()
    ab = torch.empty((B * rfn, C), device="cuda"); rab = torch.empty((B * rfn, C), dtype=torch.float64)
    ops.attention(qb[:, :512], qb[:, 512:1024], qb[:, 1024:], 8, ab, batch=B)
    qbd = _d(qb)
    ref_ops.attention(qbd[:, :512], qbd[:, 512:1024], qbd[:, 1024:], 8, rab, batch=B)
    _check(ab, rab, 1e-5, "attention batch")
    scb, shb = 0.5 + torch.rand((B, C), generator=g), _rand(g, B, C)
    xr = _rand(g, B * rfn, C); res_ = _rand(g, B * rfn, C)
    ob2 = torch.empty((B * rfn, C), device="cuda"); rob2 = torch.empty((B * rfn, C), dtype=torch.float64)
    ops.affine_act_add(xr.cuda(), ob2, scb.cuda(), shb.cuda(), relu=True, residual=res_.cuda(), rows_per_group=rfn)
    ref_ops.affine_act_add(_d(xr), rob2, _d(scb), _d(shb), relu=True, residual=_d(res_), rows_per_group=rfn)
    _check(ob2, rob2, 1e-6, "affine_act_add batch")


@pytest.mark.parametrize("B,K,O,act", [(1, 32768, 512, 2), (1, 512, 7, 0), (3, 1024, 40, 1),
                                       # g6d_linear_gemv_batch's 8-rows x K-slice kernel (B >= 2, K % 4096 == 0, O % 8 == 0): the refiner's
                                       # first FC layer with 2 / 8 queries, a ragged group (11 = 8 + 3), configs[4]'s 32, and shapes that
                                       # fall back to the row-per-block kernel (O % 8 != 0; one K slice)
                                       (2, 32768, 512, 2), (8, 32768, 512, 2), (11, 8192, 64, 1), (32, 32768, 512, 0), (5, 8192, 20, 0),
                                       (4, 4096, 16, 2),
                                       # round 5: up to 32 right-hand sides per weight pass (2 / 4 groups of 8 in one block): the bench's 16,
                                       # ragged 20 = 8 + 8 + 4, two launches for 40 = 32 + 8, 19 rows on the fall-back kernels
                                       (16, 32768, 512, 2), (20, 8192, 64, 1), (40, 8192, 64, 0), (19, 8192, 20, 0)])
@pytest.mark.parametrize("mfma", [2, 1, 0], ids=["matrix-cores", "product-rule", "vector-alu"])   # knob gemv_mfma (round 5): 2..32 right-hand sides as a GEMM
def test_linear_gemv(ops, B, K, O, act, mfma, knob):
    knob("gemv_mfma", mfma)
    g = torch.Generator().manual_seed(12)
    x, W, b = _rand(g, B, K), _rand(g, O, K, scale=K ** -0.5), _rand(g, O, scale=0.1)
    out = ops.linear_gemv(x.cuda(), W.cuda(), b.cuda(), act)
    _check(out, ref_ops.linear_gemv(_d(x), _d(W), _d(b), act), 1e-5)
    out2 = ops.linear_gemv(x.cuda(), W.cuda(), b.cuda(), act)      # slices are added in slice order: bit-equal from run to run, counters re-armed
    assert torch.equal(out, out2)


@pytest.mark.parametrize("out_float", [False, True])
def test_warp_perspective(ops, out_float):
    """g6d_warp_perspective vs a float bilinear inverse warp (identity, integer shift, affine 2x3, homography)."""
    from gen6d_amd import synth
    img = torch.from_numpy(synth.synth_images(1, 96, 128, 3)[0])
    ident = ops.warp_perspective(img.cuda(), np.eye(3), 96, 128)
    assert torch.equal(ident.cpu(), img)
    shift = ops.warp_perspective(img.cuda(), np.array([[1.0, 0, 5], [0, 1.0, -3]]), 96, 128).cpu()
    assert torch.equal(shift[:93, 5:], img[3:, :123]) and (shift[93:] == 0).all() and (shift[:, :5] == 0).all()
    H = np.array([[0.9, -0.2, 12.0], [0.15, 1.1, -7.0], [2e-4, -1e-4, 1.0]])
    got = ops.warp_perspective(img.cuda(), H, 64, 64, out_float=out_float).cpu()
    ref = ref_ops.warp_perspective(img, H, 64, 64, out_float=out_float)
    if out_float:
        assert (got - ref).abs().max() < 2e-3
    else:
        assert (got.int() - ref.int()).abs().max() <= 1


@pytest.mark.parametrize("shape,relu,pool", [((2, 64, 16, 24), True, True), ((1, 512, 22, 30), False, True),
                                             ((1, 512, 22, 30), False, False), ((3, 128, 8, 8), True, False),
                                             ((1, 64, 6, 10), True, True), ((2, 32, 15, 15), True, True), ((2, 32, 7, 7), False, True)])
def test_bias_relu_pool_nchw(ops, shape, relu, pool):
    g = torch.Generator().manual_seed(13)
    x, b = _rand(g, *shape), _rand(g, shape[1])
    out = ops.bias_relu_pool_nchw(x.cuda(), b.cuda(), relu, pool)
    _check(out, ref_ops.bias_relu_pool_nchw(_d(x), _d(b), relu, pool), 1e-6)


@pytest.mark.parametrize("H,W,Cin,Cout,k", [(22, 30, 512, 32, 15), (8, 40, 64, 8, 15), (17, 33, 96, 32, 7), (5, 5, 36, 3, 3),
                                             (44, 60, 512, 32, 15)])
def test_corr2d_patch(ops, H, W, Cin, Cout, k):
    g = torch.Generator().manual_seed(21)
    x = _rand(g, 1, 1, H, W, Cin)
    w = _rand(g, Cout, k * k, Cin, scale=(1.0 / (k * k * Cin)) ** 0.5)
    obuf = torch.full((1, 1, H, W, Cout + 4), -5.0, device="cuda")
    out = obuf[..., :Cout]
    ops.corr2d_patch(x.cuda(), w.cuda(), out, k)
    ref = torch.empty((1, 1, H, W, Cout), dtype=torch.float64)
    ref_ops.corr2d_patch(_d(x), _d(w), ref, k)
    _check(out, ref, 2e-5, "corr2d_patch")
    assert (obuf[..., Cout:] == -5.0).all()


# ---------------------------------------------------------------------------------------------------- own VGG trunk
WINO_CASES = [
    # N, H, W, Cin, Cout, relu, full, pool
    (1, 16, 16, 64, 128, True, True, True),        # one block, all four quarters full
    (1, 15, 15, 512, 512, True, True, False),      # odd map (detector reference features), quarters partly masked
    (5, 7, 7, 512, 512, False, True, True),        # <= 8x8 maps: four images per block, N % 4 != 0, c7_pre / p7 (no ReLU)
    (7, 8, 8, 512, 512, True, True, False),        # refiner 1/16 level
    (2, 44, 58, 256, 256, True, True, True),       # tap + pooled output, blocks with masked quarters (58 = 7.25 quarters)
    (1, 30, 40, 128, 256, True, False, True),      # pooled output only
    (1, 64, 64, 64, 128, True, False, True),       # first Winograd layer of a 128x128 crop
    (3, 10, 18, 8, 64, True, True, True),          # smallest channel counts the kernel accepts
    (8, 64, 64, 64, 128, True, True, True),        # 256 blocks: the un-split path (smaller grids above split the channel chunks)
    (12, 16, 16, 256, 512, True, False, True),     # 96 blocks x 2 splits
]


@pytest.mark.parametrize("shape", ["rule", "wide", "square"])      # block shape: by the launcher's rule / 2 quarters x 128 channels / 4 x 64
@pytest.mark.parametrize("N,H,W,Cin,Cout,relu,full,pool", WINO_CASES)
def test_wino_conv3x3(ops, N, H, W, Cin, Cout, relu, full, pool, shape, knob):
    if shape != "rule":
        if Cout % 128:
            pytest.skip("one block shape only")
        knob("wino_wide", 2 if shape == "wide" else 0)
    """g6d_wino_conv3x3 against F.conv2d (+bias, ReLU, max-pool) in float64: fp32 Winograd F(2x2,3x3) error class."""
    import torch.nn.functional as F
    from gen6d_amd.network.backbone import winograd_filters
    g = torch.Generator().manual_seed(1000 + H * W + Cin)
    x = _rand(g, N, H, W, Cin)
    w = _rand(g, Cout, Cin, 3, 3, scale=(2.0 / (9 * Cin)) ** 0.5 * 1.7)
    b = _rand(g, Cout, scale=0.3)
    y, yp = ops.wino_conv3x3(x.cuda(), winograd_filters(w).cuda(), b.cuda(), relu=relu, full=full, pool=pool)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1)
    if relu:
        ref = F.relu(ref)
    assert (y is not None) == full and (yp is not None) == pool
    if full:
        _check(y.permute(0, 3, 1, 2), ref, 3e-5, "wino full")
    if pool:
        assert yp.shape == (N, H // 2, W // 2, Cout)
        _check(yp.permute(0, 3, 1, 2), F.max_pool2d(ref, 2, 2), 3e-5, "wino pool")


def test_wino_conv3x3_channel_slice_input(ops):
    """Input given as a channel slice of a wider channels-last buffer (ld_in > Cin)."""
    import torch.nn.functional as F
    from gen6d_amd.network.backbone import winograd_filters
    g = torch.Generator().manual_seed(77)
    wide = _rand(g, 2, 12, 12, 96)
    w = _rand(g, 64, 64, 3, 3, scale=0.06)
    b = _rand(g, 64, scale=0.3)
    xs = wide.cuda()[..., 16:80]
    y, _ = ops.wino_conv3x3(xs, winograd_filters(w).cuda(), b.cuda(), relu=False)
    ref = F.conv2d(wide[..., 16:80].double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1)
    _check(y.permute(0, 3, 1, 2), ref, 3e-5, "wino slice")


@pytest.mark.parametrize("k,sizes,Cin,Cout", [(15, [(88, 116), (60, 80), (44, 60), (32, 40)], 512, 32), (7, [(22, 29), (15, 20), (11, 15), (8, 10)], 512, 32),
                                             (7, [(9, 33), (8, 8)], 100, 5)])
def test_corr2d_patch_multi(ops, k, sizes, Cin, Cout):
    """One launch over several maps (flat tile list across maps, common split) equals the per-map launches and the fp64 reference."""
    g = torch.Generator().manual_seed(600 + k + Cin)
    w = _rand(g, Cout, k * k, Cin, scale=(1.0 / (k * k * Cin)) ** 0.5)
    xs_cpu = [_rand(g, 1, 1, h, ww, Cin) for h, ww in sizes]
    dev = torch.device("cuda")
    xs = ops.alloc_like_segments([tuple(x.shape) for x in xs_cpu], dev)
    for d_, x in zip(xs, xs_cpu):
        d_.copy_(x)
    outs = ops.alloc_like_segments([(1, 1, h, ww, Cout) for h, ww in sizes], dev)
    for rep in range(2):                                   # twice: the split counters must be left re-armed
        for o in outs:
            o.fill_(-3.0)
        ops.corr2d_patch_multi(xs, w.cuda(), outs, k)
        for x, o, xc in zip(xs, outs, xs_cpu):
            ref = torch.empty(tuple(o.shape), dtype=torch.float64)
            ref_ops.corr2d_patch(_d(xc), _d(w), ref, k)
            _check(o, ref, 2e-5, "corr2d multi")
            single = torch.empty_like(o)
            ops.corr2d_patch(x, w.cuda(), single, k)
            assert (o - single).abs().max().item() <= 1e-5 * max(1.0, single.abs().max().item())


def test_corr2d_patch_multi_query_batch(ops):
    """Three queries per scale in one launch (G6dCorrSeg.N), incl. the 3x3 level that joined the multi launch in round 3."""
    for k, sizes in ((7, [(22, 29), (15, 20), (11, 15), (8, 10)]), (3, [(11, 15), (8, 10), (6, 8), (4, 5)]), (15, [(24, 40), (16, 20)])):
        g = torch.Generator().manual_seed(700 + k)
        Cin, Cout, N = 512, 32, 3
        w = _rand(g, Cout, k * k, Cin, scale=(1.0 / (k * k * Cin)) ** 0.5)
        xs_cpu = [_rand(g, N, 1, h, ww, Cin) for h, ww in sizes]
        dev = torch.device("cuda")
        xs = ops.alloc_like_segments([tuple(x.shape) for x in xs_cpu], dev)
        for d_, x in zip(xs, xs_cpu):
            d_.copy_(x)
        outs = ops.alloc_like_segments([(N, 1, h, ww, Cout) for h, ww in sizes], dev)
        for rep in range(2):
            for o in outs:
                o.fill_(-3.0)
            ops.corr2d_patch_multi(xs, w.cuda(), outs, k)
            for o, xc in zip(outs, xs_cpu):
                ref = torch.empty(tuple(o.shape), dtype=torch.float64)
                ref_ops.corr2d_patch(_d(xc), _d(w), ref, k)
                _check(o, ref, 2e-5, f"corr2d multi batch k={k}")


@pytest.mark.parametrize("N,sizes,Cin,Cout", [(1, [(88, 116), (60, 80), (44, 60), (32, 40)], 512, 32), (3, [(22, 30), (9, 13)], 64, 32),
                                              (2, [(16, 16)], 128, 64)])
def test_corr2d_wino_multi(ops, N, sizes, Cin, Cout):
    """15x15 correlation as 5x5 blocks of 3x3 sub-filters accumulated in the Winograd domain (wino_conv3x3_kernel<0,25,*>) against
    the fp64 direct correlation and the direct-form corr_patch kernel; twice, so the split counters are left re-armed."""
    from gen6d_amd.network.backbone import winograd_corr_filters
    g = torch.Generator().manual_seed(900 + Cin)
    k = 15
    w = _rand(g, Cout, k * k, Cin, scale=(1.0 / (k * k * Cin)) ** 0.5)
    U = winograd_corr_filters(w, k).cuda()
    xs_cpu = [_rand(g, N, 1, h, ww, Cin) for h, ww in sizes]
    dev = torch.device("cuda")
    xs = ops.alloc_like_segments([tuple(x.shape) for x in xs_cpu], dev)
    for d_, x in zip(xs, xs_cpu):
        d_.copy_(x)
    outs = ops.alloc_like_segments([(N, 1, h, ww, Cout) for h, ww in sizes], dev)
    for rep in range(2):
        for o in outs:
            o.fill_(-3.0)
        ops.corr2d_wino_multi(xs, U, outs, 5)
        for o, xc in zip(outs, xs_cpu):
            ref = torch.empty(tuple(o.shape), dtype=torch.float64)
            ref_ops.corr2d_patch(_d(xc), _d(w), ref, k)
            _check(o, ref, 4e-5, "corr2d wino multi")
    if Cout <= 32:
        direct = ops.alloc_like_segments([(N, 1, h, ww, Cout) for h, ww in sizes], dev)
        ops.corr2d_patch_multi(xs, w.cuda(), direct, k)
        for o, d_ in zip(outs, direct):
            assert (o - d_).abs().max().item() <= 4e-5 * max(1.0, d_.abs().max().item())


WINO_MULTI_CASES = [
    # segment sizes (N, H, W), Cin, Cout, relu, full, pool
    ([(1, 44, 58), (1, 30, 40), (1, 22, 30), (1, 16, 20)], 512, 512, False, True, True),     # detector pyramid, 1/16 level, c7_pre + p7
    ([(1, 22, 30), (1, 16, 20)], 64, 128, True, False, True),                                # pooled output only, blocks straddle segments
    ([(2, 9, 7), (1, 8, 8), (3, 5, 13)], 128, 64, True, True, False),                        # ragged sizes, several images per segment
    ([(1, 88, 116), (1, 60, 80), (1, 44, 60), (1, 32, 40)], 256, 512, True, True, False),    # 1/8 level: un-split
]


@pytest.mark.parametrize("shape", ["rule", "wide", "square"])
@pytest.mark.parametrize("sizes,Cin,Cout,relu,full,pool", WINO_MULTI_CASES)
def test_wino_conv3x3_multi(ops, sizes, Cin, Cout, relu, full, pool, shape, knob):
    if shape != "rule":
        if Cout % 128:
            pytest.skip("one block shape only")
        knob("wino_wide", 2 if shape == "wide" else 0)
    """One launch over several map sizes (flat quarter list across segments) against F.conv2d per segment in float64."""
    import torch.nn.functional as F
    from gen6d_amd.network.backbone import winograd_filters
    g = torch.Generator().manual_seed(4242 + Cin + len(sizes))
    w = _rand(g, Cout, Cin, 3, 3, scale=(2.0 / (9 * Cin)) ** 0.5 * 1.7)
    b = _rand(g, Cout, scale=0.3)
    xs_cpu = [_rand(g, n, h, ww, Cin) for n, h, ww in sizes]
    xs = ops.alloc_like_segments([tuple(x.shape) for x in xs_cpu], torch.device("cuda"))
    for d, x in zip(xs, xs_cpu):
        d.copy_(x)
    for rep in range(2):                                   # twice: the split counters must be left re-armed
        ys, yps = ops.wino_conv3x3_multi(xs, winograd_filters(w).cuda(), b.cuda(), relu=relu, full=full, pool=pool)
        assert (ys is not None) == full and (yps is not None) == pool
        for i, x in enumerate(xs_cpu):
            ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1)
            if relu:
                ref = F.relu(ref)
            if full:
                _check(ys[i].permute(0, 3, 1, 2), ref, 3e-5, f"wino multi full seg {i}")
            if pool:
                _check(yps[i].permute(0, 3, 1, 2), F.max_pool2d(ref, 2, 2), 3e-5, f"wino multi pool seg {i}")


def test_trunk_multi_equals_per_scale(ops):
    """The pyramid trunk (one launch per layer over all scales) gives the per-scale trunk's features."""
    from gen6d_amd import synth
    from gen6d_amd.network import name2network, backbone as B
    from gen6d_amd.network.params import fold_vgg
    net = name2network["detector"]({"name": "t"}).eval()
    net.load_state_dict(synth.synth_state_dict("detector"))
    packed = B.pack_trunk(fold_vgg(net.cuda(), "backbone.features"))
    g = torch.Generator().manual_seed(9)
    imgs = [torch.rand((1, 3, h, w), generator=g).cuda() for h, w in [(160, 224), (96, 128), (64, 96), (32, 64)]]
    keys = ("c5", "c7_pre", "p7")
    multi = B.trunk_features_multi(packed, imgs, keys)
    for im, fm in zip(imgs, multi):
        single = B.trunk_features(packed, im, keys, False)
        for a, b_ in zip(fm, single):
            assert a.shape == b_.shape
            assert (a - b_).abs().max().item() <= 1e-5 * max(1.0, b_.abs().max().item())


def test_wino_rejects_bad_args(ops):
    x = torch.zeros((1, 8, 8, 12), device="cuda")
    with pytest.raises((RuntimeError, ValueError)):
        ops.wino_conv3x3(x, torch.zeros((1, 16, 64, 8), device="cuda"), torch.zeros(64, device="cuda"))


@pytest.mark.parametrize("N,H,W", [(1, 128, 128), (2, 50, 70), (1, 33, 67), (3, 2, 2), (1, 352, 464)])
def test_vgg_conv1_pool_nhwc(ops, N, H, W):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    x, w, b = _rand(g, N, 3, H, W), _rand(g, 64, 3, 3, 3, scale=0.3), _rand(g, 64, scale=0.2)
    out = ops.vgg_conv1_pool_nhwc(x.cuda(), w.cuda(), b.cuda())
    ref = F.max_pool2d(F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)), 2, 2)
    assert out.shape == (N, H // 2, W // 2, 64)
    _check(out.permute(0, 3, 1, 2), ref, 1e-5, "conv1 nhwc")
    # the variant that normalises the image while staging it (the trunk's entry), into a caller-provided destination
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    xi = torch.rand((N, 3, H, W), generator=g)
    dst = torch.full((N, H // 2, W // 2, 64), -5.0, device="cuda")
    ops.vgg_conv1_pool_nhwc(xi.cuda(), w.cuda(), b.cuda(), out=dst, norm=(mean, std))
    xn = (xi.double() - torch.tensor(mean, dtype=torch.float64).view(1, 3, 1, 1)) / torch.tensor(std, dtype=torch.float64).view(1, 3, 1, 1)
    refn = F.max_pool2d(F.relu(F.conv2d(xn, w.double(), b.double(), padding=1)), 2, 2)
    _check(dst.permute(0, 3, 1, 2), refn, 1e-5, "conv1 nhwc norm")


def test_l2norm_rows(ops):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(6)
    x = _rand(g, 3, 5, 7, 512)
    x[0, 0, 0] = 0.0                                   # zero row: eps branch of F.normalize
    got = ops.l2norm_rows(x.cuda().clone())
    _check(got, F.normalize(x.double(), dim=-1), 1e-6, "l2norm_rows")
    # row-strided view with rows that are not 16-byte aligned: the quaternions of a batch of regressor outputs [qn,7]
    o = _rand(g, 5, 7).cuda()
    want = torch.cat([F.normalize(o[:, :4].double().cpu(), dim=1), o[:, 4:].double().cpu()], 1)
    ops.l2norm_rows(o[:, 0:4])
    _check(o, want, 1e-6, "l2norm_rows strided")


def test_own_trunk_matches_library_trunk(ops):
    """The product trunk (own channels-last Winograd kernels) against the MIOpen NCHW trunk of tools/library_trunk.py: same taps."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import library_trunk as LT
    from gen6d_amd import synth
    from gen6d_amd.network import backbone as B
    from gen6d_amd.network.params import fold_vgg
    from gen6d_amd.network import name2network
    net = name2network["detector"]({"name": "t"}).eval()
    net.load_state_dict(synth.synth_state_dict("detector"))
    net = net.cuda()
    folded = fold_vgg(net, "backbone.features")
    img = synth.imgs_to_tensor(synth.synth_images(2, 96, 160, 9)).cuda()
    with torch.no_grad():
        x = LT.img_norm(img)
        lib_t = LT.vgg_taps(folded, x, {"c3", "c5", "c7_pre", "p7"})
        own_t = B.vgg_taps_cl([folded[0]] + [(B.winograd_filters(w), b) for w, b in folded[1:]], x, {"c3", "c5", "c7_pre", "p7"})
    for k in ("c3", "c5", "c7_pre", "p7"):
        _check(own_t[k].permute(0, 3, 1, 2), lib_t[k].double().cpu(), 5e-5, k)


@pytest.mark.parametrize("N,H,W", [(2, 480, 640), (1, 128, 128), (3, 45, 77)])
def test_resize_bilinear_pyramid(ops, N, H, W):
    """The detector's image pyramid in one launch (reference network/detector.py:236-241: F.interpolate(..., mode='bilinear') per
    detection scale, align_corners False) against torch's own interpolation in fp64 on the CPU; the scale of the image's own size is the
    image itself (no copy)."""
    from gen6d_amd.network.detector import Detector
    g = torch.Generator().manual_seed(3)
    x = torch.rand((N, 3, H, W), generator=g)
    sizes = [Detector._scale_size(H, W, s) for s in (0.5, 0.0, -0.5, -1.0)]
    if (H, W) == (45, 77):
        sizes[1] = (H, W)                         # (odd sizes: the identity scale is not rounded up to a multiple of 32 here)
        sizes[2] = (23, 39)                       # a width that is not a multiple of 4: the one-output-per-thread path
    xg = x.cuda()
    outs = ops.resize_bilinear_pyramid(xg, sizes)
    torch.cuda.synchronize()
    for o, sz in zip(outs, sizes):
        assert tuple(o.shape) == (N, 3) + tuple(sz)
        if tuple(sz) == (H, W):
            assert o.data_ptr() == xg.data_ptr()
            continue
        # the source coordinate is formed in fp32, as ATen's own device kernel forms it (ulp 3e-5 at coordinate 480: the weights of a
        # random image's neighbours move by that much against an fp64 evaluation): fp64 at 1e-4, ATen's device kernel at 2e-6
        ref = torch.nn.functional.interpolate(x.double(), size=tuple(sz), mode="bilinear")
        _check(o, ref, 1e-4, f"pyramid {sz} vs fp64")
        _check(o, torch.nn.functional.interpolate(xg, size=tuple(sz), mode="bilinear").cpu(), 2e-6, f"pyramid {sz} vs ATen on the device")


@pytest.mark.parametrize("n,heads,C,B", [(64, 8, 512, 5), (37, 8, 512, 2), (100, 8, 512, 2), (20, 4, 512, 1), (64, 8, 256, 3)])
def test_attention_kernels(ops, n, heads, C, B):
    """AttentionBlock's attention (reference network/attention.py:4-17, head split c -> (d = c // heads, head = c % heads)) on both
    kernels: one block per (query, head) with Q / K / V of the head in LDS for n <= 64 tokens and head width <= 64 (round 5), one wave
    per (head, token) otherwise (n = 100; 4 heads of 128)."""
    g = torch.Generator().manual_seed(77 + n)
    qkv = _rand(g, B * n, 3 * C, scale=1.5).cuda()
    out = torch.empty((B * n, C), device="cuda"); ref = torch.empty((B * n, C), dtype=torch.float64)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], heads, out, batch=B)
    d = _d(qkv)
    ref_ops.attention(d[:, :C], d[:, C:2 * C], d[:, 2 * C:], heads, ref, batch=B)
    _check(out, ref, 1e-5, "attention")
