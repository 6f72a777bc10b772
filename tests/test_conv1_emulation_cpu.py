"""Index arithmetic of the fused first-layer kernel (gen6d_amd/csrc/vgg_conv1.hip) checked WITHOUT a GPU: the two phases
of the kernel are plain inline functions of (thread id, block origin); built with -DG6D_CONV1_HOST_EMU the same source
runs them thread by thread on the host.  Test infrastructure only — the product library has no host compute path."""
import ctypes as C
import os
import subprocess

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gen6d_amd", "csrc", "vgg_conv1.hip")


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("conv1_emu") / "conv1_emu.so")
    subprocess.run(["g++", "-O2", "-x", "c++", "-std=c++17", "-DG6D_CONV1_HOST_EMU", "-Wno-unknown-pragmas", "-shared", "-fPIC",
                    SRC, "-o", so], check=True)
    lib = C.CDLL(so)
    lib.g6d_conv1_emulate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p]
    lib.g6d_conv1_emulate.restype = C.c_int
    return lib


@pytest.mark.parametrize("nhwc,norm", [(0, False), (1, False), (1, True)])
@pytest.mark.parametrize("N,H,W", [(1, 16, 16), (2, 50, 70), (1, 33, 67), (1, 2, 2), (1, 18, 130)])
def test_emulated_kernel_matches_torch(emu, N, H, W, nhwc, norm):
    g = torch.Generator().manual_seed(11)
    x = torch.randn((N, 3, H, W), generator=g)
    mean, std = (C.c_float * 3)(0.485, 0.456, 0.406), (C.c_float * 3)(0.229, 0.224, 0.225)
    w = torch.randn((64, 3, 3, 3), generator=g) * 0.3
    b = torch.randn((64,), generator=g) * 0.2
    out = torch.full((N, H // 2, W // 2, 64) if nhwc else (N, 64, H // 2, W // 2), float("nan"))
    assert emu.g6d_conv1_emulate(x.data_ptr(), N, H, W, w.data_ptr(), b.data_ptr(), out.data_ptr(), nhwc,
                                 mean if norm else None, std if norm else None) == 0
    if norm:       # the padding stays zero: normalise first, then the zero-padded convolution
        x = (x - torch.tensor(list(mean)).view(1, 3, 1, 1)) / torch.tensor(list(std)).view(1, 3, 1, 1)
    if nhwc:
        out = out.permute(0, 3, 1, 2)
    ref = F.max_pool2d(F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)), 2, 2)
    assert not torch.isnan(out).any()
    err = (out.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    assert err < 1e-5, err
