"""The halo-patch kernel (conv16w_kernel) receives its filter fragments through hand-issued asynchronous loads and counted waits: a
register spill inside it would (a) copy a destination register before its data has arrived and (b) add vector-memory requests the
counted waits do not know about.  The build must therefore keep every instantiation free of spills — checked on the compiler's own
metadata (cross-compiles without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_conv16w_has_no_spills(tmp_path):
    out = tmp_path / "conv16.s"
    src = os.path.join(ROOT, "gen6d_amd", "csrc", "conv16_direct.hip")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-w", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                        "-o", str(out), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S*conv16w_kernel\S*)\n(.*?)\.wavefront_size", text, re.S)
    assert len(kernels) >= 5, "conv16w_kernel instantiations not found in the metadata"
    for name, body in kernels:
        vs = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", body).group(1))
        ss = int(re.search(r"\.sgpr_spill_count:\s+(\d+)", body).group(1))
        vg = int(re.search(r"\.vgpr_count:\s+(\d+)", body).group(1))
        priv = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
        assert vs == 0 and priv == 0, f"{name}: {vs} spilled vector registers, {priv} B of scratch"
        assert ss <= 96, f"{name}: {ss} spilled scalar registers (they go to lanes of a vector register: no memory traffic, but keep them few)"
        assert vg <= 256, f"{name}: {vg} vector registers (two blocks per CU need <= 256)"
    shutil.rmtree(tmp_path, ignore_errors=True)
