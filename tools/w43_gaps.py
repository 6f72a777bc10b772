#!/usr/bin/env python
"""List what the compiler placed between the MFMAs of the F(4x4,3x3) kernel's chunk loop (wino43_conv.hip).
Usage: python tools/w43_gaps.py [MODE KD NT] [-v]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("-")]
mode, kd, nt = (args + ["0", "1", "4"])[:3] if len(args) >= 3 else ("0", "1", "4")
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-fno-slp-vectorize", "-S", "-o",
                "/tmp/w43.s", os.path.join(ROOT, "gen6d_amd/csrc/wino43_conv.hip"), "--cuda-device-only"] + extra, check=True, capture_output=True)
s = open("/tmp/w43.s").read()
i = s.index(f"_ZN12_GLOBAL__N_113wino43_kernelILi{mode}ELi{kd}ELi{nt}EEEvNS_7W43ArgsE:")
k = s[i:s.index("s_endpgm", i)].splitlines()
labels = {m.group(1): n for n, l in enumerate(k) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
best = None
for n, l in enumerate(k):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        nm = sum("v_mfma" in x for x in k[labels[m.group(1)]:n])
        if best is None or nm > best[0]:
            best = (nm, labels[m.group(1)], n)
nm, a, b = best
body = [x.strip() for x in k[a:b] if x.strip() and not x.strip().startswith((";"))]
print(len(body), "lines,", nm, "MFMAs")
if "-v" in sys.argv:
    print("\n".join(body)); sys.exit()
gap, out = [], []
for x in body:
    op = x.split()[0]
    if op.startswith("v_mfma"):
        out.append(gap); gap = []
        continue
    short = (op.replace("global_load_lds_dwordx4", "GLDS").replace("buffer_load_dwordx4", "BLOAD").replace("ds_read2st64_b64", "dsr2st").replace("ds_read2_b64", "dsr2")
             .replace("ds_read_b64", "dsr").replace("ds_write_b128", "dsw").replace("v_pk_fma_f32", "pkfma"))
    if op == "s_waitcnt":
        short = "WAIT(" + x.split(None, 1)[1] + ")"
    if op == "s_nop":
        short = "nop" + x.split()[1]
    gap.append(short)
out.append(gap)
for n, g in enumerate(out):
    print(n, len(g), " ".join(g))
