#!/usr/bin/env python
"""List what the compiler placed in every MFMA gap of the Winograd kernel's chunk loop (instantiation <0,1,2>): the loop is
64 MFMAs; everything else has to hide in the 64-cycle gaps between them.  Usage: python tools/wino_gaps.py [MODE KD NWN]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode, kd, nwn = (sys.argv[1:4] + ["0", "1", "2"])[:3] if len(sys.argv) >= 4 else ("0", "1", "2")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-fno-slp-vectorize", "-S", "-o",
                "/tmp/wino.s", os.path.join(ROOT, "gen6d_amd/csrc/wino_conv.hip"), "--cuda-device-only"], check=True, capture_output=True)
s = open("/tmp/wino.s").read()
i = s.index(f"_ZN12_GLOBAL__N_119wino_conv3x3_kernelILi{mode}ELi{kd}ELi{nwn}EEEvNS_8WinoArgsE:")
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
body = [x.strip() for x in k[a:b] if x.strip() and not x.strip().startswith((";", "."))]
print(len(body), "instructions,", nm, "MFMAs")
gap, out = [], []
for x in body:
    op = x.split()[0]
    if op.startswith("v_mfma"):
        out.append(gap); gap = []
        continue
    short = op.replace("global_load_lds_dwordx4", "GLDS").replace("global_load_dwordx4", "GLOAD").replace("ds_read_b128", "dsr").replace("ds_write_b128", "dsw")
    if op == "s_waitcnt":
        short = "WAIT(" + x.split(None, 1)[1] + ")"
    if op == "s_nop":
        short = "nop" + x.split()[1]
    gap.append(short)
out.append(gap)
for n, g in enumerate(out):
    print(n, len(g), " ".join(g))
