#!/usr/bin/env python
"""Instruction mix of the MFMA loops of a kernel: for every backward branch whose body holds MFMAs, the counts per class and the
VALU instructions per MFMA (on gfx950 every vector-ALU instruction beside an fp32 MFMA costs its issue cycles — tools/ubench).
Usage: python tools/loop_mix.py gen6d_amd/csrc/corr_patch.hip [substring of the mangled kernel name]"""
import re
import subprocess
import sys
from collections import Counter

src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "-o", "/tmp/mix.s", src,
                "--cuda-device-only"] + (["-fno-slp-vectorize"] if "wino" in src else []), check=True, capture_output=True)
s = open("/tmp/mix.s").read()
for km in re.finditer(r"^(_Z\w+):[^\n]*\n", s, re.M):
    name = km.group(1)
    if pat not in name or "kernel" not in name:
        continue
    k = s[km.end():s.index("s_endpgm", km.end())].splitlines()
    labels = {m.group(1): n for n, l in enumerate(k) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for n, l in enumerate(k):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            body = [x.strip() for x in k[labels[m.group(1)]:n] if x.strip() and not x.strip().startswith((";", "."))]
            nm = sum(x.startswith("v_mfma") for x in body)
            if nm >= 8:
                loops.append((nm, body))
    if not loops:
        continue
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
    for nm, body in sorted(loops, key=lambda t: -t[0])[:2]:
        c = Counter()
        for x in body:
            op = x.split()[0]
            cls = ("mfma" if op.startswith("v_mfma") else "ds_read" if op.startswith("ds_read") else "ds_write" if op.startswith("ds_write")
                   else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "valu" if op.startswith("v_") else "wait" if op == "s_waitcnt"
                   else "salu")
            c[cls] += 1
        print(f"{demangled[:70]:70s} loop {len(body):4d} instrs: " + " ".join(f"{k_}={v}" for k_, v in sorted(c.items())) + f"  VALU/MFMA={c['valu'] / nm:.2f}")
