#!/usr/bin/env python
"""Register / scratch / occupancy summary of every kernel in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py gen6d_amd/csrc/wino_conv.hip [...]"""
import re
import subprocess
import sys

for src in sys.argv[1:]:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-c", src, "-o",
                          "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()}
            continue
        for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m:
                cur[key] = int(m.group(1))
                if key.startswith("LDS"):
                    name = re.sub(r"\(anonymous namespace\)::", "", cur["name"]).split("(")[0]
                    print(f"{name:60s} vgpr {cur.get('VGPRs')} agpr {cur.get('AGPRs')} scratch {cur.get('ScratchSize [bytes/lane]')} "
                          f"occ {cur.get('Occupancy [waves/SIMD]')}")
