#!/usr/bin/env python
"""Instruction-class counts of every wino43_kernel instantiation in the product build's ISA (hipcc -S with the Makefile's flags):
packed / scalar fp32 vector-ALU instructions, accumulator reads, MFMAs, global stores, scratch traffic.
Usage: python tools/w43_isa.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gen6d_amd", "csrc", "wino43_conv.hip")
out = "/tmp/w43_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-fno-slp-vectorize", "-S",
                "--cuda-device-only", src, "-o", out] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", text, re.M)]
for i, (pos, name) in enumerate(starts):
    if "wino43_kernel" not in name:
        continue
    end = text.find(".end_amdhsa_kernel", pos)
    body = text[pos:end]
    code = body[:body.find("s_endpgm")]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", dn)
    c = lambda pat: len(re.findall(pat, code))
    scr = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
    print(f"{dn:28s} instr {c(chr(10)):6d} pk_fma {c(r'v_pk_fma_f32'):4d} pk_mul {c(r'v_pk_mul_f32'):4d} pk_add {c(r'v_pk_add_f32'):4d} "
          f"fma {c(r'v_fma(c|ak|mk)?_f32'):4d} mul {c(r'v_mul_f32'):4d} add/sub {c(r'v_(add|sub|subrev)_f32'):4d} max {c(r'v_max_f32'):4d} "
          f"accread {c(r'v_accvgpr_read'):4d} accwrite {c(r'v_accvgpr_write'):4d} mov {c(r'v_mov_b32'):4d} mfma {c(r'v_mfma'):4d} "
          f"gstore {c(r'global_store_dword '):4d} gstore4 {c(r'global_store_dwordx4'):4d} scratch ld/st {c(r'scratch_load'):3d}/{c(r'scratch_store'):3d} "
          f"scratch bytes {scr.group(1) if scr else '?'}")
