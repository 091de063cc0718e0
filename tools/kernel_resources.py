#!/usr/bin/env python3
"""Registers / LDS / spills of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py fused_kernels.hip [substring ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "torch-tts_amd", "csrc", sys.argv[1])
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]
r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"],
                   capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
for b in blocks:
    name = b.split("\n")[0]
    g = lambda pat: re.search(pat, b).group(1)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("ttsdec::", "")
    dn = re.sub(r"\(.*", "", dn)
    if len(sys.argv) > 2 and not all(f in dn for f in sys.argv[2:]):
        continue
    pats = {"vgpr": r"VGPRs: (\d+)", "agpr": r"AGPRs: (\d+)", "sgpr": r"TotalSGPRs: (\d+)", "spill": r"VGPRs Spill: (\d+)",
            "scratch": r"ScratchSize \[bytes/lane\]: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)", "occ": r"Occupancy \[waves/SIMD\]: (\d+)"}
    print(f"{dn[:120]:120s} " + " ".join(f"{k}={g(v)}" for k, v in pats.items()))
