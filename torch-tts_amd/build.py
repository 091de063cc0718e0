"""Builds libttsdec.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python torch-tts_amd/build.py [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so lands in torch-tts_amd/lib/ (git-ignored,
but it travels with the repo snapshot to the GPU box)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libttsdec.so")
SOURCES = ["decode_kernels.hip", "frame_kernel.hip", "fused_kernels.hip", "api.hip", "encoder.hip", "vits2.hip", "conv256.hip"]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",  # epilogue math mirrors the reference's separately rounded ops
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _digest() -> str:
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for fn in sorted(os.listdir(root)):
            if fn.endswith((".hip", ".h")):
                with open(os.path.join(root, fn), "rb") as f:
                    h.update(fn.encode())
                    h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libttsdec.digest")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs = [os.path.join(LIBDIR, src.replace(".hip", ".o")) for src in SOURCES]

    def compile_one(src, obj):
        # (-Rpass-analysis: the compiler's per-kernel register / LDS / scratch report, kept beside the object - tests/ checks that
        # no kernel of the library touches scratch memory: a spill inside a K loop once cost the split-fp16 step 45 %)
        cmd = [hipcc, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        with open(obj[:-2] + ".resources.txt", "w") as f:
            f.write("\n".join(l for l in r.stderr.splitlines() if "remark:" in l))
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, cmd)

    # the translation units are independent: one hipcc per source, side by side (each is itself single-threaded)
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        for f in [pool.submit(compile_one, s, o) for s, o in zip(SOURCES, objs)]:
            f.result()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
