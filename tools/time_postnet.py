#!/usr/bin/env python3
"""Times MelPostnet (LJSpeech dims) on the HIP library in its three arithmetic modes.
Usage: python tools/time_postnet.py [--batch 256] [--frames 600] [--iters 10]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_tts_amd as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--frames", type=int, default=600)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
pn = T.MelPostnet(80, 512, 5, 3).to(dev).eval()
y = torch.randn(args.batch, args.frames, 80, device=dev)
ref = None
for mode in ("f32", "split_f16", "bf16"):
    pn.precision = mode
    with torch.no_grad():
        out = pn(y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            out = pn(y)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.iters * 1e3
    if ref is None:
        ref = out
    flops = 2 * args.batch * args.frames * (5 * (80 * 512 + 2 * 512 * 512) + 512 * 80)
    print(f"{mode:10s} {ms:7.3f} ms   {flops / ms / 1e9:7.1f} TFLOP/s (algorithmic)   max |out - f32| = {(out - ref).abs().max().item():.2e}")
