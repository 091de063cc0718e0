#!/usr/bin/env python3
"""Times Encoder2 at the bench size: HIP path vs the stock PyTorch-ROCm ops of the same module."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_tts_amd as T  # noqa: E402

torch.manual_seed(0)
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 120
enc = T.Encoder2(40, dim_out=512, dim_emb=512).cuda().eval()
ids = torch.randint(1, 40, (B, L)).cuda()
lens = torch.full((B,), L, dtype=torch.long)
with torch.no_grad():
    for use_hip in (True, False):
        enc.use_hip = use_hip
        for _ in range(2):
            out = enc(ids, lens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            out = enc(ids, lens)
        torch.cuda.synchronize()
        print(("hip  " if use_hip else "stock"), f"B={B} L={L}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per batch")
        if use_hip:
            ref = out
    print("max |hip - stock| =", float((ref - out).abs().max()))
