#!/usr/bin/env python3
"""Times the VITS2 path (TextEncoder + reverse flow, ModelConfig-default sizes) on the HIP library.
Usage: python tools/time_vits2.py [--batch 64] [--tx 120] [--ty 600] [--iters 10]"""
import argparse
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore", category=FutureWarning)
import torch_tts_amd as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--tx", type=int, default=120)
ap.add_argument("--ty", type=int, default=600)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--precision", default="split_f16", choices=["split_f16", "f32"])
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
te = T.vits2.TextEncoder(178, 192, 192, 768, 2, 6, 3, 0.1).to(dev).eval()
fl = T.vits2.ResidualCouplingTransformersBlock(192, 192, 5, 1, 4, use_transformer_flows=True).to(dev).eval()
for l in fl.flows:
    if hasattr(l, "post"):
        torch.nn.init.normal_(l.post.weight, 0, 0.05)
te.precision = fl.precision = args.precision
B = args.batch
ids = torch.randint(0, 178, (B, args.tx), device=dev)
xl = torch.full((B,), args.tx, device=dev)
z = torch.randn(B, 192, args.ty, device=dev)
ym = torch.ones(B, 1, args.ty, device=dev)
with torch.no_grad():
    for name, fn in (("text_encoder", lambda: te(ids, xl)), ("flow_reverse", lambda: fl(z, ym, reverse=True))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / args.iters * 1e3:.3f} ms  (B={B}, {args.precision})")
