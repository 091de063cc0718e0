#!/usr/bin/env python3
"""Runs each decode-step kernel `iters` times on LJSpeech dims (through
ttsdec_profile_step) so rocprofv3 --kernel-trace / --pmc can be pointed at a short,
steady workload.  Usage: python tools/prof_kernels.py [--batch 256] [--iters 20]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_tts_amd as T  # noqa: E402
from torch_tts_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--mem-len", type=int, default=120)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--precision", default="f32", choices=["f32", "split_f16"])
ap.add_argument("--dropout", default="philox", choices=["philox", "off"])
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(42)
cell = T.Taco2ProdDecoderCell(512, 80, 1, [1024, 1024], dim_pre=256, dim_att=1024)
dec = T.Decoder(cell, 1, 80)
for m in dec.modules():
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.xavier_normal_(m.weight, gain=1.5)
dec = dec.to(dev).eval()
dec.precision = args.precision
eng = dec.engine(dev)
mem = torch.tanh(torch.randn(args.batch, args.mem_len, 512, device=dev) * 0.5)
ms = eng.profile_step(mem, iters=args.iters, dropout_mode=_lib.DROPOUT_PHILOX if args.dropout == "philox" else _lib.DROPOUT_OFF,
                      masks=None, seed=1)
torch.cuda.synchronize()
print(json.dumps({k: round(v * 1e3, 2) for k, v in ms.items()}))
