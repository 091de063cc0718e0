#!/usr/bin/env python3
"""Per-launch times of the decode step (ttsdec_profile_step: every launch on its own, no hand-off waits) under the GEMM core's
measurement ablations (option profile_ablation -> the kernels' dbg switch): 0 = none, 1 = every tile load reads the 16-byte
zero block (same instruction stream, no memory traffic), 3 = no LDS-DMA inside the K loop, 4 = no MFMAs.  LJSpeech dims."""
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
ap.add_argument("--precision", default="split_f16")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(42)
cell = T.Taco2ProdDecoderCell(512, 80, 1, [1024, 1024], dim_pre=256, dim_att=1024)
dec = T.Decoder(cell, 1, 80).to(dev).eval()
dec.precision = args.precision
eng = dec.engine(dev)
mem = torch.tanh(torch.randn(args.batch, 120, 512, device=dev) * 0.5)
out = {}
for abl in (0, 1, 3, 4, 5, 6):
    eng.set_option("profile_ablation", abl)
    ms = eng.profile_step(mem, iters=50, dropout_mode=_lib.DROPOUT_PHILOX, masks=None, seed=1)
    out[abl] = {k: round(v * 1e3, 2) for k, v in ms.items()}
print(json.dumps(out, indent=1))
