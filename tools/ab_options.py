#!/usr/bin/env python3
"""Same-process A/B of the decoder's tuning options (include/ttsdec.h TTSDEC_OPT_*) on LJSpeech dims.

    python tools/ab_options.py --batch 256 --variants "base:" "gate:lstm_start_gate=1" "nohead:head_proj=0"

Every variant is a set of options applied to ONE engine (ttsdec_set_option drops the captured graph); the variants are timed in
interleaved rounds (cdna_hip_programming.md rule 24) on a 600-frame Philox decode, and each variant's outputs are compared with
the first variant's (max |dy|, argmax mismatches): the options may only reorder a GEMM's K segments.  Prints one JSON line."""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_tts_amd as T  # noqa: E402
from torch_tts_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--mem-len", type=int, default=120)
ap.add_argument("--frames", type=int, default=600)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--precision", default="split_f16", choices=["f32", "split_f16"])
ap.add_argument("--variants", nargs="+", default=["base:"])
ap.add_argument("--kernels", action="store_true", help="also the per-launch times of ttsdec_profile_step per variant")
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
B, L, NF = args.batch, args.mem_len, args.frames
g = torch.Generator().manual_seed(1234)
mem = (torch.tanh(torch.randn(B, L, 512, generator=g) * 0.5)).to(dev)
y = torch.empty(B, NF, 80, device=dev)
s = torch.empty(B, NF, device=dev)
w = torch.empty(B, NF, L, device=dev)
t_out = torch.zeros(2, dtype=torch.int32, device=dev)
names = list(_lib.option_ids())


def parse(v):
    name, _, rest = v.partition(":")
    opts = {}
    for item in filter(None, rest.split(",")):
        k, _, val = item.partition("=")
        opts[k] = int(val)
    return name, opts


variants = [parse(v) for v in args.variants]


def apply(opts):
    for n in names:
        if n not in ("debug_flags", "spin_limit", "profile_ablation"):
            eng.set_option(n, opts.get(n, -1))


def decode():
    eng.decode(mem, t_begin=0, n_steps=NF, stop_threshold=-2.0, check_stop=True, dropout_mode=_lib.DROPOUT_PHILOX, masks=None, seed=123,
               teacher=None, teacher_flags=None, y=y, s=s, w=w, t_out=t_out)


ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
times = {n: [] for n, _ in variants}
ref = None
check = {}
for rnd in range(args.rounds + 1):  # round 0 = warm-up (graph capture) + the output comparison
    for name, opts in variants:
        apply(opts)
        if rnd == 0:
            decode()
        ev0.record()
        decode()
        ev1.record()
        ev1.synchronize()
        assert t_out.tolist() == [NF, 0], (name, t_out.tolist())
        if rnd == 0:
            if ref is None:
                ref = (y.clone(), w.argmax(-1).clone())
            check[name] = {"max_abs_dy": float((y - ref[0]).abs().max()), "argmax_mismatches": int((w.argmax(-1) != ref[1]).sum()),
                           "finite": bool(torch.isfinite(y).all())}
        else:
            times[name].append(ev0.elapsed_time(ev1) / NF * 1e3)  # us per step
out = {"batch": B, "precision": args.precision, "frames": NF, "rounds": args.rounds, "variants": {}}
for name, opts in variants:
    t = times[name]
    out["variants"][name] = {"options": opts, "us_per_step_median": round(statistics.median(t), 2), "us_per_step_min": round(min(t), 2),
                             "us_per_step_all": [round(v, 2) for v in t], **check[name]}
    if args.kernels:
        apply(opts)
        ms = eng.profile_step(mem, iters=50, dropout_mode=_lib.DROPOUT_PHILOX, masks=None, seed=1)
        out["variants"][name]["kernels_us_alone"] = {k: round(v * 1e3, 2) for k, v in ms.items()}
print(json.dumps(out))
