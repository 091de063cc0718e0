#!/usr/bin/env python3
"""End-to-end `Tacotron.forward` latency at B = 1 (ids -> y_post), the configuration BASELINE.md section 2 quotes for the
reference on 8 CPU cores: LJSpeech config, L = 120 ids, 401 decode frames, 0.98 s -> 408 frames/s (tacotron/tacotron.py:29-56).

    python tools/e2e_latency.py [--frames 401] [--mem-len 120] [--batch 1] [--rounds 9]

Two configurations of the same drop-in model are timed, each from the host's point of view (wall clock around
`model(ids, lengths, max_steps=...)` plus a device synchronize, inputs resident on the device, weights packed by a warm-up call):
  * fast_inference()        PreNet dropout drawn on the device (Philox), split-fp16 GEMMs;
  * reference_compatible()  the defaults: the reference's CPU generator replayed for the dropout masks (drawn on the host and
                            uploaded per chunk), exact fp32 GEMMs.
Prints one JSON line."""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the shipped configs' dims)
import torch_tts_amd as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=401)
ap.add_argument("--mem-len", type=int, default=120)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--rounds", type=int, default=9)
args = ap.parse_args()

dev = torch.device("cuda:0")
torch.manual_seed(42)
model = T.build_tacotron(bench.CONFIGS["ljspeech"]).eval().to(dev)
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1, 40, (args.batch, args.mem_len), generator=g).to(dev)
lens = torch.full((args.batch,), args.mem_len, dtype=torch.long, device=dev)


def run():
    with torch.no_grad():
        y, y_post, s, extra = model(ids, lens, max_steps=args.frames)
    torch.cuda.synchronize()
    return y, y_post


def timed(setup):
    setup()
    y, y_post = run()  # packs the weights, captures the step graph, sizes the workspaces
    run()
    ts = []
    for _ in range(args.rounds):
        t0 = time.perf_counter()
        y, y_post = run()
        ts.append(time.perf_counter() - t0)
    n = y.shape[1]
    return {"frames": int(n), "ms_median": round(statistics.median(ts) * 1e3, 3), "ms_min": round(min(ts) * 1e3, 3),
            "frames_per_s": round(n * args.batch / statistics.median(ts), 1), "finite": bool(torch.isfinite(y_post).all()),
            "y_post_shape": list(y_post.shape)}


out = {
    "what": "Tacotron.forward end to end (ids -> encoder -> decoder loop -> Postnet), host wall clock incl. the final sync",
    "batch": args.batch, "mem_len": args.mem_len, "max_steps": args.frames,
    "fast_inference": timed(lambda: model.fast_inference(seed=7)),
    "reference_compatible": timed(lambda: model.reference_compatible()),
    "reference_cpu_8_cores": {"s": 0.98, "frames_per_s": 408, "source": "BASELINE.md section 2 (measured by the survey; B = 1, 401 frames)"},
}
print(json.dumps(out))
