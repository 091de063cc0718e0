#!/usr/bin/env python3
"""Benchmark of the Tacotron decoder hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic LJSpeech-shaped input:
600 autoregressive decode frames (max_steps=599; random-init stop logits never cross -2.0)
for B utterances per GPU with memory length L=120, followed by the Postnet over the
[B, 600, 80] mel.  Metric: mel-frames/s over the whole job (all ranks), inputs resident
in HBM.  Weak scaling: per-GPU batch fixed at 256 (BASELINE.json configs[2]/[3]:
2048 = 8 x 256).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# configs/config-ljspeech.yaml:47-69 of the reference (the file itself does not travel)
LJSPEECH = {
    "text": {"alphabet": "x" * 39},  # alphabet_size = 1 + 39 = 40 (tacotron.py:189)
    "audio": {"num_mels": 80, "sample_rate": 22050, "hop_length": 256},
    "model": {
        "encoder": {"type": "tacotron2", "dim_emb": 512, "dim_out": 512},
        "decoder": {"type": "tacotron2prod", "r": 1, "dim_pre": 256, "dim_att": 1024, "dim_rnn": [1024, 1024]},
        "postnet": {"type": "tacotron2", "dim_hidden": 512, "num_layers": 3},
    },
}
# the reference's other shipped configs (configs/config-rdh.yaml:54-69, config-sandra.yaml:54-69):
# Taco2DecoderCell; sandra additionally r = 2, narrower model and MelPostnet2
RDH = {
    "text": {"alphabet": "x" * 39}, "audio": {"num_mels": 80},
    "model": {"encoder": {"type": "tacotron2", "dim_emb": 512, "dim_out": 512},
              "decoder": {"type": "tacotron2", "r": 1, "dim_pre": 256, "dim_att": 256, "dim_rnn": [1024, 1024]},
              "postnet": {"type": "tacotron2", "dim_hidden": 512, "num_layers": 3}},
}
SANDRA = {
    "text": {"alphabet": "x" * 39}, "audio": {"num_mels": 80},
    "model": {"encoder": {"type": "tacotron2", "dim_emb": 256, "dim_out": 256},
              "decoder": {"type": "tacotron2", "r": 2, "dim_pre": 256, "dim_att": 256, "dim_rnn": [512, 512]},
              "postnet": {"dim_hidden": 256, "num_layers": 3}},
}
CONFIGS = {"ljspeech": LJSPEECH, "rdh": RDH, "sandra": SANDRA}
FRAME_SEC = 256.0 / 22050.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# Algorithmic bytes per decode step (SURVEY.md 8d / BASELINE.md 4), split by kernel.
# weights (fp32 params incl. biases) + per-utterance reads/writes, L = memory length.
def step_bytes(B, L, d):
    P, D, Ha, Hd, M = d["dim_pre"], 512, d["dim_rnn"][0], d["dim_rnn"][1], 80
    w = {
        "prenet": (P * M + P + P * P + P) * 4,
        "lstm_att": (4 * Ha * (P + D + Ha) + 8 * Ha) * 4,
        "query": D * Ha * 4,
        "lstm_dec": (4 * Hd * (Ha + D + Hd) + 8 * Hd) * 4,
        "proj": ((M + 1) * (Hd + D) + M + 1) * 4,
    }
    per_utt = {
        "prenet": M * 4 + 2 * P,                       # y_prev + 2 uint8 masks
        "lstm_att": 4 * Ha * 4 + D * 4,                # h,c read+write + ctx read
        "query": 0,
        "attention": L * D * 4 + 2 * L * 4 + L * 4 + D * 4,  # memory pass + w r/w + w_out + ctx write
        "lstm_dec": 4 * Hd * 4,
        "proj": M * 4 + 4,                             # y + s out
    }
    out = {k: w.get(k, 0) + B * per_utt.get(k, 0) for k in set(w) | set(per_utt)}
    out["step"] = sum(out.values())
    return out


FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak


def vits2_flops(Tx, Ty, d):
    """Algorithmic FLOPs of one utterance through TextEncoder (Tx tokens) + reverse flow (Ty frames)."""
    def stack(T, C, F, k, layers, heads):
        per_frame = 2 * (4 * C * C + 2 * k * C * F)           # q,k,v,o 1x1 convs + the two FFN convs
        attn = 2 * 2 * T * C                                   # QK^T and PV per frame (all heads)
        return layers * T * (per_frame + attn)
    H, I, Fh = d["hidden_channels"], d["inter_channels"], d["flow_hidden"]
    te = stack(Tx, H, d["filter_channels"], d["kernel_size"], d["n_layers"], d["n_heads"]) + Tx * 2 * H * 2 * I
    half = I // 2
    per_flow = stack(Ty, half, half, 3, 2, 2) + Ty * 2 * (half * Fh + Fh * half)
    for j in range(d["flow_wn_layers"]):
        per_flow += Ty * 2 * (d["flow_kernel"] * Fh * 2 * Fh + Fh * (2 * Fh if j < d["flow_wn_layers"] - 1 else Fh))
    return te, d["n_flows"] * per_flow


def bench_vits2(args, T, dist, dev, world, rank):
    """Second hot path: one step = TextEncoder over [B, 120] tokens + the reverse
    flow over [B, 192, 600] latent frames.  Utterances are independent: ranks take equal shares, no collective."""
    import warnings

    warnings.filterwarnings("ignore", category=FutureWarning)
    B = args.batch if args.batch != 256 else 64
    Tx, Ty = args.mem_len, args.frames
    D = dict(n_vocab=178, inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3, window_size=4,
             flow_hidden=192, flow_kernel=5, flow_wn_layers=4, n_flows=4)
    torch.manual_seed(42)
    te = T.vits2.TextEncoder(D["n_vocab"], D["inter_channels"], D["hidden_channels"], D["filter_channels"], D["n_heads"], D["n_layers"],
                             D["kernel_size"], 0.1).to(dev).eval()
    fl = T.vits2.ResidualCouplingTransformersBlock(D["inter_channels"], D["flow_hidden"], D["flow_kernel"], 1, D["flow_wn_layers"], n_flows=D["n_flows"],
                                                   use_transformer_flows=True).to(dev).eval()
    for m in fl.modules():  # the reference zero-initialises `post`; give it weight so the coupling does real work
        if isinstance(m, T.vits2.ResidualCouplingTransformersLayer):
            torch.nn.init.normal_(m.post.weight, 0.0, 0.05)
    g = torch.Generator().manual_seed(1234 + rank)
    ids = torch.randint(0, D["n_vocab"], (B, Tx), generator=g).to(dev)
    xl = torch.full((B,), Tx, device=dev)
    z = torch.randn(B, D["inter_channels"], Ty, generator=g).to(dev)
    ym = torch.ones(B, 1, Ty, device=dev)

    def one_step():
        with torch.no_grad():
            a = te(ids, xl)
            return a, fl(z, ym, reverse=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        one_step()
    fence()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    te_ms = fl_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        with torch.no_grad():
            ev[0].record(); te(ids, xl); ev[1].record(); out = fl(z, ym, reverse=True); ev[2].record()
        ev[2].synchronize()
        te_ms += ev[0].elapsed_time(ev[1]); fl_ms += ev[1].elapsed_time(ev[2])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert bool(torch.isfinite(out).all())
    Bg = B * world
    f_te, f_fl = vits2_flops(Tx, Ty, D)
    fl_s = fl_ms / args.steps * 1e-3
    res = {
        "metric": "mel-frames/s (whole node) + real-time-factor, LJSpeech 22.05kHz hop256",
        "value": round(Bg * Ty * args.steps / elapsed, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"vits2 second hot path (BASELINE.json configs[4]): TextEncoder [B={B}/GPU, {Tx} tokens] + reverse flow "
                               f"[B, 192, {Ty} frames], ModelConfig defaults", "global_batch": Bg, "parallelism": f"utterance-shard x{world}"},
        "rtf": round((elapsed / args.steps) / (Ty * FRAME_SEC), 6),
        "text_encoder_ms": round(te_ms / args.steps, 3), "flow_reverse_ms": round(fl_ms / args.steps, 3),
        "roofline": {"bound": "mfma", "kernel": "flow_reverse (whole pass: GEMMs + attention)", "achieved": round(B * f_fl / fl_s / 1e12, 2),
                     "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(B * f_fl / fl_s / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                     "traffic": None, "alg_flops_per_utterance": {"text_encoder": f_te, "flow_reverse": f_fl}},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import vits2_oracle as V

        d = V.Vits2Dims()
        wts = {"enc_p." + k: v.detach().cpu() for k, v in te.state_dict().items()}
        wts.update({"flow." + k: v.detach().cpu() for k, v in fl.state_dict().items()})
        cores = max(1, min(16, os.cpu_count() or 1, torch.get_num_threads()))
        torch.set_num_threads(cores)
        bc = min(B, 4)
        t1 = time.perf_counter()
        with torch.no_grad():
            V.text_encoder(ids[:bc].cpu(), xl[:bc].cpu(), wts, d)
            V.flow_reverse(z[:bc].cpu(), ym[:bc].cpu(), wts, d)
        ct = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": round(bc * Ty / ct, 1), "unit": "mel-frames/s", "cores": cores, "kind": "port",
                               "sample": f"oracle (torch-CPU restatement of the reference blocks) on B={bc}: TextEncoder {Tx} tokens + reverse flow {Ty} frames, {cores} threads"}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--mem-len", type=int, default=120)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--postnet", choices=["f32", "bf16", "split_f16"], default="split_f16")
    ap.add_argument("--precision", choices=["f32", "split_f16"], default="split_f16",
                    help="arithmetic of the LSTM gate GEMMs (include/ttsdec.h TTSDEC_PREC_*)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="ljspeech",
                    help="model dims: ljspeech = BASELINE.json's config (default); rdh / sandra = the other shipped configs")
    ap.add_argument("--workload", choices=["tacotron", "vits2"], default="tacotron",
                    help="tacotron = the headline decoder path (default); vits2 = the second hot path of BASELINE.json configs[4] "
                         "(TextEncoder on [B, 120] + reverse flow on [B, 192, 600], ModelConfig defaults; --batch defaults to 64)")
    ap.add_argument("--dropout", choices=["philox", "masks"], default="philox",
                    help="PreNet dropout source: philox = drawn on the device (default, SURVEY 8d 'mode 2'); masks = injected keep-masks "
                         "resident in HBM ('mode 1': what the parity runs use)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="decode frames for the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # Rehearsal switch (not for measurements): TTSDEC_BENCH_BACKEND=gloo lets several ranks share the
    # GPUs of a smaller box to exercise the N > 1 code path; the real run is RCCL, one rank per GPU.
    backend = os.environ.get("TTSDEC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % ndev if backend != "nccl" else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import torch_tts_amd as T
    from torch_tts_amd import _lib
    from torch_tts_amd import distributed as D

    if args.workload == "vits2":
        return bench_vits2(args, T, dist, dev, world, rank)

    B, L, NF = args.batch, args.mem_len, args.frames
    Bg = B * world  # global batch

    # ---- model: LJSpeech dims, random init seed 42 (xavier_normal gain 1.5, default LSTMCell init) ----
    torch.manual_seed(42)
    CFG = CONFIGS[args.config]
    if args.config != "ljspeech":
        args.no_cpu_baseline = True  # the CPU leg and the byte model below are written for the LJSpeech cell
    model = T.build_tacotron(CFG).eval()
    model.to(dev)
    dec, post = model.decoder, model.postnet
    dec.dropout_source, dec.dropout_seed = "philox", 123
    dec.precision = args.precision
    post.precision = args.postnet

    # ---- weights: rank 0 packs, one RCCL broadcast of the blob, other ranks bind ----
    eng = dec._engines.get(dec.decoder_cell.engine_dims(), dev)
    peng = post._engines.get(post.engine_dims(), dev)
    if world > 1:
        D.broadcast_engine_weights(eng, dec.weight_tensors(), src=0)
        D.broadcast_engine_weights(peng, post.weight_tensors(), src=0)
        eng._fingerprint = None
    else:
        eng.ensure_packed(dec.weight_tensors())
        peng.ensure_packed(post.weight_tensors())
    eng.set_precision(args.precision)

    # ---- synthetic input: ids -> stock encoder -> memory, this rank's shard of the global batch ----
    g = torch.Generator().manual_seed(1234)
    ids_all = torch.randint(1, 40, (Bg, L), generator=g)
    lo, hi = D.shard_bounds(Bg, world, rank)
    ids = ids_all[lo:hi].to(dev)
    lens = torch.full((hi - lo,), L, dtype=torch.long, device=dev)
    with torch.no_grad():
        mem = torch.cat([model.encoder(ids[i : i + 64], lens[i : i + 64]) for i in range(0, hi - lo, 64)]).contiguous()
    assert mem.shape == (B, L, CFG["model"]["encoder"]["dim_out"])
    R = CFG["model"]["decoder"]["r"]
    NS = NF // R  # decode steps for NF frames

    y = torch.empty(B, NF, 80, device=dev)
    s = torch.empty(B, NF, device=dev)
    w = torch.empty(B, NS, L, device=dev)
    t_out = torch.zeros(2, dtype=torch.int32, device=dev)
    prec = {"f32": _lib.POSTNET_F32, "bf16": _lib.POSTNET_BF16, "split_f16": _lib.POSTNET_SPLIT_F16}[args.postnet]

    dmode, dmasks = _lib.DROPOUT_PHILOX, None
    if args.dropout == "masks":  # [NS, 2, B, d_pre] uint8 keep-masks (p = 0.5), generated outside the timed region
        dd = CFG["model"]["decoder"]
        ph = 128 if dd["type"] == "tacotron2" else dd["dim_pre"]
        gm = torch.Generator(device=dev).manual_seed(123 + rank)
        dmasks = torch.randint(0, 2, (NS, B * (ph + dd["dim_pre"])), generator=gm, device=dev, dtype=torch.uint8)
        dmode = _lib.DROPOUT_MASKS

    def one_step():
        eng.decode(mem, t_begin=0, n_steps=NS, stop_threshold=-2.0, check_stop=True, dropout_mode=dmode,
                   masks=dmasks, seed=123, teacher=None, teacher_flags=None, y=y, s=s, w=w, t_out=t_out)
        return peng.postnet(y, prec)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter()
    dec_ms = 0.0
    for _ in range(args.steps):
        ev0.record()
        eng.decode(mem, t_begin=0, n_steps=NS, stop_threshold=-2.0, check_stop=True, dropout_mode=dmode,
                   masks=dmasks, seed=123, teacher=None, teacher_flags=None, y=y, s=s, w=w, t_out=t_out)
        ev1.record()
        y_post = peng.postnet(y, prec)
        ev2.record()
        ev2.synchronize()
        dec_ms += ev0.elapsed_time(ev1)
    fence()
    elapsed = time.perf_counter() - t0
    assert t_out.tolist() == [NS, 0], f"decode ended early: {t_out.tolist()}"
    assert bool(torch.isfinite(y_post).all())
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    frames_total = Bg * NF * args.steps
    value = frames_total / elapsed
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- roofline of the dominant kernel: HIP events inside the library, on the launch stream ----
    kms = eng.profile_step(mem, iters=50, dropout_mode=_lib.DROPOUT_PHILOX, masks=None, seed=123)
    bytes_k = step_bytes(B, L, LJSPEECH["model"]["decoder"])  # (byte model of the LJSpeech cell)
    grp = {"prenet": ["prenet", "prenet0", "prenet1"], "lstm_att": ["lstm_att"], "query": ["query"],
           "attention": ["attention"], "lstm_dec": ["lstm_dec"], "proj": ["proj"]}
    per_kernel = {}
    for k, names in grp.items():
        ms = sum(kms[n] for n in names if n in kms)
        per_kernel[k] = {"ms": round(ms, 5), "alg_bytes": bytes_k[k], "GBps": round(bytes_k[k] / (ms * 1e-3) / 1e9, 1)}
    dom = max(per_kernel, key=lambda k: per_kernel[k]["ms"])
    step_ms_kernels = sum(v["ms"] for v in per_kernel.values())
    decode_step_ms = dec_ms / args.steps / NS
    # HBM traffic of the dominant kernel from the PMC passes (cannot be collected inside this
    # process; see profiles/r01_traffic.json for the command and the gfx950 FETCH_SIZE correction)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if args.precision == "split_f16" and B == 256 and L == 120:
            traffic = tj["per_launch"].get(dom, {}).get("hbm_bytes")
    except Exception:
        traffic = None
    roofline = {
        "bound": "hbm",
        "kernel": dom,
        "achieved": per_kernel[dom]["GBps"],
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(per_kernel[dom]["GBps"] / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "alg_bytes_per_launch": per_kernel[dom]["alg_bytes"],
        "kernel_ms": per_kernel[dom]["ms"],
        "decode_step": {
            "alg_bytes": bytes_k["step"],
            "ms_in_loop": round(decode_step_ms, 5),
            "GBps": round(bytes_k["step"] / (decode_step_ms * 1e-3) / 1e9, 1),
            "frac": round(bytes_k["step"] / (decode_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "sum_kernel_ms": round(step_ms_kernels, 5),
        },
        "per_kernel": per_kernel,
    }

    out = {
        "metric": "mel-frames/s (whole node) + real-time-factor, LJSpeech 22.05kHz hop256",
        "value": round(value, 1),
        "unit": "mel-frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f32" if args.precision == "f32" else "f32 via split-fp16 (hi+lo fp16 planes, 3 f16 MFMA products, fp32 accumulate) for the LSTM and PreNet GEMMs; f32 elsewhere")
                 + {"f32": "; Postnet f32", "split_f16": "; Postnet split-fp16", "bf16": "; Postnet bf16"}[args.postnet],
        "data": "synthetic",
        "config": {
            "workload": f"{args.config} dims, batch={B}/GPU (global {Bg}), L={L}, {NF} decode frames + Postnet({args.postnet}); "
                        "BASELINE.json configs[2] per GPU, configs[3] at 8 GPUs",
            "global_batch": Bg, "mem_len": L, "frames": NF, "dropout": args.dropout, "parallelism": f"utterance-shard x{world}",
            "lstm_precision": eng.precision(),
        },
        "rtf": round((elapsed / args.steps) / (NF * FRAME_SEC), 6),
        "audio_seconds_per_s": round(value * FRAME_SEC, 1),
        "decode_only_frames_per_s": round(Bg * NF / (dec_ms / args.steps * 1e-3), 1),
        "roofline": roofline,
    }

    # ---- CPU baseline: the oracle (a port of the reference's CPU path) on this box's host cores ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import tacotron_oracle as O

        dims = O.DecoderDims()
        sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
        pw = {k: v.detach().cpu() for k, v in post.state_dict().items()}
        memc = mem.cpu()
        # the GPU box gives one GPU a 16-CPU share: do not oversubscribe it
        cores = max(1, min(16, os.cpu_count() or 1, torch.get_num_threads()))
        torch.set_num_threads(cores)
        tc = args.cpu_frames
        if tc <= 0:
            t1 = time.perf_counter()
            O.decode(sd, dims, memc, max_steps=1, masks=O.synthetic_masks(2, B, 256))
            per = (time.perf_counter() - t1) / 2
            tc = max(4, min(NF, int(15.0 / max(per, 1e-4))))
        masks = O.synthetic_masks(tc, B, 256)
        t1 = time.perf_counter()
        with torch.no_grad():
            cy, _, _ = O.decode(sd, dims, memc, max_steps=tc - 1, masks=masks)
            O.mel_postnet(cy, pw, 3)
        ct = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": round(B * tc / ct, 1), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU restatement of the reference path) on B={B}, L={L}, {tc} decode frames + Postnet, {cores} threads",
        }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
