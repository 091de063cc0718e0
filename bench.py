#!/usr/bin/env python3
"""Benchmark of the Tacotron decoder hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic LJSpeech-shaped input:
600 autoregressive decode frames (max_steps=599; random-init stop logits never cross -2.0)
for B utterances per GPU with memory length L=120, followed by the Postnet over the
[B, 600, 80] mel.  Metric: mel-frames/s over the whole job (all ranks), inputs resident
in HBM.  Weak scaling: per-GPU batch fixed at 256 (BASELINE.json configs[2]/[3]:
2048 = 8 x 256).  Rank 0 prints ONE JSON line.

With --gpus N > 1 and no torchrun environment the script starts its own N ranks (a
`python -m torch.distributed.run` child, before this process has made any GPU call) and
exits with the child's code; under torchrun it is one of the ranks.

One invocation times these legs, each with the same K steps / W warm-up steps, barrier +
synchronize on both sides and the maximum over ranks:
  * the headline leg (value / dtype / roofline of the line): the reference's own arithmetic - exact fp32 on the
    fp32-input matrix instruction for every GEMM, Postnet fp32 (--precision / --postnet change it);
  * "split_f16": the same workload in the library's opt-in split-fp16 mode (include/ttsdec.h TTSDEC_PREC_SPLIT_F16:
    two fp16 planes per operand, narrower than fp32 - a named sub-record, never the headline);
  * "b64_f32": BASELINE.json configs[1] (batch 64 per GPU, fp32);
  * "vits2": BASELINE.json configs[4] (TextEncoder + reverse flow, --workload vits2's step on a short run) - in the
    headline's arithmetic, with the other mode nested as ITS sub-record.
The injected-mask parity of the timed configurations against the CPU oracle (all 600 frames when the CPU sample
covers them), including the bf16 Postnet of configs[2], is reported under "parity".
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

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
METRIC = "mel-frames/s (whole node) + real-time-factor, LJSpeech 22.05kHz hop256"
DTYPE_TEXT = {
    "f32": "f32 (exact fp32 matrix instruction for every GEMM)",
    "split_f16": "f32 via split-fp16 (hi+lo fp16 planes = 22 significand bits per operand, 3 f16 MFMA products, fp32 accumulate) "
                 "for the LSTM, PreNet, query and projection GEMMs; f32 elsewhere",
}
POSTNET_TEXT = {"f32": "; Postnet f32", "split_f16": "; Postnet split-fp16", "bf16": "; Postnet bf16"}


def cell_model(cfg):
    """The decoder cell's sizes as the byte / FLOP model needs them, from the SELECTED config (reference: tacotron.py:165-214,
    decoder_cell.py:66-140 Taco2DecoderCell, :143-195 Taco2ProdDecoderCell)."""
    dd = cfg["model"]["decoder"]
    taco2 = dd["type"] == "tacotron2"
    r, M = dd["r"], cfg["audio"].get("num_mels", 80)
    D, Ha, Hd, P = cfg["model"]["encoder"]["dim_out"], dd["dim_rnn"][0], dd["dim_rnn"][1], dd["dim_pre"]
    return {
        "taco2": taco2, "r": r, "M": M, "D": D, "Ha": Ha, "Hd": Hd, "P": P,
        "P0": 128 if taco2 else P,                      # PreNet hidden width (decoder_cell.py:74-76 / :152)
        "Kq": Ha + Hd if taco2 else Ha,                 # query input: cat[h0, h1, zeros] (the zero block is not read) / h_att
        "Kproj": Ha + Hd if taco2 else Hd + D,          # projection input: cat[h0, h1, zeros] / cat[h_dec, ctx]
        "Nproj": r * M + r,                             # fc_mel + fc_stop rows
    }


# Algorithmic bytes per decode step (SURVEY.md 8d / BASELINE.md 4), split by kernel.
# weights (fp32 params incl. biases) + per-utterance reads/writes, L = memory length; m = cell_model(config).
def step_bytes(B, L, m):
    P, P0, D, Ha, Hd, M, r = m["P"], m["P0"], m["D"], m["Ha"], m["Hd"], m["M"], m["r"]
    w = {
        "prenet": (P0 * M + P0 + P * P0 + P) * 4,
        "lstm_att": (4 * Ha * (P + D + Ha) + 8 * Ha) * 4,
        "query": D * m["Kq"] * 4,
        "lstm_dec": (4 * Hd * (Ha + D + Hd) + 8 * Hd) * 4,
        "proj": (m["Nproj"] * m["Kproj"] + m["Nproj"]) * 4,
    }
    per_utt = {
        "prenet": M * 4 + P0 + P,                      # y_prev (the last frame of the previous step) + the two uint8 masks
        "lstm_att": 4 * Ha * 4 + D * 4,                # h,c read+write + ctx read
        "query": 0,
        "attention": L * D * 4 + 2 * L * 4 + L * 4 + D * 4,  # memory pass + w r/w + w_out + ctx write
        "lstm_dec": 4 * Hd * 4,
        "proj": r * M * 4 + r * 4,                     # y + s out
    }
    out = {k: w.get(k, 0) + B * per_utt.get(k, 0) for k in set(w) | set(per_utt)}
    out["step"] = sum(out.values())
    return out


def step_flops(B, L, m):
    """2 per weight of the five GEMM groups (biases excluded) + the two passes over `memory`."""
    P, P0, D, Ha, Hd, M = m["P"], m["P0"], m["D"], m["Ha"], m["Hd"], m["M"]
    n_w = P0 * M + P * P0 + 4 * Ha * (P + D + Ha) + D * m["Kq"] + 4 * Hd * (Ha + D + Hd) + m["Nproj"] * m["Kproj"]
    return B * (2 * n_w + 4 * L * D)


F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16/f16 matrix peak
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32: 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz


def postnet_flops_per_frame(cfg):
    """MelPostnet (tacotron/modules/modules.py: num_layers x [Conv1d(k=5) -> BN -> isru] -> Linear + residual): 2 per weight."""
    pn, M = cfg["model"]["postnet"], cfg["audio"].get("num_mels", 80)
    H, n, k = pn["dim_hidden"], pn["num_layers"], 5
    return 2 * (k * M * H + (n - 1) * k * H * H + H * M)


def sources_digest():
    """sha256 over the decoder's kernel sources: ties a PMC capture of the decode step (profiles/*traffic.json) to the code it was
    taken on.  (encoder.hip, vits2.hip and conv256.hip hold no kernel of the step: editing them leaves the capture valid.)"""
    h = hashlib.sha256()
    for root in (os.path.join(ROOT, "torch-tts_amd", "csrc"), os.path.join(ROOT, "include")):
        for fn in sorted(os.listdir(root)):
            if fn.endswith((".hip", ".h")) and fn not in ("encoder.hip", "vits2.hip", "conv256.hip"):
                with open(os.path.join(root, fn), "rb") as f:
                    h.update(fn.encode())
                    h.update(f.read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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


def bench_vits2(args, T, torch, dist, dev, world, rank, cpu_baseline=True):
    """Second hot path: one step = TextEncoder over [B, 120] tokens + the reverse
    flow over [B, 192, 600] latent frames.  Utterances are independent: ranks take equal shares, no collective.
    Returns the record (rank 0 prints it when this is the invocation's workload; the tacotron workload carries it as
    its "vits2" sub-record)."""
    import warnings

    warnings.filterwarnings("ignore", category=FutureWarning)
    B = args.batch if (args.workload == "vits2" and args.batch != 256) else 64
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

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    Bg = B * world
    f_te, f_fl = vits2_flops(Tx, Ty, D)

    def timed(precision, steps, warmup):
        """K timed passes in one arithmetic mode: exact fp32 (the reference's own) or the library's split-fp16."""
        te.precision = fl.precision = precision

        def one_step():
            with torch.no_grad():
                a = te(ids, xl)
                return a, fl(z, ym, reverse=True)

        for _ in range(max(1, warmup)):
            one_step()
        fence()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        te_ms = fl_ms = 0.0
        out = None
        t0 = time.perf_counter()
        for _ in range(steps):
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
        fl_s = fl_ms / steps * 1e-3
        if precision == "f32":
            peak, note = 157.3, "fp32 matrix instruction (v_mfma_f32_32x32x2_f32), 157.3 TFLOP/s"
            dtype = "f32 (exact fp32 matrix instruction for every GEMM and for the attention's QK^T / PV); elementwise math f32"
        else:
            # every GEMM runs as 3 f16 MFMA products per fp32 product (split-fp16), so the matrix pipe that bounds the
            # pass is the f16 one and one algorithmic FLOP costs three of its FLOPs
            peak, note = F16_MFMA_PEAK_TFLOPS / 3.0, "dense f16 MFMA peak (2500 TFLOP/s) / 3 products per fp32 product"
            dtype = ("f32 via split-fp16 (hi+lo fp16 planes, 3 f16 MFMA products, fp32 accumulate) for every GEMM and the flow's attention; "
                     "elementwise math f32")
        return {
            "value": round(Bg * Ty * steps / elapsed, 1), "unit": "mel-frames/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed * 1e3 / steps, 3), "dtype": dtype,
            "rtf": round((elapsed / steps) / (Ty * FRAME_SEC), 6),
            "text_encoder_ms": round(te_ms / steps, 3), "flow_reverse_ms": round(fl_ms / steps, 3),
            "roofline": {"bound": "mfma", "kernel": "flow_reverse (whole pass: GEMMs + attention)", "achieved": round(B * f_fl / fl_s / 1e12, 2),
                         "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(B * f_fl / fl_s / 1e12 / peak, 4), "peak_note": note,
                         "traffic": None, "alg_flops_per_utterance": {"text_encoder": f_te, "flow_reverse": f_fl}},
        }

    prec = getattr(args, "vits2_precision", None) or ("f32" if args.precision == "f32" else "split_f16")
    head = timed(prec, args.steps, args.warmup)
    res = {
        "metric": METRIC, "value": head["value"], "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": head["dtype"],
        "data": "synthetic",
        "config": {"workload": f"vits2 second hot path (BASELINE.json configs[4]): TextEncoder [B={B}/GPU, {Tx} tokens] + reverse flow "
                               f"[B, 192, {Ty} frames], ModelConfig defaults", "global_batch": Bg, "parallelism": f"utterance-shard x{world}",
                   "precision": prec},
        "rtf": head["rtf"], "text_encoder_ms": head["text_encoder_ms"], "flow_reverse_ms": head["flow_reverse_ms"], "roofline": head["roofline"],
    }
    if not args.no_extra_legs:  # the other arithmetic mode as a named sub-record
        other = "split_f16" if prec == "f32" else "f32"
        leg = timed(other, args.steps, args.warmup)
        res["split_f16" if other == "split_f16" else "f32_exact"] = dict(
            leg, what=("the same pass in the library's opt-in split-fp16 mode (two fp16 planes per operand: narrower than fp32)" if other == "split_f16"
                       else "the same pass on the reference's own arithmetic: exact fp32"))
        te.precision = fl.precision = prec
    if rank == 0 and cpu_baseline and not args.no_cpu_baseline:
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
        res["cpu_baseline"] = {"value": round(bc * Ty / ct, 1), "unit": "mel-frames/s", "cores": cores, "cpu": cpu_model(), "kind": "port",
                               "sample": f"oracle (torch-CPU restatement of the reference blocks) on B={bc}: TextEncoder {Tx} tokens + reverse flow {Ty} frames, {cores} threads"}
    return res


class Workload:
    """The model (built once), and per batch size the resident inputs / output buffers."""

    def __init__(self, args, T, torch, D, dev, world, rank):
        self.args, self.T, self.torch, self.D, self.dev, self.world, self.rank = args, T, torch, D, dev, world, rank
        from torch_tts_amd import _lib

        self._lib = _lib
        # model: LJSpeech dims, random init seed 42 (xavier_normal gain 1.5, default LSTMCell init)
        torch.manual_seed(42)
        self.cfg = CONFIGS[args.config]
        self.model = T.build_tacotron(self.cfg).eval().to(dev)
        self.dec, self.post = self.model.decoder, self.model.postnet
        self.dec.dropout_source, self.dec.dropout_seed = "philox", 123
        # weights: rank 0 packs, one RCCL broadcast of the blob, other ranks bind
        self.eng = self.dec._engines.get(self.dec.decoder_cell.engine_dims(), dev)
        self.peng = self.post._engines.get(self.post.engine_dims(), dev)
        if world > 1:
            D.broadcast_engine_weights(self.eng, self.dec.weight_tensors(), src=0)
            D.broadcast_engine_weights(self.peng, self.post.weight_tensors(), src=0)
        else:
            self.eng.ensure_packed(self.dec.weight_tensors())
            self.peng.ensure_packed(self.post.weight_tensors())
        self.R = self.cfg["model"]["decoder"]["r"]
        self._inputs = {}

    def inputs(self, B):
        """ids -> stock encoder -> memory for this rank's shard of the global batch (B per GPU), plus output buffers."""
        if B in self._inputs:
            return self._inputs[B]
        torch, dev, L, NF = self.torch, self.dev, self.args.mem_len, self.args.frames
        Bg = B * self.world
        g = torch.Generator().manual_seed(1234)
        ids_all = torch.randint(1, 40, (Bg, L), generator=g)
        lo, hi = self.D.shard_bounds(Bg, self.world, self.rank)
        ids = ids_all[lo:hi].to(dev)
        lens = torch.full((hi - lo,), L, dtype=torch.long, device=dev)
        with torch.no_grad():
            mem = torch.cat([self.model.encoder(ids[i : i + 64], lens[i : i + 64]) for i in range(0, hi - lo, 64)]).contiguous()
        assert mem.shape == (B, L, self.cfg["model"]["encoder"]["dim_out"])
        NS = NF // self.R
        io = {
            "mem": mem, "NS": NS,
            "y": torch.empty(B, NF, 80, device=dev), "s": torch.empty(B, NF, device=dev), "w": torch.empty(B, NS, L, device=dev),
            "t_out": torch.zeros(2, dtype=torch.int32, device=dev),
        }
        self._inputs[B] = io
        return io

    def decode(self, io, mode, masks):
        self.eng.decode(io["mem"], t_begin=0, n_steps=io["NS"], stop_threshold=-2.0, check_stop=True, dropout_mode=mode,
                        masks=masks, seed=123, teacher=None, teacher_flags=None, y=io["y"], s=io["s"], w=io["w"], t_out=io["t_out"])

    def time_leg(self, dist, B, precision, postnet, steps, warmup, dropout="philox"):
        """K timed steps of decode + Postnet at batch B per GPU; returns the leg's record."""
        torch, dev, world, _lib = self.torch, self.dev, self.world, self._lib
        io = self.inputs(B)
        self.eng.set_precision(precision)
        pprec = {"f32": _lib.POSTNET_F32, "bf16": _lib.POSTNET_BF16, "split_f16": _lib.POSTNET_SPLIT_F16}[postnet]
        mode, masks = _lib.DROPOUT_PHILOX, None
        if dropout == "masks":  # [NS, 2, B, d_pre] uint8 keep-masks (p = 0.5), generated outside the timed region
            dd = self.cfg["model"]["decoder"]
            ph = 128 if dd["type"] == "tacotron2" else dd["dim_pre"]
            gm = torch.Generator(device=dev).manual_seed(123 + self.rank)
            masks = torch.randint(0, 2, (io["NS"], B * (ph + dd["dim_pre"])), generator=gm, device=dev, dtype=torch.uint8)
            mode = _lib.DROPOUT_MASKS

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(warmup):
            self.decode(io, mode, masks)
            self.peng.postnet(io["y"], pprec)
        fence()
        ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        t0 = time.perf_counter()
        dec_ms = 0.0
        y_post = None
        for _ in range(steps):
            ev0.record()
            self.decode(io, mode, masks)
            ev1.record()
            y_post = self.peng.postnet(io["y"], pprec)
            ev2.record()
            ev2.synchronize()
            dec_ms += ev0.elapsed_time(ev1)
        fence()
        elapsed = time.perf_counter() - t0
        assert io["t_out"].tolist() == [io["NS"], 0], f"decode ended early or saturated: {io['t_out'].tolist()}"
        assert bool(torch.isfinite(y_post).all())
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        NF, Bg = self.args.frames, B * world
        value = Bg * NF * steps / elapsed
        return {
            "value": round(value, 1), "ms_per_step": round(elapsed * 1e3 / steps, 3),
            "rtf": round((elapsed / steps) / (NF * FRAME_SEC), 6),
            "decode_only_frames_per_s": round(Bg * NF / (dec_ms / steps * 1e-3), 1),
            "decode_step_ms": dec_ms / steps / io["NS"],
            "lstm_precision": self.eng.precision(), "postnet_precision": postnet, "batch_per_gpu": B, "global_batch": Bg,
        }

    def time_postnet_modes(self, B, steps):
        """MelPostnet alone on y [B, frames, 80] in exact fp32 / split-fp16 / bf16: ms per pass and the fraction of the matrix
        peak of the instruction each mode runs on (algorithmic FLOPs: postnet_flops_per_frame)."""
        torch, _lib = self.torch, self._lib
        io = self.inputs(B)
        flop = postnet_flops_per_frame(self.cfg) * B * self.args.frames
        peaks = {"f32": F32_MFMA_PEAK_TFLOPS, "split_f16": F16_MFMA_PEAK_TFLOPS / 3.0, "bf16": F16_MFMA_PEAK_TFLOPS}
        modes = {"f32": _lib.POSTNET_F32, "split_f16": _lib.POSTNET_SPLIT_F16, "bf16": _lib.POSTNET_BF16}
        rec = {"what": "MelPostnet alone, ms per pass over this rank's [B, frames, 80]; frac = algorithmic TFLOP/s over the dense matrix peak "
                       "of the mode's instruction (split-fp16: three f16 products per term)",
               "batch_per_gpu": B, "frames": self.args.frames, "alg_tflop_per_pass": round(flop / 1e12, 4)}
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for name, pm in modes.items():
            self.peng.postnet(io["y"], pm)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(steps):
                yp = self.peng.postnet(io["y"], pm)
            ev1.record()
            ev1.synchronize()
            ms = ev0.elapsed_time(ev1) / steps
            tf = flop / (ms * 1e-3) / 1e12
            rec[name] = {"ms": round(ms, 4), "tflop_per_s": round(tf, 1), "peak_tflop_per_s": round(peaks[name], 1), "frac": round(tf / peaks[name], 4),
                         "finite": bool(torch.isfinite(yp).all())}
        return rec

    def roofline(self, B, precision, decode_step_ms):
        """Roofline record of the decode STEP (north_star's quantity) and of every launch in it, from in-loop times.

        frac = algorithmic bytes of one step / the step's time inside the timed loop / 8 TB/s.  The per-launch times come from
        the same captured-graph loop run once more with workgroup 0 of every step kernel storing its start time
        (ttsdec_profile_loop; one 8-byte store per launch): each entry is the span from a launch's start to the next launch's
        start inside the replayed graph - the launch's duration in the loop plus the gap behind it - so the entries add up to
        that loop's step time, which is printed next to the timed loop's.
        Exact fp32 is bound by the fp32 matrix pipe above ~46 utterances: that leg reports the FLOP fraction as
        its roofline and the HBM fraction beside it."""
        args, _lib = self.args, self._lib
        io = self.inputs(B)
        self.eng.set_precision(precision)
        cm = cell_model(self.cfg)  # (byte / FLOP model of the SELECTED config's cell)
        bytes_k = step_bytes(B, args.mem_len, cm)
        alg = dict(bytes_k)
        alg.update({"prenet0": 0, "prenet1": 0})

        def alg_bytes(name):  # "a+b+c" = one launch running those roles: the sum of their bytes
            return sum(alg.get(part, 0) for part in name.split("+"))

        loop_ms, loop_step_ms = self.eng.profile_loop(io["mem"], n_steps=600, dropout_mode=_lib.DROPOUT_PHILOX, masks=None, seed=123)
        alone_ms = self.eng.profile_step(io["mem"], iters=50, dropout_mode=_lib.DROPOUT_PHILOX, masks=None, seed=123)
        # HBM-side traffic per launch from the PMC passes: those cannot be collected inside this process (rocprofv3 --pmc,
        # separate passes), so the figures come from the committed capture - and only when that capture was taken on exactly
        # these kernel sources (digest match), else null
        traffic, traffic_src = {}, None
        pdir = os.path.join(ROOT, "profiles")
        for fn in sorted(f for f in os.listdir(pdir) if f.startswith("r04_traffic") and f.endswith(".json")):
            try:
                tj = json.load(open(os.path.join(pdir, fn)))
                if tj.get("config", "ljspeech") != args.config:
                    continue
                if tj.get("sources_digest") == sources_digest() and tj.get("precision") == precision and tj.get("batch") == B:
                    traffic = {k: v.get("hbm_bytes") for k, v in tj["per_launch"].items()}
                    traffic_src = {"file": "profiles/" + fn, "sources_digest": tj["sources_digest"], "commit": tj.get("captured_at_commit")}
            except Exception:
                pass
        per_kernel = {}
        for name, ms in loop_ms.items():
            key = "prenet" if name in ("prenet0", "prenet1") else name
            ent = per_kernel.setdefault(key, {"ms_in_loop": 0.0, "alg_bytes": alg_bytes(key), "ms_alone": 0.0})
            ent["ms_in_loop"] += ms
            ent["ms_alone"] += alone_ms.get(name, 0.0)
        for key, ent in per_kernel.items():
            ent["GBps"] = round(ent["alg_bytes"] / (ent["ms_in_loop"] * 1e-3) / 1e9, 1)
            ent["frac"] = round(ent["GBps"] / HBM_PEAK_GBS, 4)
            ent["traffic"] = traffic.get(key)
            ent["traffic_over_alg"] = round(traffic[key] / ent["alg_bytes"], 3) if traffic.get(key) and ent["alg_bytes"] else None
            ent["ms_in_loop"], ent["ms_alone"] = round(ent["ms_in_loop"], 5), round(ent["ms_alone"], 5)
        flops = step_flops(B, args.mem_len, cm)
        step_s = decode_step_ms * 1e-3
        hbm_gbs = bytes_k["step"] / step_s / 1e9
        if precision == "f32":
            pipe, pipe_peak, pipe_flops = "fp32 matrix instruction (v_mfma_f32_32x32x2_f32)", 157.3, flops
        else:
            pipe, pipe_peak, pipe_flops = "f16 matrix instruction, 3 products per fp32 product (split-fp16)", F16_MFMA_PEAK_TFLOPS, 3 * flops
        tflops = pipe_flops / step_s / 1e12
        t_hbm, t_pipe = bytes_k["step"] / (HBM_PEAK_GBS * 1e9), pipe_flops / (pipe_peak * 1e12)
        hbm = {"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
               "alg_bytes_per_step": bytes_k["step"], "roofline_us": round(t_hbm * 1e6, 2)}
        mfma = {"bound": "mfma", "achieved": round(tflops, 2), "peak": pipe_peak, "unit": "TFLOP/s", "frac": round(tflops / pipe_peak, 4),
                "pipe": pipe, "alg_flops_per_step": flops, "pipe_flops_per_step": pipe_flops, "roofline_us": round(t_pipe * 1e6, 2)}
        top = dict(hbm if t_hbm >= t_pipe else mfma)  # the roofline that binds this leg; the other one rides beside it
        step_traffic = sum(v for v in (per_kernel[k]["traffic"] for k in per_kernel) if v) if traffic and all(per_kernel[k]["traffic"] for k in per_kernel) else None
        top.update({
            "kernel": "decode step = " + " -> ".join(loop_ms.keys()) + " (every launch listed under per_kernel; none singled out)",
            "traffic": step_traffic, "traffic_capture": traffic_src,
            "step_us_in_loop": round(decode_step_ms * 1e3, 3),
            "other_roofline": mfma if t_hbm >= t_pipe else hbm,
            "in_loop_check": {
                "timed_loop_step_us": round(decode_step_ms * 1e3, 3), "stamped_loop_step_us": round(loop_step_ms * 1e3, 3),
                "sum_per_kernel_us": round(sum(v["ms_in_loop"] for v in per_kernel.values()) * 1e3, 3),
                "stamped_over_timed": round(loop_step_ms / decode_step_ms, 4),
                "how": "ttsdec_profile_loop: the same captured-graph decode with workgroup 0 of every step kernel storing its start time "
                       "(s_memrealtime); an entry = start-to-next-start span inside the last graph replay = the launch plus the gap behind it",
            },
            "per_kernel": per_kernel,
        })
        return top

    def cpu_baseline_and_parity(self, B, precisions, postnets):
        """The oracle (a port of the reference's CPU path) timed on this box's host cores, on the timed inputs
        with injected masks; the same masks then go through the HIP path (outside any timed region) and the
        two are compared - the parity of the timed configuration."""
        torch, args, _lib = self.torch, self.args, self._lib
        from oracle import tacotron_oracle as O

        io = self.inputs(B)
        NF, L = args.frames, args.mem_len
        dims = O.DecoderDims()
        sd = {k: v.detach().cpu() for k, v in self.dec.state_dict().items()}
        pw = {k: v.detach().cpu() for k, v in self.post.state_dict().items()}
        memc = io["mem"].cpu()
        # the GPU box gives one GPU a 16-CPU share: do not oversubscribe it
        cores = max(1, min(16, os.cpu_count() or 1, torch.get_num_threads()))
        torch.set_num_threads(cores)
        tc = args.cpu_frames
        if tc <= 0:
            t1 = time.perf_counter()
            O.decode(sd, dims, memc, max_steps=1, masks=O.synthetic_masks(2, B, 256))
            per = (time.perf_counter() - t1) / 2
            tc = max(4, min(NF, int(20.0 / max(per, 1e-4))))
        masks = O.synthetic_masks(tc, B, 256, seed=123)
        t1 = time.perf_counter()
        with torch.no_grad():
            cy, cs, cw = O.decode(sd, dims, memc, max_steps=tc - 1, masks=masks)
            cpost = O.mel_postnet(cy, pw, 3)
        ct = time.perf_counter() - t1
        cpu = {
            "value": round(B * tc / ct, 1), "unit": "mel-frames/s", "cores": cores, "cpu": cpu_model(), "kind": "port",
            "sample": f"oracle (torch-CPU restatement of the reference path) on B={B}, L={L}, {tc} decode frames + Postnet, {cores} threads",
        }
        # parity of the HIP path on the same inputs and masks (bar: 1e-4 relative with 1e-5 absolute floor; argmax exact)
        dmasks = masks.to(self.dev).contiguous()
        full = torch.zeros(io["NS"], *masks.shape[1:], dtype=torch.uint8, device=self.dev)
        full[:tc] = dmasks

        def rel(a, b):
            return float(((a - b).abs() / (1e-5 / 1e-4 + b.abs())).max())

        parity = {"frames_compared": tc, "batch": B, "tolerance": "max |hip - oracle| / (0.1 + |oracle|) <= 1e-4 (rtol 1e-4, atol 1e-5); argmax(w) exact; "
                  "the bf16 Postnet (BASELINE.json configs[2]: 8 significand bits per operand) is REPORTED, not held to that bar"}
        for precision, postnet in zip(precisions, postnets):
            self.eng.set_precision(precision)
            self.decode(io, _lib.DROPOUT_MASKS, full)
            pprec = {"f32": _lib.POSTNET_F32, "bf16": _lib.POSTNET_BF16, "split_f16": _lib.POSTNET_SPLIT_F16}[postnet]
            yp = self.peng.postnet(io["y"][:, :tc].contiguous(), pprec)
            torch.cuda.synchronize()
            y, s, w = io["y"][:, :tc].cpu(), io["s"][:, :tc].unsqueeze(2).cpu(), io["w"][:, :tc].cpu()
            parity[f"{precision}+postnet_{postnet}"] = {
                "max_rel_y": rel(y, cy), "max_rel_s": rel(s, cs), "max_rel_w": rel(w, cw), "max_rel_y_post": rel(yp.cpu(), cpost),
                "argmax_mismatches": int((w.argmax(-1) != cw.argmax(-1)).sum()), "argmax_rows": int(w.shape[0] * w.shape[1]),
            }
        return cpu, parity


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--mem-len", type=int, default=120)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--postnet", choices=["f32", "bf16", "split_f16"], default="f32")
    ap.add_argument("--precision", choices=["f32", "split_f16"], default="f32",
                    help="arithmetic of the headline leg's LSTM / PreNet / query / projection GEMMs (include/ttsdec.h TTSDEC_PREC_*): "
                         "f32 = the reference's own (default), split_f16 = the library's opt-in fast mode")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="ljspeech",
                    help="model dims: ljspeech = BASELINE.json's config (default); rdh / sandra = the other shipped configs")
    ap.add_argument("--workload", choices=["tacotron", "vits2"], default="tacotron",
                    help="tacotron = the headline decoder path (default); vits2 = the second hot path of BASELINE.json configs[4] "
                         "(TextEncoder on [B, 120] + reverse flow on [B, 192, 600], ModelConfig defaults; --batch defaults to 64)")
    ap.add_argument("--dropout", choices=["philox", "masks"], default="philox",
                    help="PreNet dropout source: philox = drawn on the device (default, SURVEY 8d 'mode 2'); masks = injected keep-masks "
                         "resident in HBM ('mode 1': what the parity runs use)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the parity record")
    ap.add_argument("--no-extra-legs", action="store_true", help="time only the headline leg (no split_f16 / b64_f32 / vits2 sub-records)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="decode frames for the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # Started like the N = 1 case (`python bench.py --gpus N ...`): become the launcher.  Nothing in this
        # process has touched the GPU (torch is not even imported), so starting children is safe; the ranks are
        # fresh interpreters.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start with `python bench.py --gpus N` or torch.distributed.run with N ranks")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    from torch_tts_amd import distributed as D

    dist_info = {"backend": dist.get_backend() if world > 1 else None, "world_size": dist.get_world_size() if world > 1 else 1,
                 "collectives": "one broadcast of the packed weight blob per engine at start-up; none in the step loop"}

    def finish(out):
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()  # (rank 0 may still be in its CPU-baseline leg: nobody tears the group down under it)
            dist.destroy_process_group()

    if args.workload == "vits2":
        res = bench_vits2(args, T, torch, dist, dev, world, rank)
        res["distributed"] = dict(dist_info, collectives="none (every rank builds the same weights from the seed)")
        return finish(res)

    wk = Workload(args, T, torch, D, dev, world, rank)
    B, L, NF = args.batch, args.mem_len, args.frames
    lj = args.config == "ljspeech"

    head = wk.time_leg(dist, B, args.precision, args.postnet, args.steps, args.warmup, args.dropout)
    out = {
        "metric": METRIC, "value": head["value"], "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_TEXT[head["lstm_precision"]] + POSTNET_TEXT[args.postnet],
        "data": "synthetic",
        "config": {
            "workload": f"{args.config} dims, batch={B}/GPU (global {B * world}), L={L}, {NF} decode frames + Postnet({args.postnet}); "
                        "BASELINE.json configs[2] per GPU, configs[3] at 8 GPUs",
            "global_batch": B * world, "mem_len": L, "frames": NF, "dropout": args.dropout, "parallelism": f"utterance-shard x{world}",
            "lstm_precision": head["lstm_precision"],
        },
        "distributed": dist_info,
        "rtf": head["rtf"], "audio_seconds_per_s": round(head["value"] * FRAME_SEC, 1),
        "decode_only_frames_per_s": head["decode_only_frames_per_s"],
        "roofline": wk.roofline(B, args.precision, head["decode_step_ms"]),
    }
    legs = []
    if lj and not args.no_extra_legs:
        if args.precision != "split_f16":
            legs.append(("split_f16", B, "split_f16", "split_f16",
                         "the headline workload in the library's opt-in split-fp16 mode: two fp16 planes per operand (narrower than fp32), Postnet split-fp16"))
        else:
            legs.append(("f32_exact", B, "f32", "f32", "the headline workload on the reference's own arithmetic: exact fp32 for every GEMM, Postnet fp32"))
        if B != 64:
            legs.append(("b64_f32", 64, "f32", "f32", "BASELINE.json configs[1]: batch 64 per GPU, exact fp32"))
    for name, b, pr, pp, what in legs:
        leg = wk.time_leg(dist, b, pr, pp, args.steps, args.warmup, args.dropout)
        rf = wk.roofline(b, pr, leg["decode_step_ms"])
        out[name] = {
            "what": what, "value": leg["value"], "unit": "mel-frames/s", "ms_per_step": leg["ms_per_step"], "steps": args.steps,
            "warmup": args.warmup, "dtype": DTYPE_TEXT[leg["lstm_precision"]] + POSTNET_TEXT[pp], "rtf": leg["rtf"],
            "decode_only_frames_per_s": leg["decode_only_frames_per_s"], "global_batch": leg["global_batch"],
            "roofline": rf,
        }
    if lj and not args.no_extra_legs:
        # BASELINE.json configs[4] on the driver's clock too: the vits2 workload's own step, a short run (its CPU leg stays
        # with `--workload vits2`)
        va = argparse.Namespace(**vars(args))
        va.steps, va.warmup = max(3, min(args.steps, 10)), max(1, min(args.warmup, 2))
        v = bench_vits2(va, T, torch, dist, dev, world, rank, cpu_baseline=False)
        out["vits2"] = {k: v[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "rtf", "text_encoder_ms",
                                           "flow_reverse_ms", "roofline") + tuple(k for k in ("split_f16", "f32_exact") if k in v)}
        out["vits2"]["what"] = "BASELINE.json configs[4]: VITS2 TextEncoder + reverse flow, `bench.py --workload vits2`'s step"

    if lj and not args.no_extra_legs:
        # BASELINE.json configs[2] names the Postnet's bf16 MFMA mode: the Postnet alone in each of its arithmetic modes on the
        # driver's clock (every rank runs it - the timing is rank-local, device events around `steps` passes over this rank's y)
        out["postnet_modes"] = wk.time_postnet_modes(B, max(3, min(args.steps, 10)))

    if rank == 0 and lj and not args.no_cpu_baseline:  # (rank 0's shard and host cores, whatever the world size)
        pairs = [(args.precision, args.postnet)]
        if not args.no_extra_legs:
            for pr in (("f32", "f32"), ("split_f16", "split_f16"), ("f32", "bf16")):
                if pr not in pairs:
                    pairs.append(pr)
        out["cpu_baseline"], out["parity"] = wk.cpu_baseline_and_parity(B, [a for a, _ in pairs], [b for _, b in pairs])
    finish(out)


if __name__ == "__main__":
    main()
