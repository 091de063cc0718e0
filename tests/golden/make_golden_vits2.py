#!/usr/bin/env python3
"""Golden vectors for the VITS2 path, produced by importing the reference's
own building blocks (vits2/attentions.py, modules.py, commons.py) on CPU.  vits2/models.py is NOT
imported (it needs the unbuilt monotonic_align extension); the glue of TextEncoder.forward and of the
reverse coupling layer is composed HERE from the reference blocks, following models.py:369-380,
506-531 and 803-810, and is marked as such in the fixture's meta.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_vits2.py

Writes tests/golden/vits2_small.npz + vits2_meta.json (reduced dims; no reference source)."""
import json
import math
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference/vits2"
sys.path.insert(0, REF)
import attentions  # noqa: E402
import commons  # noqa: E402
import modules  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(7)

D = dict(n_vocab=23, inter_channels=16, hidden_channels=32, filter_channels=48, n_heads=2, n_layers=2, kernel_size=3,
         window_size=4, flow_hidden=24, flow_kernel=5, flow_wn_layers=3, n_flows=2, flow_tf_layers=2, flow_tf_heads=2,
         flow_tf_kernel=3)
out = {}


def randomize(mod, scale=0.3):
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if n.endswith("gamma"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif n.endswith("weight_g"):
                p.copy_(0.5 + torch.rand_like(p))
            elif p.dim() == 1:
                p.copy_(0.1 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) * scale)


def save_sd(prefix, mod):
    for k, v in mod.state_dict().items():
        out[f"w/{prefix}.{k}"] = v.detach().numpy().copy()


# ---- text encoder: Embedding + attentions.Encoder(window 4) + proj ----
H, I = D["hidden_channels"], D["inter_channels"]
emb = torch.nn.Embedding(D["n_vocab"], H)
enc = attentions.Encoder(H, D["filter_channels"], D["n_heads"], D["n_layers"], D["kernel_size"], 0.1).eval()
proj = torch.nn.Conv1d(H, 2 * I, 1)
randomize(enc); randomize(proj)
with torch.no_grad():
    emb.weight.normal_(0.0, H**-0.5)
B, T = 3, 13
ids = torch.randint(0, D["n_vocab"], (B, T))
lengths = torch.tensor([13, 7, 1])
with torch.no_grad():
    x = emb(ids) * math.sqrt(H)                       # models.py:370
    x = torch.transpose(x, 1, -1)                     # :371
    x_mask = torch.unsqueeze(commons.sequence_mask(lengths, x.size(2)), 1).to(x.dtype)  # :372-374
    xe = enc(x * x_mask, x_mask)                      # :376
    stats = proj(xe) * x_mask                         # :377
    m, logs = torch.split(stats, I, dim=1)            # :379
save_sd("enc_p.emb", emb); save_sd("enc_p.encoder", enc); save_sd("enc_p.proj", proj)
out["te/ids"] = ids.numpy(); out["te/lengths"] = lengths.numpy()
out["te/x"] = xe.numpy(); out["te/m"] = m.numpy(); out["te/logs"] = logs.numpy()

# ---- unit vectors: one attention layer (with and without window), FFN, LayerNorm, WN, gate ----
with torch.no_grad():
    xin = torch.randn(B, H, T)
    amask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    out["unit/x"] = xin.numpy()
    out["unit/mha_win"] = enc.attn_layers[0](xin, xin, amask).numpy()
    out["unit/ffn"] = enc.ffn_layers[0](xin, x_mask).numpy()
    out["unit/ln"] = enc.norm_layers_1[0](xin).numpy()
    a, b = torch.randn(B, 2 * 8, T), torch.randn(B, 2 * 8, T)
    out["unit/gate_a"] = a.numpy(); out["unit/gate_b"] = b.numpy()
    out["unit/gate"] = commons.fused_add_tanh_sigmoid_multiply(a, b, torch.IntTensor([8])).numpy()

# ---- flow reverse: [layer, Flip] x n_flows, composed from the reference blocks ----
half, Fh = I // 2, D["flow_hidden"]
layers = []
for i in range(D["n_flows"]):
    pre_tf = attentions.Encoder(half, half, n_heads=D["flow_tf_heads"], n_layers=D["flow_tf_layers"], kernel_size=D["flow_tf_kernel"],
                                p_dropout=0.1, window_size=None).eval()
    pre = torch.nn.Conv1d(half, Fh, 1)
    wn = modules.WN(Fh, D["flow_kernel"], 1, D["flow_wn_layers"], p_dropout=0, gin_channels=0).eval()
    post = torch.nn.Conv1d(Fh, half, 1)
    for mod in (pre_tf, pre, wn, post):
        randomize(mod)
    p = f"flow.flows.{2 * i}"
    save_sd(p + ".pre_transformer", pre_tf); save_sd(p + ".pre", pre); save_sd(p + ".enc", wn); save_sd(p + ".post", post)
    layers.append((pre_tf, pre, wn, post))
flip = modules.Flip()
Ty = 17
ylen = torch.tensor([17, 9, 2])
y_mask = torch.unsqueeze(commons.sequence_mask(ylen, Ty), 1).to(torch.float32)
z = torch.randn(B, I, Ty)
with torch.no_grad():
    xx = z
    first = None
    for i in reversed(range(D["n_flows"])):           # models.py:807-809 (reversed(self.flows))
        xx = flip(xx, y_mask, reverse=True)
        pre_tf, pre, wn, post = layers[i]
        x0, x1 = torch.split(xx, [half] * 2, 1)        # models.py:507
        x0_ = pre_tf(x0 * y_mask, y_mask)              # :508
        x0_ = x0_ + x0                                 # :509
        h = pre(x0_) * y_mask                          # :510
        h = wn(h, y_mask)                              # :511
        if first is None:
            out["flow/wn_out_first"] = h.numpy()
        mm = post(h) * y_mask                          # :517 (mean_only: logs = 0)
        x1 = (x1 - mm) * torch.exp(-torch.zeros_like(mm)) * y_mask  # :529
        xx = torch.cat([x0, x1], 1)                    # :530
        first = True
out["flow/z"] = z.numpy(); out["flow/lengths"] = ylen.numpy(); out["flow/out"] = xx.numpy()
# WN alone
with torch.no_grad():
    hh = torch.randn(B, Fh, Ty)
    out["unit/wn_in"] = hh.numpy()
    out["unit/wn_out"] = layers[0][2](hh * y_mask, y_mask).numpy()
# encoder without window alone
with torch.no_grad():
    x0 = torch.randn(B, half, Ty)
    out["unit/enc_nowin_in"] = x0.numpy()
    out["unit/enc_nowin_out"] = layers[0][0](x0 * y_mask, y_mask).numpy()

# sequences shorter than the relative-position window (attentions.py:304-318 slices the tables instead of padding)
with torch.no_grad():
    for Ts in (1, 3):
        ids_s = torch.randint(0, D["n_vocab"], (2, Ts))
        len_s = torch.tensor([Ts, max(1, Ts - 1)])
        xs = torch.transpose(emb(ids_s) * math.sqrt(H), 1, -1)
        ms = torch.unsqueeze(commons.sequence_mask(len_s, Ts), 1).to(xs.dtype)
        out[f"short{Ts}/ids"] = ids_s.numpy(); out[f"short{Ts}/lengths"] = len_s.numpy()
        out[f"short{Ts}/x"] = enc(xs * ms, ms).numpy()

np.savez_compressed(os.path.join(HERE, "vits2_small.npz"), **out)
json.dump({"dims": D, "reference": "kgoba/torch-tts @ 2024_10_08, vits2/{attentions,modules,commons}.py imported on CPU",
           "glue": "TextEncoder.forward and the reverse coupling layer are composed in make_golden_vits2.py from the reference blocks "
                   "(models.py is not importable here: monotonic_align is an unbuilt extension)",
           "torch": torch.__version__}, open(os.path.join(HERE, "vits2_meta.json"), "w"), indent=1)
print("wrote", len(out), "arrays")
