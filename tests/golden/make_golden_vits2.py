#!/usr/bin/env python3
"""Golden vectors for the VITS2 path, produced by importing the reference's
own modules (vits2/attentions.py, modules.py, commons.py, models.py) on CPU.  The vectors are first composed step
by step from the reference blocks following models.py:369-380, 506-531 and 803-810 (that also gives intermediate
vectors), and then reproduced bit for bit by models.TextEncoder / models.ResidualCouplingTransformersBlock
themselves: models.py is importable once monotonic_align/core.pyx - the reference's own source - is compiled into a
scratch directory (done below; Cython is in the image).

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



# ---- the same through vits2/models.py ITSELF ----
# models.py:12 imports monotonic_align, whose __init__ wants the compiled extension monotonic_align.monotonic_align.core.
# Build exactly that from the reference's own core.pyx (Cython is in the image) into a scratch directory - nothing is
# written under /root/reference, nothing of the reference is copied into this repository - and register it under the
# name the package asks for.  Then models.TextEncoder and models.ResidualCouplingTransformersBlock are instantiated,
# given the weights above, and must reproduce the composed vectors bit for bit: the te/* and flow/* arrays of the
# fixture ARE outputs of models.py.
def import_reference_models():
    import importlib.machinery
    import importlib.util
    import subprocess
    import sysconfig
    import tempfile
    import types

    scratch = tempfile.mkdtemp(prefix="vits2_ma_")
    c_file = os.path.join(scratch, "core.c")
    subprocess.run([sys.executable, "-m", "cython", "-3", "-o", c_file, os.path.join(REF, "monotonic_align", "core.pyx")], check=True)
    so = os.path.join(scratch, "core" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(), c_file, "-o", so],
                   check=True)
    sub = types.ModuleType("monotonic_align.monotonic_align")  # (the directory setup.py build_ext --inplace would make)
    sub.__path__ = [scratch]
    sys.modules["monotonic_align.monotonic_align"] = sub
    loader = importlib.machinery.ExtensionFileLoader("monotonic_align.monotonic_align.core", so)
    spec = importlib.util.spec_from_loader("monotonic_align.monotonic_align.core", loader)
    core = importlib.util.module_from_spec(spec)
    loader.exec_module(core)
    sys.modules["monotonic_align.monotonic_align.core"] = core
    import models  # the reference's vits2/models.py

    return models


models = import_reference_models()
with torch.no_grad():
    te_ref = models.TextEncoder(D["n_vocab"], I, H, D["filter_channels"], D["n_heads"], D["n_layers"], D["kernel_size"], 0.1).eval()
    te_ref.load_state_dict({k[len("w/enc_p."):]: torch.from_numpy(v) for k, v in out.items() if k.startswith("w/enc_p.")})
    rx, rm, rlogs, rmask = te_ref(ids, lengths)                                   # models.py:369-380
    fl_ref = models.ResidualCouplingTransformersBlock(I, Fh, D["flow_kernel"], 1, D["flow_wn_layers"], n_flows=D["n_flows"],
                                                      use_transformer_flows=True, transformer_flow_type="pre_conv").eval()
    missing, unexpected = fl_ref.load_state_dict({k[len("w/flow."):]: torch.from_numpy(v) for k, v in out.items() if k.startswith("w/flow.")},
                                                 strict=False)
    assert not unexpected and all("post_transformer" in k for k in missing), (missing, unexpected)  # (unused by forward, models.py:513-515)
    rflow = fl_ref(z, y_mask, reverse=True)                                        # models.py:803-810
via_models = {
    "te/x": float((rx - xe).abs().max()), "te/m": float((rm - m).abs().max()), "te/logs": float((rlogs - logs).abs().max()),
    "flow/out": float((rflow - xx).abs().max()),
}
print("models.py vs the composition (max abs diff):", via_models)
assert all(v == 0.0 for v in via_models.values()), via_models
assert torch.equal(rmask, x_mask)

# ---- speaker conditioning (gin_channels > 0): models.TextEncoder / ResidualCouplingTransformersBlock with g [B, gin, 1] ----
# (placed after everything above so that the earlier arrays keep their RNG draws; a third encoder layer because
# attentions.py:47-52 conditions at layer 2 and asserts 2 < n_layers)
DG = dict(D, n_layers=3, gin_channels=8, cond_layer_idx=2)
with torch.no_grad():
    te_g = models.TextEncoder(DG["n_vocab"], I, H, DG["filter_channels"], DG["n_heads"], DG["n_layers"], DG["kernel_size"], 0.1,
                              gin_channels=DG["gin_channels"]).eval()
    randomize(te_g.encoder); randomize(te_g.proj)
    te_g.encoder.spk_emb_linear.weight.normal_(0.0, 0.5)
    fl_g = models.ResidualCouplingTransformersBlock(I, Fh, DG["flow_kernel"], 1, DG["flow_wn_layers"], n_flows=DG["n_flows"],
                                                    gin_channels=DG["gin_channels"], use_transformer_flows=True,
                                                    transformer_flow_type="pre_conv").eval()
    randomize(fl_g)
    for k, v in te_g.state_dict().items():
        out[f"wg/enc_p.{k}"] = v.detach().numpy().copy()
    for k, v in fl_g.state_dict().items():
        if "post_transformer" not in k:
            out[f"wg/flow.{k}"] = v.detach().numpy().copy()
    g_spk = torch.randn(B, DG["gin_channels"], 1)
    gx, gm, glogs, _ = te_g(ids, lengths, g=g_spk)               # models.py:369-380 with attentions.py:80-84
    gx0, _, _, _ = te_g(ids, lengths, g=None)
    gflow = fl_g(z, y_mask, g=g_spk, reverse=True)                # models.py:803-810 with modules.py:189-199
    gflow0 = fl_g(z, y_mask, g=None, reverse=True)
    assert float((gx - gx0).abs().max()) > 1e-2 and float((gflow - gflow0).abs().max()) > 1e-2  # (g matters in both)
out["g/spk"] = g_spk.numpy()
out["g/te/x"] = gx.numpy(); out["g/te/m"] = gm.numpy(); out["g/te/logs"] = glogs.numpy(); out["g/te/x_no_g"] = gx0.numpy()
out["g/flow/out"] = gflow.numpy(); out["g/flow/out_no_g"] = gflow0.numpy()

np.savez_compressed(os.path.join(HERE, "vits2_small.npz"), **out)
json.dump({"dims": D, "dims_g": DG, "reference": "kgoba/torch-tts @ 2024_10_08, vits2/{attentions,modules,commons,models}.py imported on CPU",
           "glue": "te/* and flow/* equal, bit for bit, the outputs of models.TextEncoder.forward and "
                   "models.ResidualCouplingTransformersBlock.forward(reverse=True) themselves (vits2/models.py imported with "
                   "monotonic_align/core.pyx compiled into a scratch directory); the step-by-step composition in "
                   "make_golden_vits2.py is kept because it also yields the intermediate vectors (flow/wn_out_first, unit/*)",
           "models_py_vs_composition_maxabs": via_models,
           "torch": torch.__version__}, open(os.path.join(HERE, "vits2_meta.json"), "w"), indent=1)
print("wrote", len(out), "arrays")
