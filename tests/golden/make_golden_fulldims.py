#!/usr/bin/env python3
"""Reference outputs at FULL dims, without committing weights: the reference's own modules are built at
ModelConfig / LJSpeech defaults, their parameters filled by the key-seeded recipe of tests/recipes.py (the drop-in modules are
filled the same way on the GPU box), and only seeded-input recipes + the reference's outputs are stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fulldims.py

  vits2 (vits2/models.py, imported with monotonic_align/core.pyx compiled into a scratch directory as in make_golden_vits2.py):
      TextEncoder.forward at ModelConfig defaults (vits2/cli.py:159-180), B = 2, 120 tokens, ragged;
      ResidualCouplingTransformersBlock.forward(reverse=True), 4 flows, B = 2, 600 frames, ragged.
  tacotron (tacotron/modules/modules.py, decoder.py, decoder_cell.py):
      MelPostnet(num_layers = 5) - north_star's "5-layer Postnet" - and MelPostnet(num_layers = 1, kernel_size = 3), reduced width;
      Decoder(Taco2ProdDecoderCell) at config-ljspeech.yaml dims, B = 3, L = 41 (ragged), 24 frames, dropout replayed from the seed.
Writes tests/golden/fulldims.npz + fulldims_meta.json (data only)."""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from recipes import fill_by_key, seeded_ids, seeded_randn  # noqa: E402

torch.set_num_threads(8)
out, meta = {}, {"torch": torch.__version__, "reference": "kgoba/torch-tts @ 2024_10_08"}

# ---------------------------------------------------------------- VITS2
spec = importlib.util.spec_from_file_location("mgv", os.path.join(HERE, "make_golden_vits2.py"))
# (only its import helper is wanted: load the source, cut at the helper, exec that part)
src = open(os.path.join(HERE, "make_golden_vits2.py")).read()
ns = {"__name__": "mgv_helper"}
head = src[: src.index("HERE = os.path.dirname")]
helper = src[src.index("def import_reference_models():") : src.index("models = import_reference_models()")]
exec(compile(head + "\nREF = '/root/reference/vits2'\n" + helper, "make_golden_vits2_helper", "exec"), ns)
models = ns["import_reference_models"]()
import commons  # noqa: E402  (the reference's, on sys.path through the helper's header)

VD = dict(n_vocab=178, inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3, flow_hidden=192,
          flow_kernel=5, flow_wn_layers=4, n_flows=4)
with torch.no_grad():
    te = models.TextEncoder(VD["n_vocab"], VD["inter_channels"], VD["hidden_channels"], VD["filter_channels"], VD["n_heads"], VD["n_layers"],
                            VD["kernel_size"], 0.1).eval()
    fill_by_key(te, 11)
    ids = seeded_ids(12, VD["n_vocab"], 2, 120)
    lens = torch.tensor([120, 77])
    x, m, logs, mask = te(ids, lens)
    out["vits_te/x"], out["vits_te/m"], out["vits_te/logs"] = x.numpy(), m.numpy(), logs.numpy()
    fl = models.ResidualCouplingTransformersBlock(VD["inter_channels"], VD["flow_hidden"], VD["flow_kernel"], 1, VD["flow_wn_layers"],
                                                  n_flows=VD["n_flows"], use_transformer_flows=True, transformer_flow_type="pre_conv").eval()
    fill_by_key(fl, 13, gain=0.8)
    z = seeded_randn(14, 2, VD["inter_channels"], 600)
    ylen = torch.tensor([600, 411])
    ymask = torch.unsqueeze(commons.sequence_mask(ylen, 600), 1).to(torch.float32)
    out["vits_flow/out"] = fl(z, ymask, reverse=True).numpy()
meta["vits2"] = {"dims": VD, "text_encoder": {"weights_seed": 11, "ids_seed": 12, "lengths": [120, 77]},
                 "flow": {"weights_seed": 13, "weights_gain": 0.8, "z_seed": 14, "lengths": [600, 411], "T": 600},
                 "state_dict_keys": {"text_encoder": len(te.state_dict()), "flow": len(fl.state_dict())}}

# ---------------------------------------------------------------- Tacotron
for k in [k for k in sys.modules if k in ("modules", "commons", "attentions", "models")]:
    del sys.modules[k]  # (the vits2 package's top-level names shadow tacotron/modules)
sys.path = [p for p in sys.path if not p.startswith("/root/reference/vits2")]
sys.path.insert(0, "/root/reference/tacotron")
import decoder as ref_decoder  # noqa: E402
import decoder_cell as ref_cell  # noqa: E402
from modules.modules import MelPostnet as RefMelPostnet  # noqa: E402

with torch.no_grad():
    for name, (hidden, k, layers, seed) in {"post5": (64, 5, 5, 21), "post1k3": (48, 3, 1, 22)}.items():
        pn = RefMelPostnet(80, dim_hidden=hidden, kernel_size=k, num_layers=layers).eval()
        fill_by_key(pn, seed)
        y = seeded_randn(seed + 100, 3, 29, 80)
        out[f"{name}/out"] = pn(y).numpy()
        meta[name] = {"dim_mel": 80, "dim_hidden": hidden, "kernel_size": k, "num_layers": layers, "weights_seed": seed, "y_seed": seed + 100,
                      "shape": [3, 29, 80]}
    # the LJSpeech decoder at full dims (configs/config-ljspeech.yaml:47-69): dim_ctx 512, mel 80, r 1, rnn [1024, 1024], pre 256
    cell = ref_cell.Taco2ProdDecoderCell(512, 80, 1, [1024, 1024], dim_pre=256, dim_att=1024)
    dec = ref_decoder.Decoder(cell, 1, 80).eval()
    fill_by_key(dec, 31)
    B, L, T = 3, 41, 24
    lengths = [41, 30, 9]
    mem = torch.tanh(seeded_randn(32, B, L, 512))
    for b, n in enumerate(lengths):
        mem[b, n:] = 0.0  # padded encoder rows are exactly zero
    torch.manual_seed(33)  # the always-on PreNet dropout draws from the default CPU generator (modules.py:40)
    y, s, w = dec(mem, None, None, max_steps=T - 1)
    out["dec/y"], out["dec/s"], out["dec/w"] = y.numpy(), s.numpy(), w.numpy()
    meta["decoder"] = {"weights_seed": 31, "memory_seed": 32, "lengths": lengths, "rng_seed": 33, "max_steps": T - 1, "T": int(y.shape[1]),
                       "state_dict_keys": len(dec.state_dict())}

np.savez_compressed(os.path.join(HERE, "fulldims.npz"), **out)
json.dump(meta, open(os.path.join(HERE, "fulldims_meta.json"), "w"), indent=1)
print({k: v.shape for k, v in out.items()})
print("wrote", os.path.getsize(os.path.join(HERE, "fulldims.npz")), "bytes")
