#!/usr/bin/env python3
"""Where does the reverse flow's error against the CPU oracle come from?  Prints, per number of coupling layers,
max abs error, the scale of the reference and the error in units of the 1e-4 / 1e-5 bar - with the library's GEMMs in
split-fp16 (default) and in exact fp32 (module attribute precision = "f32": ttsvits_set_precision)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_tts_amd as T  # noqa: E402
from oracle import vits2_oracle as V  # noqa: E402

torch.manual_seed(0)
for n_flows in (1, 2, 4):
    d = V.Vits2Dims(n_flows=n_flows)
    wts = V.random_vits2_weights(d, seed=6)
    g = torch.Generator().manual_seed(3)
    B, Tn = 4, 600
    z = torch.randn(B, d.inter_channels, Tn, generator=g)
    lens = torch.tensor([600, 411, 87, 2])
    ymask = V.sequence_mask(lens, Tn).unsqueeze(1).float()
    ref = V.flow_reverse(z, ymask, wts, d)
    ref64 = V.flow_reverse(z.double(), ymask.double(), {k: v.double() for k, v in wts.items()}, d).float()
    fl = T.vits2.ResidualCouplingTransformersBlock(d.inter_channels, d.flow_hidden, d.flow_kernel, 1, d.flow_wn_layers, n_flows=n_flows,
                                                   use_transformer_flows=True)
    sd = {k[len("flow."):]: v for k, v in wts.items() if k.startswith("flow.")}
    missing, unexpected = fl.load_state_dict(sd, strict=False)
    fl = fl.cuda().eval()
    fl.precision = os.environ.get("VITS_PRECISION", "split_f16")
    with torch.no_grad():
        out = fl(z.cuda(), ymask.cuda(), reverse=True).cpu()
    def rep(name, a, b):
        err = (a - b).abs()
        bar = err / (1e-5 + 1e-4 * b.abs())
        i = bar.argmax()
        print(f"  n_flows={n_flows} {name:28s} max abs {float(err.max()):.3e}  |ref| max {float(b.abs().max()):.2f}  worst err/bar {float(bar.max()):.2f} (ref there {float(b.flatten()[i]):.4f})")
    rep("hip vs oracle(fp32)", out, ref)
    rep("hip vs oracle(fp64)", out, ref64)
    rep("oracle fp32 vs fp64", ref, ref64)
