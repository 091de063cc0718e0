"""Pins oracle/vits2_oracle.py to vectors produced by the reference's own vits2 building blocks
(tests/golden/make_golden_vits2.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vits2_oracle as V

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gv():
    z = np.load(os.path.join(HERE, "golden", "vits2_small.npz"))
    meta = json.load(open(os.path.join(HERE, "golden", "vits2_meta.json")))
    wts = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}
    cases = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("w/")}
    return {"wts": wts, "c": cases, "dims": V.Vits2Dims(**meta["dims"]), "meta": meta}


def close(a, b, tol=2e-6):
    assert a.shape == b.shape
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), err


def test_text_encoder_matches_reference_blocks(gv):
    c, d = gv["c"], gv["dims"]
    x, m, logs, mask = V.text_encoder(c["te/ids"], c["te/lengths"], gv["wts"], d)
    close(x, c["te/x"]); close(m, c["te/m"]); close(logs, c["te/logs"])
    assert mask.shape == (3, 1, 13) and mask.sum().item() == 21
    assert float(x[2, :, 1:].abs().max()) == 0.0  # padded frames are exactly zero


def test_units_match_reference_blocks(gv):
    c, d, w = gv["c"], gv["dims"], gv["wts"]
    mask = V.sequence_mask(c["te/lengths"], 13).unsqueeze(1).float()
    am = mask.unsqueeze(2) * mask.unsqueeze(-1)
    close(V.mha(c["unit/x"], am, w, "enc_p.encoder.attn_layers.0", d.n_heads, d.window_size), c["unit/mha_win"])
    close(V.ffn(c["unit/x"], mask, w, "enc_p.encoder.ffn_layers.0", d.kernel_size), c["unit/ffn"])
    close(V.layer_norm_c(c["unit/x"], w["enc_p.encoder.norm_layers_1.0.gamma"], w["enc_p.encoder.norm_layers_1.0.beta"]), c["unit/ln"])
    s = c["unit/gate_a"] + c["unit/gate_b"]
    close(torch.tanh(s[:, :8]) * torch.sigmoid(s[:, 8:]), c["unit/gate"])
    ymask = V.sequence_mask(c["flow/lengths"], 17).unsqueeze(1).float()
    close(V.wn(c["unit/wn_in"] * ymask, ymask, w, "flow.flows.0.enc", d.flow_wn_layers, d.flow_kernel), c["unit/wn_out"])
    close(V.encoder_stack(c["unit/enc_nowin_in"] * ymask, ymask, w, "flow.flows.0.pre_transformer", d.flow_tf_layers, d.flow_tf_heads,
                          None, d.flow_tf_kernel), c["unit/enc_nowin_out"])


def test_flow_reverse_matches_reference_blocks(gv):
    c, d = gv["c"], gv["dims"]
    ymask = V.sequence_mask(c["flow/lengths"], 17).unsqueeze(1).float()
    out = V.flow_reverse(c["flow/z"], ymask, gv["wts"], d)
    close(out, c["flow/out"], 5e-6)
    # the layer applied last (flows.0) leaves its x0 half untouched: out[:, :half] is an input-only function of the
    # previous layers, and the padded frames of the transformed half are exactly zero
    half = d.inter_channels // 2
    assert float(out[2, half:, 2:].abs().max()) == 0.0


def test_random_weights_cover_every_key_the_functions_read():
    d = V.Vits2Dims(n_vocab=11, inter_channels=8, hidden_channels=16, filter_channels=24, n_layers=1, flow_hidden=8, flow_wn_layers=2, n_flows=2)
    w = V.random_vits2_weights(d, seed=3)
    ids = torch.randint(0, 11, (2, 9)); lens = torch.tensor([9, 4])
    x, m, logs, mask = V.text_encoder(ids, lens, w, d)
    assert x.shape == (2, 16, 9) and m.shape == (2, 8, 9)
    z = torch.randn(2, 8, 12)
    out = V.flow_reverse(z, V.sequence_mask(torch.tensor([12, 5]), 12).unsqueeze(1).float(), w, d)
    assert out.shape == z.shape and bool(torch.isfinite(out).all())


def test_sequences_shorter_than_the_window(gv):
    """T < window + 1: the reference slices its relative-position tables (attentions.py:304-318)."""
    c, d = gv["c"], gv["dims"]
    for Ts in (1, 3):
        x, _, _, _ = V.text_encoder(c[f"short{Ts}/ids"], c[f"short{Ts}/lengths"], gv["wts"], d)
        close(x, c[f"short{Ts}/x"])


def test_speaker_conditioning_matches_models_py(gv):
    """gin_channels > 0: g [B, gin, 1] through WN.cond_layer (modules.py:189-199) and the text encoder's spk_emb_linear
    (attentions.py:80-84); the vectors are outputs of models.TextEncoder / ResidualCouplingTransformersBlock themselves."""
    z = np.load(os.path.join(HERE, "golden", "vits2_small.npz"))
    wg = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("wg/")}
    c, dg = gv["c"], V.Vits2Dims(**gv["meta"]["dims_g"])
    assert dg.gin_channels == 8 and dg.cond_layer_idx == 2 and dg.n_layers == 3
    g = c["g/spk"]
    x, m, logs, _ = V.text_encoder(c["te/ids"], c["te/lengths"], wg, dg, g=g)
    close(x, c["g/te/x"], 5e-6); close(m, c["g/te/m"], 5e-6); close(logs, c["g/te/logs"], 5e-6)
    x0, _, _, _ = V.text_encoder(c["te/ids"], c["te/lengths"], wg, dg, g=None)
    close(x0, c["g/te/x_no_g"], 5e-6)
    ymask = V.sequence_mask(c["flow/lengths"], 17).unsqueeze(1).float()
    close(V.flow_reverse(c["flow/z"], ymask, wg, dg, g=g), c["g/flow/out"], 5e-6)
    close(V.flow_reverse(c["flow/z"], ymask, wg, dg, g=None), c["g/flow/out_no_g"], 5e-6)
    assert float((c["g/flow/out"] - c["g/flow/out_no_g"]).abs().max()) > 1e-2  # (the conditioning is not a no-op in the fixture)
    # random weights cover the conditioned keys too
    d2 = V.Vits2Dims(n_vocab=11, inter_channels=8, hidden_channels=16, filter_channels=24, n_layers=3, flow_hidden=8, flow_wn_layers=2, n_flows=2,
                     gin_channels=4)
    w2 = V.random_vits2_weights(d2, seed=3)
    out = V.flow_reverse(torch.randn(2, 8, 12), V.sequence_mask(torch.tensor([12, 5]), 12).unsqueeze(1).float(), w2, d2, g=torch.randn(2, 4, 1))
    assert bool(torch.isfinite(out).all())
    xx, _, _, _ = V.text_encoder(torch.randint(0, 11, (2, 9)), torch.tensor([9, 4]), w2, d2, g=torch.randn(2, 4, 1))
    assert bool(torch.isfinite(xx).all())
