"""Reference outputs at FULL dims (tests/golden/fulldims.npz, produced by running the reference itself:
tests/golden/make_golden_fulldims.py).  No weights are stored: the reference modules there and the drop-in modules here are
filled by the same key-seeded recipe (tests/recipes.py).  CPU: the oracles reproduce the reference's outputs.  GPU: the HIP
path is held to north_star's bar (1e-4 relative, 1e-5 absolute) against the reference's own fp32 outputs - the VITS2 reverse
flow at ModelConfig defaults included (round 2 held it to the oracle evaluated in fp64 only), the 5-layer Postnet, and the
LJSpeech-size decoder under the reference's own RNG stream."""
import json
import os

import numpy as np
import pytest
import torch

import torch_tts_amd as T
from oracle import tacotron_oracle as O
from oracle import vits2_oracle as V
from recipes import fill_by_key, seeded_ids, seeded_randn

HERE = os.path.dirname(os.path.abspath(__file__))
RTOL, ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def fd():
    z = np.load(os.path.join(HERE, "golden", "fulldims.npz"))
    return {"c": {k: torch.from_numpy(z[k]) for k in z.files}, "m": json.load(open(os.path.join(HERE, "golden", "fulldims_meta.json")))}


def _close(a, b, what, rtol=RTOL, atol=ATOL):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bool(bad.any()), f"{what}: max abs err {err.max().item():.3e} (ref max {b.abs().max().item():.3e}), {int(bad.sum())} of {bad.numel()} out"


def _vits_modules(m):
    d = m["vits2"]["dims"]
    te = T.vits2.TextEncoder(d["n_vocab"], d["inter_channels"], d["hidden_channels"], d["filter_channels"], d["n_heads"], d["n_layers"], d["kernel_size"], 0.1)
    fl = T.vits2.ResidualCouplingTransformersBlock(d["inter_channels"], d["flow_hidden"], d["flow_kernel"], 1, d["flow_wn_layers"], n_flows=d["n_flows"],
                                                   use_transformer_flows=True)
    # the reference block also owns `post_transformer` parameters it never uses (models.py:513-515): the drop-in's state dict
    # holds the keys it shares with the reference, filled identically (the key seeds the values, not the order)
    assert len(te.state_dict()) == m["vits2"]["state_dict_keys"]["text_encoder"]
    fill_by_key(te, m["vits2"]["text_encoder"]["weights_seed"])
    fill_by_key(fl, m["vits2"]["flow"]["weights_seed"], gain=m["vits2"]["flow"]["weights_gain"])
    return te.eval(), fl.eval()


def _vits_inputs(m):
    d = m["vits2"]["dims"]
    ids = seeded_ids(m["vits2"]["text_encoder"]["ids_seed"], d["n_vocab"], 2, 120)
    lens = torch.tensor(m["vits2"]["text_encoder"]["lengths"])
    z = seeded_randn(m["vits2"]["flow"]["z_seed"], 2, d["inter_channels"], m["vits2"]["flow"]["T"])
    ylen = torch.tensor(m["vits2"]["flow"]["lengths"])
    ymask = V.sequence_mask(ylen, m["vits2"]["flow"]["T"]).unsqueeze(1).float()
    return ids, lens, z, ymask


def _postnet(mm):
    pn = T.MelPostnet(mm["dim_mel"], dim_hidden=mm["dim_hidden"], kernel_size=mm["kernel_size"], num_layers=mm["num_layers"])
    fill_by_key(pn, mm["weights_seed"])
    return pn.eval(), seeded_randn(mm["y_seed"], *mm["shape"])


def _decoder(mm):
    cell = T.Taco2ProdDecoderCell(512, 80, 1, [1024, 1024], dim_pre=256, dim_att=1024)
    dec = T.Decoder(cell, 1, 80)
    assert len(dec.state_dict()) == mm["state_dict_keys"]
    fill_by_key(dec, mm["weights_seed"])
    L = max(mm["lengths"])
    mem = torch.tanh(seeded_randn(mm["memory_seed"], len(mm["lengths"]), L, 512))
    for b, n in enumerate(mm["lengths"]):
        mem[b, n:] = 0.0
    return dec.eval(), mem


# ------------------------------------------------------------------ CPU: the oracles against the reference's outputs
def test_vits2_oracle_reproduces_the_reference_at_default_dims(fd):
    c, m = fd["c"], fd["m"]
    te, fl = _vits_modules(m)
    ids, lens, z, ymask = _vits_inputs(m)
    d = V.Vits2Dims()
    wts = {"enc_p." + k: v for k, v in te.state_dict().items()}
    wts.update({"flow." + k: v for k, v in fl.state_dict().items()})
    with torch.no_grad():
        x, mm, logs, _ = V.text_encoder(ids, lens, wts, d)
        out = V.flow_reverse(z, ymask, wts, d)
    _close(x, c["vits_te/x"], "TextEncoder x", 1e-5, 1e-5)
    _close(mm, c["vits_te/m"], "TextEncoder m", 1e-5, 1e-5)
    _close(logs, c["vits_te/logs"], "TextEncoder logs", 1e-5, 1e-5)
    _close(out, c["vits_flow/out"], "flow reverse")  # (two fp32 evaluations of four coupling layers: each carries ~0.3 of this bar)


def test_postnet_oracle_reproduces_the_reference_at_5_layers_and_at_1_layer_k3(fd):
    for name in ("post5", "post1k3"):
        mm = fd["m"][name]
        pn, y = _postnet(mm)
        with torch.no_grad():
            out = O.mel_postnet(y, {k: v for k, v in pn.state_dict().items()}, mm["num_layers"])
        _close(out, fd["c"][name + "/out"], name, 1e-5, 1e-5)


def test_decoder_oracle_reproduces_the_reference_at_ljspeech_dims(fd):
    mm = fd["m"]["decoder"]
    dec, mem = _decoder(mm)
    torch.manual_seed(mm["rng_seed"])
    with torch.no_grad():
        y, s, w = O.decode({k: v for k, v in dec.state_dict().items()}, O.DecoderDims(), mem, max_steps=mm["max_steps"], dropout="rng")
    _close(y, fd["c"]["dec/y"], "y", 1e-5, 1e-6)
    _close(s, fd["c"]["dec/s"], "s", 1e-5, 1e-6)
    _close(w, fd["c"]["dec/w"], "w", 1e-5, 1e-6)


# ------------------------------------------------------------------ GPU: the HIP path against the reference's outputs
@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["split_f16", "f32"])
def test_vits2_hip_vs_reference_outputs_at_default_dims(fd, prec):
    c, m = fd["c"], fd["m"]
    te, fl = _vits_modules(m)
    ids, lens, z, ymask = _vits_inputs(m)
    te, fl = te.cuda(), fl.cuda()
    te.precision = fl.precision = prec  # (ttsvits_set_precision)
    with torch.no_grad():
        x, mm, logs, _ = te(ids.cuda(), lens.cuda())
        out = fl(z.cuda(), ymask.cuda(), reverse=True)
    _close(x, c["vits_te/x"], "TextEncoder x")
    _close(mm, c["vits_te/m"], "TextEncoder m")
    _close(logs, c["vits_te/logs"], "TextEncoder logs")
    ref = c["vits_flow/out"]
    err = (out.cpu() - ref).abs()
    assert fl._engines.get(fl._dims(), z.cuda().device).precision() == prec
    print(f"{prec}: flow reverse vs the reference's fp32 output: max abs err {err.max().item():.3e}, max err / (1e-5 + 1e-4 |ref|) = {(err / (ATOL + RTOL * ref.abs())).max().item():.3f}")
    _close(out, ref, "flow reverse (4 coupling layers, 600 frames) vs the reference's own fp32 output")


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_postnet_hip_vs_reference_outputs_5_layers_and_1_layer_k3(fd, prec):
    for name in ("post5", "post1k3"):
        pn, y = _postnet(fd["m"][name])
        pn = pn.cuda()
        pn.precision = prec
        with torch.no_grad():
            out = pn(y.cuda())
        _close(out, fd["c"][name + "/out"], f"{name} {prec}")


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_decoder_hip_vs_reference_outputs_at_ljspeech_dims(fd, prec):
    mm = fd["m"]["decoder"]
    dec, mem = _decoder(mm)
    dec = dec.cuda()
    dec.precision = prec
    assert dec.dropout_source == "reference_rng"  # (the default: replays the reference's draws from the seeded CPU generator)
    torch.manual_seed(mm["rng_seed"])
    with torch.no_grad():
        y, s, w = dec(mem.cuda(), None, None, max_steps=mm["max_steps"])
    _close(y, fd["c"]["dec/y"], "y")
    _close(s, fd["c"]["dec/s"], "s")
    _close(w, fd["c"]["dec/w"], "w")
    assert torch.equal(w.cpu().argmax(-1), fd["c"]["dec/w"].argmax(-1))
