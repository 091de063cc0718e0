"""Pins the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerance: 1e-6 abs (the vectors were
bit-identical when generated; the slack covers a different BLAS thread count)."""
import torch

from oracle import tacotron_oracle as O

TOL = 1e-6


def _dims(g):
    d = g["meta"]["small_dims"]
    return O.DecoderDims(d_mel=d["d_mel"], r=d["r"], d_pre=d["d_pre"], d_ctx=d["d_ctx"], h_att=d["h_att"], h_dec=d["h_dec"])


def _close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float((a - b).abs().max())
    assert err <= tol, err


def test_memory_padding_rows_are_zero(golden):
    mem, lens = golden["memory"], golden["lengths"]
    for b, n in enumerate(lens.tolist()):
        assert float(mem[b, n:].abs().max()) == 0.0 if n < mem.shape[1] else True


def test_decode_inference(golden):
    c, m = golden["cases"], golden["meta"]["infer"]
    y, s, w = O.decode(golden["dec"], _dims(golden), golden["memory"], max_steps=m["max_steps"], masks=c["infer/masks"])
    assert y.shape[1] == m["T"] == m["max_steps"] + 1
    _close(y, c["infer/y"])
    _close(s, c["infer/s"])
    _close(w, c["infer/w"])
    assert torch.equal(w.argmax(-1), c["infer/w"].argmax(-1))
    yp = O.mel_postnet(y, golden["post"], golden["meta"]["small_dims"]["postnet_layers"])
    _close(yp, c["infer/y_post"], 2e-6)


def test_end_to_end_ids_to_postnet(golden):
    """Tacotron.forward (tacotron.py:29-56) restated as encoder -> decoder -> postnet, from token ids."""
    c, m = golden["cases"], golden["meta"]["e2e"]
    mem = O.encoder2(golden["ids"], golden["lengths"], golden["enc"])
    _close(mem, golden["memory"], 2e-6)
    y, s, w = O.decode(golden["dec"], _dims(golden), mem, max_steps=m["max_steps"], masks=c["infer/masks"])
    _close(y, c["e2e/y"], 5e-6)
    _close(s, c["e2e/s"], 5e-6)
    _close(w, c["e2e/w"], 5e-6)
    _close(O.mel_postnet(y, golden["post"], golden["meta"]["small_dims"]["postnet_layers"]), c["e2e/y_post"], 5e-6)
    assert float(c["e2e/kl_loss"]) == 0.0


def test_decode_stop_rule_is_batch_global_and_inclusive(golden):
    c, m = golden["cases"], golden["meta"]["stop"]
    y, s, w = O.decode(
        golden["dec"], _dims(golden), golden["memory"], max_steps=m["max_steps"], stop_threshold=m["threshold"], masks=c["infer/masks"]
    )
    assert y.shape[1] == m["T"]
    _close(y, c["stop/y"])
    _close(s, c["stop/s"])
    _close(w, c["stop/w"])
    # the frame that fired the rule is kept; no earlier frame fired it
    assert bool((s[:, -1] < m["threshold"]).any())
    assert not bool((s[:, :-1] < m["threshold"]).any())


def test_decode_max_steps_one(golden):
    c = golden["cases"]
    y, s, w = O.decode(golden["dec"], _dims(golden), golden["memory"], max_steps=1, masks=c["infer/masks"])
    _close(y, c["t2/y"])
    _close(w, c["t2/w"])


def test_decode_teacher_forced(golden):
    c = golden["cases"]
    y, s, w = O.decode(golden["dec"], _dims(golden), golden["memory"], masks=c["teacher/masks"], x=c["teacher/x"], p_no_forcing=None)
    _close(y, c["teacher/y"])
    _close(s, c["teacher/s"])
    _close(w, c["teacher/w"])


def test_decode_teacher_partial_forcing_flags_and_rng_replay(golden):
    c, m = golden["cases"], golden["meta"]["teacher_p"]
    flags = c["teacher_p/flags"].bool().tolist()
    y, s, w = O.decode(golden["dec"], _dims(golden), golden["memory"], masks=c["teacher_p/masks"], x=c["teacher/x"], teacher_flags=flags)
    _close(y, c["teacher_p/y"])
    _close(w, c["teacher_p/w"])
    # same thing drawing from the default generator in the reference's order
    torch.manual_seed(m["seed"])
    y2, _, _ = O.decode(golden["dec"], _dims(golden), golden["memory"], dropout="rng", x=c["teacher/x"], p_no_forcing=m["p_no_forcing"])
    _close(y2, c["teacher_p/y"])


def test_unit_prenet(golden):
    c = golden["cases"]
    _close(O.prenet(c["unit/prenet_x"], golden["dec"], c["unit/prenet_masks"]), c["unit/prenet_out"])


def test_unit_lstm_cells(golden):
    c = golden["cases"]
    for name, pref in (("lstm1", "decoder_cell.attention_rnn"), ("lstm2", "decoder_cell.decoder_rnn")):
        h, cc = O.lstm_zoneout_cell(c[f"unit/{name}_x"], c[f"unit/{name}_h"], c[f"unit/{name}_c"], golden["dec"], pref, 0.1)
        _close(h, c[f"unit/{name}_ho"])
        _close(cc, c[f"unit/{name}_co"])


def test_unit_attention(golden):
    c = golden["cases"]
    w = O.stepwise_monotonic_attention(c["unit/att_h"], c["unit/att_w"], golden["memory"], golden["dec"])
    _close(w, c["unit/att_wo"])
    # mass conservation: column L-1 absorbs (p0(1e4) == 1.0 exactly in fp32)
    _close(w.sum(1), c["unit/att_w"].sum(1), 1e-6)


def test_unit_postnet(golden):
    c = golden["cases"]
    _close(O.mel_postnet(c["unit/post_y"], golden["post"], 3), c["unit/post_out"], 2e-6)


def test_isru_sigmoid_saturates_exactly():
    assert float(O.isru_sigmoid(torch.tensor([1e4]))) == 1.0
    assert float(O.isru_sigmoid(torch.tensor([0.0]))) == 0.5


def test_attention_properties_long_run():
    """Properties from attention.py:117-123: rows sum to 1, support after k
    steps within [0, k], expected position non-decreasing."""
    dims = O.DecoderDims(d_mel=8, d_pre=16, d_ctx=32, h_att=32, h_dec=32)
    wts = O.random_decoder_weights(dims, seed=3)
    mem = O.synthetic_memory(2, 12, 32, lengths=[12, 7])
    T = 40
    y, s, w = O.decode(wts, dims, mem, max_steps=T - 1, masks=O.synthetic_masks(T, 2, 16))
    assert w.shape == (2, T, 12)
    assert float((w.sum(-1) - 1).abs().max()) < 1e-5
    pos = (w * torch.arange(12.0)).sum(-1)
    assert bool((pos[:, 1:] >= pos[:, :-1] - 1e-6).all())
    for k in range(5):
        assert float(w[:, k, k + 2 :].abs().max()) == 0.0


# ---- Taco2DecoderCell (r = 2) and MelPostnet2 ----
class _LayerMasks:
    """masks[step][layer]: the two PreNet layers of this cell have different widths (128, d_pre)."""

    def __init__(self, m0, m1):
        self.m0, self.m1 = m0, m1

    def __getitem__(self, t):
        return [self.m0[t], self.m1[t]]


def _taco2_dims(g):
    d = g["meta"]["dims"]
    return O.DecoderDims(d_mel=d["d_mel"], r=d["r"], d_pre=d["d_pre"], d_ctx=d["d_ctx"], h_att=d["h_att"], h_dec=d["h_dec"])


def test_taco2_cell_decode_and_postnet2(golden_taco2):
    g, c = golden_taco2, golden_taco2["cases"]
    dims = _taco2_dims(g)
    y, s, w = O.taco2_decode(g["dec"], dims, g["memory"], max_steps=8, masks=_LayerMasks(c["infer/m0"], c["infer/m1"]))
    assert y.shape == c["infer/y"].shape  # 9 steps x r=2 frames
    _close(y, c["infer/y"])
    _close(s, c["infer/s"])
    _close(w, c["infer/w"])
    _close(O.mel_postnet2(y, g["post"], 2), c["infer/y_post"], 2e-6)
    y2, s2, w2 = O.taco2_decode(g["dec"], dims, g["memory"], masks=_LayerMasks(c["teacher/m0"], c["teacher/m1"]), x=c["teacher/x"])
    _close(y2, c["teacher/y"])
    _close(w2, c["teacher/w"])
    _close(O.mel_postnet2(c["unit/post_y"], g["post"], 2), c["unit/post_out"], 2e-6)


def test_conv1d_fix_pairs_flat_weight_with_rolled_copies():
    """mps_fixes.py:22-29 is NOT torch's conv1d on the same weight tensor: column n*C_in + c of the
    flat weight meets x[c, t + pad - n].  It equals conv1d with the re-indexed, tap-flipped weight."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 6, 11, generator=g)
    w = torch.randn(4, 6, 5, generator=g)
    ours = O.conv1d_fix(x, w, 2)
    w_eff = w.reshape(4, 5, 6).permute(0, 2, 1).flip(2)  # [o, n, c] -> [o, c, tap = 4 - n]
    ref = torch.nn.functional.conv1d(x, w_eff, padding=2)
    assert float((ours - ref).abs().max()) < 1e-5
    assert float((ours - torch.nn.functional.conv1d(x, w, padding=2)).abs().max()) > 1e-2


# ---- Encoder2 ----
def test_encoder2_oracle_matches_reference_vectors():
    import os

    import numpy as np

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_small.npz"))
    wts = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w/")}
    for suf in ("", "2"):
        ids, lens, mem = (torch.from_numpy(g[n + suf]) for n in ("ids", "lengths", "memory"))
        out = O.encoder2(ids, lens, wts)
        assert out.shape == mem.shape
        _close(out, mem, 1e-6)
        for b, n in enumerate(lens.tolist()):
            assert float(out[b, n:].abs().max()) == 0.0 if n < out.shape[1] else True
