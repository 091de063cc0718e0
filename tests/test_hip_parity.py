"""GPU parity tests: the HIP path (through the C ABI / the drop-in modules) against
(a) golden vectors produced by the reference itself and (b) the CPU oracle on the same
seeded inputs.  Tolerance (BASELINE.json north_star): mel / stop logits within 1e-4
relative (atol 1e-5 for values near zero), attention argmax indices bit-exact."""
import os

import pytest
import torch

from oracle import tacotron_oracle as O

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def H():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import hip_helpers

    return hip_helpers


def _small_dims(g):
    d = g["meta"]["small_dims"]
    return O.DecoderDims(d_mel=d["d_mel"], r=d["r"], d_pre=d["d_pre"], d_ctx=d["d_ctx"], h_att=d["h_att"], h_dec=d["h_dec"])


# --------------------------------------------------------------------------
# golden vectors from the reference (reduced dims, ragged lengths, B=3, L=17)
# --------------------------------------------------------------------------
def test_golden_inference(golden, H):
    c, m = golden["cases"], golden["meta"]["infer"]
    dec = H.make_decoder(_small_dims(golden), golden["dec"])
    y, s, w, fired = H.run_decoder_with_masks(dec, golden["memory"], c["infer/masks"], max_steps=m["max_steps"])
    assert not fired and y.shape[1] == m["T"]
    H.assert_close(y, c["infer/y"], RTOL, ATOL, "y")
    H.assert_close(s, c["infer/s"], RTOL, ATOL, "s")
    H.assert_close(w, c["infer/w"], RTOL, ATOL, "w")
    assert torch.equal(w.argmax(-1), c["infer/w"].argmax(-1))
    d = golden["meta"]["small_dims"]
    pn = H.make_postnet(d["d_mel"], d["postnet_hidden"], d["postnet_layers"], golden["post"])
    with torch.no_grad():
        yp = pn(c["infer/y"].cuda()).cpu()
    H.assert_close(yp, c["infer/y_post"], RTOL, ATOL, "y_post")


def test_golden_stop_rule(golden, H):
    c, m = golden["cases"], golden["meta"]["stop"]
    dec = H.make_decoder(_small_dims(golden), golden["dec"], stop_threshold=m["threshold"])
    y, s, w, fired = H.run_decoder_with_masks(dec, golden["memory"], c["infer/masks"], max_steps=m["max_steps"])
    assert fired
    assert y.shape[1] == m["T"], "stop rule must be batch-global and inclusive of the firing step"
    H.assert_close(y, c["stop/y"], RTOL, ATOL, "y")
    H.assert_close(s, c["stop/s"], RTOL, ATOL, "s")
    H.assert_close(w, c["stop/w"], RTOL, ATOL, "w")


def test_golden_max_steps_one(golden, H):
    c = golden["cases"]
    dec = H.make_decoder(_small_dims(golden), golden["dec"])
    y, s, w, _ = H.run_decoder_with_masks(dec, golden["memory"], c["infer/masks"], max_steps=1)
    assert y.shape[1] == 2
    H.assert_close(y, c["t2/y"], RTOL, ATOL, "y")
    H.assert_close(w, c["t2/w"], RTOL, ATOL, "w")


def test_golden_teacher_forced(golden, H):
    c = golden["cases"]
    dec = H.make_decoder(_small_dims(golden), golden["dec"])
    y, s, w, _ = H.run_decoder_with_masks(dec, golden["memory"], c["teacher/masks"], x=c["teacher/x"])
    H.assert_close(y, c["teacher/y"], RTOL, ATOL, "y")
    H.assert_close(s, c["teacher/s"], RTOL, ATOL, "s")
    H.assert_close(w, c["teacher/w"], RTOL, ATOL, "w")


def test_golden_teacher_partial_forcing(golden, H):
    c = golden["cases"]
    dec = H.make_decoder(_small_dims(golden), golden["dec"])
    y, s, w, _ = H.run_decoder_with_masks(dec, golden["memory"], c["teacher_p/masks"], x=c["teacher/x"], flags=c["teacher_p/flags"])
    H.assert_close(y, c["teacher_p/y"], RTOL, ATOL, "y")
    H.assert_close(w, c["teacher_p/w"], RTOL, ATOL, "w")


def test_golden_through_module_forward_with_reference_rng(golden, H):
    """Decoder.forward with dropout_source='reference_rng' under the reference's seed
    reproduces the reference's own run (masks replayed from torch's CPU generator),
    for inference, the stop rule and teacher forcing with p_no_forcing."""
    c, meta = golden["cases"], golden["meta"]
    dims = _small_dims(golden)
    mem = golden["memory"].cuda()
    dec = H.make_decoder(dims, golden["dec"])
    with torch.no_grad():
        torch.manual_seed(meta["infer"]["seed"])
        y, s, w = dec(mem, None, None, meta["infer"]["max_steps"])
        H.assert_close(y.cpu(), c["infer/y"], RTOL, ATOL, "y")
        assert s.shape == c["infer/s"].shape
        # generator state afterwards equals the reference's (T steps of draws consumed)
        after = torch.rand(1)
        torch.manual_seed(meta["infer"]["seed"])
        for _ in range(meta["infer"]["T"]):
            O.draw_prenet_masks(3, dims.d_pre, dims.d_pre)
        assert torch.equal(after, torch.rand(1))

        dec2 = H.make_decoder(dims, golden["dec"], stop_threshold=meta["stop"]["threshold"])
        torch.manual_seed(meta["stop"]["seed"])
        y2, s2, w2 = dec2(mem, None, None, meta["stop"]["max_steps"])
        assert y2.shape[1] == meta["stop"]["T"]
        H.assert_close(y2.cpu(), c["stop/y"], RTOL, ATOL, "y stop")
        after = torch.rand(1)
        torch.manual_seed(meta["stop"]["seed"])
        for _ in range(meta["stop"]["T"]):
            O.draw_prenet_masks(3, dims.d_pre, dims.d_pre)
        assert torch.equal(after, torch.rand(1)), "early stop must leave the CPU generator where the reference leaves it"

        torch.manual_seed(meta["teacher_p"]["seed"])
        y5, s5, w5 = dec(mem, None, c["teacher/x"].cuda(), 0, p_no_forcing=meta["teacher_p"]["p_no_forcing"])
        H.assert_close(y5.cpu(), c["teacher_p/y"], RTOL, ATOL, "y teacher_p")
        H.assert_close(w5.cpu(), c["teacher_p/w"], RTOL, ATOL, "w teacher_p")


def test_golden_unbounded_decode_chunks(golden, H):
    """max_steps=0 decodes in chunks until the stop rule fires; chunking must not change results."""
    c, m = golden["cases"], golden["meta"]["stop"]
    dec = H.make_decoder(_small_dims(golden), golden["dec"], stop_threshold=m["threshold"])
    dec.chunk_steps = 3  # (rounded up to 4) forces several ttsdec_decode calls with state carried in the workspace
    with torch.no_grad():
        torch.manual_seed(m["seed"])
        y, s, w = dec(golden["memory"].cuda(), None, None, 0)
    assert y.shape[1] == m["T"]
    H.assert_close(y.cpu(), c["stop/y"], RTOL, ATOL, "y")
    H.assert_close(w.cpu(), c["stop/w"], RTOL, ATOL, "w")


def test_golden_postnet_unit(golden, H):
    c, d = golden["cases"], golden["meta"]["small_dims"]
    pn = H.make_postnet(d["d_mel"], d["postnet_hidden"], d["postnet_layers"], golden["post"])
    with torch.no_grad():
        out = pn(c["unit/post_y"].cuda()).cpu()
    H.assert_close(out, c["unit/post_out"], RTOL, ATOL, "postnet")


def test_golden_cell_step(golden, H):
    """Taco2ProdDecoderCell.forward (ttsdec_cell_step) against the oracle's single step."""
    dims = _small_dims(golden)
    dec = H.make_decoder(dims, golden["dec"])
    cell = dec.decoder_cell
    cell.dropout_source = "reference_rng"
    mem = golden["memory"]
    B, L, _ = mem.shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, 1, dims.d_mel, generator=g)
    w = torch.rand(B, L, generator=g)
    w = w / w.sum(1, keepdim=True)
    ctx = torch.randn(B, dims.d_ctx, generator=g) * 0.3
    ha, ca = torch.randn(B, dims.h_att, generator=g) * 0.3, torch.randn(B, dims.h_att, generator=g) * 0.3
    hd, cd = torch.randn(B, dims.h_dec, generator=g) * 0.3, torch.randn(B, dims.h_dec, generator=g) * 0.3
    torch.manual_seed(5)
    m0, m1 = O.draw_prenet_masks(B, dims.d_pre, dims.d_pre)
    ox, (ow, octx, (oha, oca), (ohd, ocd)) = O.decoder_cell_step(
        x[:, -1], (w, ctx, (ha, ca), (hd, cd)), mem, golden["dec"], dims, torch.stack([m0, m1])
    )
    dev = "cuda:0"
    with torch.no_grad():
        torch.manual_seed(5)
        xd, cx, (w2, ctx2, ((ha2, ca2), (hd2, cd2))) = cell(
            x.to(dev), (w.to(dev), ctx.to(dev), ((ha.to(dev), ca.to(dev)), (hd.to(dev), cd.to(dev)))), mem.to(dev), None
        )
    H.assert_close(xd.cpu(), ox, RTOL, ATOL, "x_dec")
    H.assert_close(w2.cpu(), ow, RTOL, ATOL, "w")
    H.assert_close(ctx2.cpu(), octx, RTOL, ATOL, "ctx")
    H.assert_close(ha2.cpu(), oha, RTOL, ATOL, "h_att")
    H.assert_close(cd2.cpu(), ocd, RTOL, ATOL, "c_dec")


# --------------------------------------------------------------------------
# oracle on the same seeded inputs, LJSpeech dims (BASELINE.json configs[1] and [2])
# --------------------------------------------------------------------------
@pytest.mark.parametrize("B,L,T,lengths", [(64, 120, 24, None), (256, 120, 6, None), (150, 33, 6, None), (5, 37, 10, [37, 30, 12, 37, 1]), (1, 9, 12, None)])
def test_ljspeech_dims_vs_oracle(H, B, L, T, lengths):
    dims = O.DecoderDims()
    wts = O.random_decoder_weights(dims, seed=42, nonzero_init_state=True)
    mem = O.synthetic_memory(B, L, dims.d_ctx, lengths=lengths, seed=1234)
    masks = O.synthetic_masks(T, B, dims.d_pre, seed=123)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T - 1, masks=masks)
    dec = H.make_decoder(dims, wts)
    y, s, w, fired = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    assert not fired and y.shape == oy.shape
    H.assert_close(y, oy, RTOL, ATOL, "y")
    H.assert_close(s, os_, RTOL, ATOL, "s")
    H.assert_close(w, ow, RTOL, ATOL, "w")
    assert torch.equal(w.argmax(-1), ow.argmax(-1)), "attention argmax must be bit-exact"


@pytest.mark.parametrize("B,L,T", [(256, 120, 40), (64, 120, 60), (128, 57, 20), (150, 33, 16), (3, 9, 12)])
def test_split_f16_precision_mode_vs_oracle(H, B, L, T):
    """TTSDEC_PREC_SPLIT_F16 (hi/lo fp16 planes, 3 products on the f16 MFMA, fp32 accumulate)
    must meet the same bar as the exact path: 1e-4 relative on mel / stop / weights, argmax exact."""
    dims = O.DecoderDims()
    wts = O.random_decoder_weights(dims, seed=42, nonzero_init_state=True)
    mem = O.synthetic_memory(B, L, dims.d_ctx, seed=1234)
    masks = O.synthetic_masks(T, B, dims.d_pre, seed=123)
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T - 1, masks=masks)
    dec = H.make_decoder(dims, wts)
    dec.precision = "split_f16"
    y, s, w, fired = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    assert dec.engine(torch.device("cuda:0")).precision() == "split_f16"
    assert not fired and y.shape == oy.shape
    H.assert_close(y, oy, RTOL, ATOL, "y")
    H.assert_close(s, os_, RTOL, ATOL, "s")
    H.assert_close(w, ow, RTOL, ATOL, "w")
    assert torch.equal(w.argmax(-1), ow.argmax(-1)), "attention argmax must be bit-exact"
    # report how close the two arithmetic modes are to each other and to the oracle
    dec.precision = "f32"
    y32, _, _, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    print(f"B={B} T={T}: max|y_split - oracle| = {float((y - oy).abs().max()):.3e}, "
          f"max|y_f32 - oracle| = {float((y32 - oy).abs().max()):.3e}, max|y| = {float(oy.abs().max()):.3f}")


def test_split_f16_handles_tiny_and_large_values(H):
    """Operand magnitudes across the fp16 range edges: subnormal-range activations (hi plane
    forced to zero) and weights scaled up/down."""
    dims = O.DecoderDims(d_mel=16, d_pre=32, d_ctx=64, h_att=64, h_dec=96)
    wts = O.random_decoder_weights(dims, seed=3)
    wts["decoder_cell.attention_rnn.weight_hh"] = wts["decoder_cell.attention_rnn.weight_hh"] * 1e-4
    wts["decoder_cell.decoder_rnn.weight_ih"][:, :7] *= 30.0
    B, L, T = 5, 13, 10
    mem = O.synthetic_memory(B, L, dims.d_ctx, lengths=[13, 2, 13, 7, 1]) * 1e-3
    masks = O.synthetic_masks(T, B, dims.d_pre)
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T - 1, masks=masks)
    dec = H.make_decoder(dims, wts)
    dec.precision = "split_f16"
    y, s, w, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    H.assert_close(y, oy, RTOL, ATOL, "y")
    H.assert_close(w, ow, RTOL, ATOL, "w")


@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_two_frames_per_step_r2(H, prec):
    """decoder.r = 2 (config-rdh/sandra use r=2): fc_mel emits two frames, only the last is fed
    back (decoder.py:48), fc_stop emits two logits; teacher frame index is t*r-1 (decoder.py:41-42,66)."""
    dims = O.DecoderDims(d_mel=24, r=2, d_pre=32, d_ctx=64, h_att=64, h_dec=96)
    wts = O.random_decoder_weights(dims, seed=8, nonzero_init_state=True)
    B, L, T = 6, 14, 9
    mem = O.synthetic_memory(B, L, dims.d_ctx, lengths=[14, 14, 3, 9, 14, 1])
    masks = O.synthetic_masks(T, B, dims.d_pre, seed=5)
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T - 1, masks=masks)
    assert oy.shape == (B, T * 2, 24) and os_.shape == (B, T * 2, 1)
    dec = H.make_decoder(dims, wts)
    dec.precision = prec
    y, s, w, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    H.assert_close(y, oy, RTOL, ATOL, "y")
    H.assert_close(s, os_, RTOL, ATOL, "s")
    H.assert_close(w, ow, RTOL, ATOL, "w")
    # teacher forcing, r = 2, 17 teacher frames -> 8 steps (the odd frame is dropped, decoder.py:41)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 17, 24, generator=g) * 0.5
    flags = [True, False, True, True, False, True, True]
    oy2, os2, ow2 = O.decode(wts, dims, mem, masks=masks, x=x, teacher_flags=flags)
    assert oy2.shape[1] == 16
    y2, s2, w2, _ = H.run_decoder_with_masks(dec, mem, masks, x=x, flags=flags)
    H.assert_close(y2, oy2, RTOL, ATOL, "y teacher r=2")
    H.assert_close(w2, ow2, RTOL, ATOL, "w teacher r=2")


@pytest.mark.parametrize("B,L,T", [(2, 1, 5), (3, 2, 6), (4, 301, 8), (2, 520, 4), (33, 77, 5)])  # (L = 520: more rows per wave than lanes)
def test_memory_length_edges(H, B, L, T):
    """L = 1 (the only column is the absorbing one, e = 1e4), L = 2, and L longer than the
    LJSpeech case; B = 33 leaves a ragged last row tile."""
    dims = O.DecoderDims()
    wts = O.random_decoder_weights(dims, seed=11)
    mem = O.synthetic_memory(B, L, dims.d_ctx, seed=2)
    masks = O.synthetic_masks(T, B, dims.d_pre, seed=9)
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T - 1, masks=masks)
    for prec in ("f32", "split_f16"):
        dec = H.make_decoder(dims, wts)
        dec.precision = prec
        y, s, w, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
        H.assert_close(y, oy, RTOL, ATOL, f"y {prec}")
        H.assert_close(w, ow, RTOL, ATOL, f"w {prec}")
        assert torch.equal(w.argmax(-1), ow.argmax(-1))
    if L == 1:
        assert float((ow - 1.0).abs().max()) == 0.0  # all mass stays on the single, absorbing column


def test_teacher_forcing_through_graph_replay(H):
    """>= 15 steps go through the captured hipGraph: teacher frames, per-step force flags and the
    masks are all read through the device control block at replay time."""
    dims = O.DecoderDims(d_mel=16, d_pre=32, d_ctx=64, h_att=64, h_dec=96)
    wts = O.random_decoder_weights(dims, seed=12, nonzero_init_state=True)
    B, L, Tx = 5, 21, 47
    mem = O.synthetic_memory(B, L, dims.d_ctx, lengths=[21, 21, 8, 15, 2])
    masks = O.synthetic_masks(Tx, B, dims.d_pre, seed=2)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Tx, 16, generator=g) * 0.5
    flags = (torch.rand(Tx - 1, generator=g) > 0.3).tolist()
    oy, os_, ow = O.decode(wts, dims, mem, masks=masks, x=x, teacher_flags=flags)
    for prec in ("f32", "split_f16"):
        dec = H.make_decoder(dims, wts)
        dec.precision = prec
        y, s, w, _ = H.run_decoder_with_masks(dec, mem, masks, x=x, flags=flags)
        H.assert_close(y, oy, RTOL, ATOL, f"y {prec}")
        H.assert_close(w, ow, RTOL, ATOL, f"w {prec}")


def test_back_to_back_calls_with_changing_shapes_and_modes(H):
    """One Decoder, many calls: batch / memory length / precision / weights change between calls, so
    the cached graph, workspace and packed blob must each be refreshed exactly when needed."""
    dims = O.DecoderDims()
    wts = O.random_decoder_weights(dims, seed=21)
    dec = H.make_decoder(dims, wts)
    T = 20
    ref = {}
    for (B, L) in [(64, 40), (7, 40), (64, 33), (64, 40)]:
        mem = O.synthetic_memory(B, L, dims.d_ctx, seed=B + L)
        masks = O.synthetic_masks(T, B, dims.d_pre, seed=3)
        if (B, L) not in ref:
            ref[(B, L)] = O.decode(wts, dims, mem, max_steps=T - 1, masks=masks)
        for prec in ("split_f16", "f32", "split_f16"):
            dec.precision = prec
            y, s, w, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
            H.assert_close(y, ref[(B, L)][0], RTOL, ATOL, f"y B={B} L={L} {prec}")
            H.assert_close(w, ref[(B, L)][2], RTOL, ATOL, f"w B={B} L={L} {prec}")
    # change the weights in place: the next call must repack (fingerprint = data_ptr + version)
    with torch.no_grad():
        dec.fc_mel.bias.add_(0.25)
    wts2 = dict(wts)
    wts2["fc_mel.bias"] = wts["fc_mel.bias"] + 0.25
    mem = O.synthetic_memory(7, 40, dims.d_ctx, seed=47)
    masks = O.synthetic_masks(T, 7, dims.d_pre, seed=3)
    oy, _, _ = O.decode(wts2, dims, mem, max_steps=T - 1, masks=masks)
    y, _, _, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    H.assert_close(y, oy, RTOL, ATOL, "y after in-place weight update")


def test_large_single_gpu_batch_runs_and_is_finite(H):
    """B = 2048 on one GPU (BASELINE.json configs[3] unsharded): many row tiles per kernel."""
    dims = O.DecoderDims()
    wts = O.random_decoder_weights(dims, seed=5)
    B, L, T = 2048, 120, 32
    mem = O.synthetic_memory(B, L, dims.d_ctx, seed=1).cuda()
    dec = H.make_decoder(dims, wts)
    dec.dropout_source, dec.dropout_seed, dec.precision = "philox", 9, "split_f16"
    with torch.no_grad():
        y, s, w = dec(mem, None, None, T - 1)
    assert y.shape == (B, T, 80) and torch.isfinite(y).all()
    assert float((w.sum(-1) - 1).abs().max()) < 1e-4
    # rows are independent: the first 512 utterances decoded alone give the same bits (same launch sequence - above 384
    # utterances the attention query is a launch of its own - hence the same summation order)
    with torch.no_grad():
        y2, _, w2 = dec(mem[:512].contiguous(), None, None, T - 1)
    assert torch.equal(y[:512], y2) and torch.equal(w[:512], w2)
    # ... and under the other launch sequence (192 utterances: the query a job of the attention role) the same values within the parity bar
    with torch.no_grad():
        y3, _, w3 = dec(mem[:192].contiguous(), None, None, T - 1)
    H.assert_close(y3.cpu(), y[:192].cpu(), RTOL, ATOL, "y, 192 of 2048")
    H.assert_close(w3.cpu(), w[:192].cpu(), RTOL, ATOL, "w, 192 of 2048")


def test_c_abi_rejects_bad_arguments(H):
    """Error behaviour at the boundary: codes, never crashes."""
    from torch_tts_amd import _lib

    dims = O.DecoderDims(d_mel=8, d_pre=16, d_ctx=32, h_att=32, h_dec=32)
    dec = H.make_decoder(dims, O.random_decoder_weights(dims, seed=1))
    dev = torch.device("cuda:0")
    eng = dec.engine(dev)
    mem = torch.zeros(2, 5, 32, device=dev)
    y = torch.empty(2, 4, 8, device=dev)
    s = torch.empty(2, 4, device=dev)
    w = torch.empty(2, 4, 5, device=dev)
    t_out = torch.zeros(2, dtype=torch.int32, device=dev)
    kw = dict(stop_threshold=-2.0, check_stop=True, seed=0, teacher=None, teacher_flags=None, y=y, s=s, w=w, t_out=t_out)
    with pytest.raises(_lib.TtsdecError) as ei:  # masks mode without masks
        eng.decode(mem, t_begin=0, n_steps=4, dropout_mode=_lib.DROPOUT_MASKS, masks=None, **kw)
    assert ei.value.code == _lib.ERR_INVALID_ARG
    with pytest.raises(_lib.TtsdecError) as ei:  # odd t_begin
        eng.decode(mem, t_begin=3, n_steps=1, dropout_mode=_lib.DROPOUT_OFF, masks=None, **kw)
    assert ei.value.code == _lib.ERR_INVALID_ARG
    with pytest.raises(_lib.TtsdecError) as ei:  # more steps than the output buffers hold
        eng.decode(mem, t_begin=0, n_steps=5, dropout_mode=_lib.DROPOUT_OFF, masks=None, **kw)
    assert ei.value.code == _lib.ERR_INVALID_ARG
    lib = _lib.load()
    ws = eng.workspace(2, 5)
    rc = lib.ttsdec_decode(eng._h, mem.data_ptr(), 2, 5, 0, 4, 4, -2.0, 1, 0, None, 0, None, 0, None, y.data_ptr(), s.data_ptr(),
                           w.data_ptr(), t_out.data_ptr(), ws.data_ptr(), 128, None)  # workspace too small
    assert rc == _lib.ERR_WORKSPACE
    eng.decode(mem, t_begin=0, n_steps=4, dropout_mode=_lib.DROPOUT_OFF, masks=None, **kw)  # and a good call still works
    torch.cuda.synchronize()
    assert t_out.tolist() == [4, 0]


# --------------------------------------------------------------------------
# Taco2DecoderCell (config-rdh / sandra / template), golden vectors from the reference
# --------------------------------------------------------------------------
def _t2dims(g):
    d = g["meta"]["dims"]
    return O.DecoderDims(d_mel=d["d_mel"], r=d["r"], d_pre=d["d_pre"], d_ctx=d["d_ctx"], h_att=d["h_att"], h_dec=d["h_dec"])


@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_taco2_cell_golden(golden_taco2, H, prec):
    g, c = golden_taco2, golden_taco2["cases"]
    dec = H.make_taco2_decoder(_t2dims(g), g["dec"])
    dec.precision = prec
    y, s, w, fired = H.run_decoder_with_masks(dec, g["memory"], H.flat_masks(c["infer/m0"], c["infer/m1"]), max_steps=8)
    assert not fired and y.shape == c["infer/y"].shape
    H.assert_close(y, c["infer/y"], RTOL, ATOL, "y")
    H.assert_close(s, c["infer/s"], RTOL, ATOL, "s")
    H.assert_close(w, c["infer/w"], RTOL, ATOL, "w")
    assert torch.equal(w.argmax(-1), c["infer/w"].argmax(-1))
    y2, s2, w2, _ = H.run_decoder_with_masks(dec, g["memory"], H.flat_masks(c["teacher/m0"], c["teacher/m1"]), x=c["teacher/x"])
    H.assert_close(y2, c["teacher/y"], RTOL, ATOL, "y teacher")
    H.assert_close(w2, c["teacher/w"], RTOL, ATOL, "w teacher")


def test_taco2_cell_module_forward_reference_rng_and_cell_step(golden_taco2, H):
    """Decoder.forward around Taco2DecoderCell under the reference's seed, and the bare cell step."""
    g, c = golden_taco2, golden_taco2["cases"]
    dims = _t2dims(g)
    dec = H.make_taco2_decoder(dims, g["dec"])
    mem = g["memory"].cuda()
    with torch.no_grad():
        torch.manual_seed(g["meta"]["seeds"]["infer"])
        y, s, w = dec(mem, None, None, 8)
    H.assert_close(y.cpu(), c["infer/y"], RTOL, ATOL, "y")
    H.assert_close(w.cpu(), c["infer/w"], RTOL, ATOL, "w")
    # one bare cell step vs the oracle
    cell = dec.decoder_cell
    B, L, _ = g["memory"].shape
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(B, 2, dims.d_mel, generator=gen)
    w0 = torch.rand(B, L, generator=gen)
    w0 = w0 / w0.sum(1, keepdim=True)
    hc = [(torch.randn(B, H_, generator=gen) * 0.3, torch.randn(B, H_, generator=gen) * 0.3) for H_ in (dims.h_att, dims.h_dec)]
    torch.manual_seed(9)
    m0 = torch.empty(B, 128).bernoulli_(0.5).to(torch.uint8)
    m1 = torch.empty(B, dims.d_pre).bernoulli_(0.5).to(torch.uint8)
    ox, octx, (ow, ohc) = O.taco2_cell_step(x[:, -1], (w0, hc), g["memory"], g["dec"], dims, [m0, m1])
    with torch.no_grad():
        torch.manual_seed(9)
        xd, ctx, (w1, hc1) = cell(x.cuda(), (w0.cuda(), [(a.cuda(), b.cuda()) for a, b in hc]), mem, None)
    H.assert_close(xd.cpu(), ox, RTOL, ATOL, "x_dec")
    H.assert_close(ctx.cpu(), octx, RTOL, ATOL, "ctx")
    H.assert_close(w1.cpu(), ow, RTOL, ATOL, "w")
    H.assert_close(hc1[1][0].cpu(), ohc[1][0], RTOL, ATOL, "h1")


def test_taco2_cell_rdh_dims_vs_oracle(H):
    """config-rdh.yaml dims (r=1, dim_rnn [1024, 1024], dim_pre 256, encoder 512) at B=64."""
    dims = O.DecoderDims(d_mel=80, r=1, d_pre=256, d_ctx=512, h_att=1024, h_dec=1024)
    wts = O.random_taco2_weights(dims, seed=2)
    B, L, T = 64, 90, 12
    mem = O.synthetic_memory(B, L, dims.d_ctx, seed=8)
    g = torch.Generator().manual_seed(1)
    m0 = (torch.rand(T, B, 128, generator=g) >= 0.5).to(torch.uint8)
    m1 = (torch.rand(T, B, 256, generator=g) >= 0.5).to(torch.uint8)

    class M:
        def __getitem__(self, t):
            return [m0[t], m1[t]]
    oy, os_, ow = O.taco2_decode(wts, dims, mem, max_steps=T - 1, masks=M())
    for prec in ("f32", "split_f16"):
        dec = H.make_taco2_decoder(dims, wts)
        dec.precision = prec
        y, s, w, _ = H.run_decoder_with_masks(dec, mem, H.flat_masks(m0, m1), max_steps=T - 1)
        H.assert_close(y, oy, RTOL, ATOL, f"y {prec}")
        H.assert_close(s, os_, RTOL, ATOL, f"s {prec}")
        H.assert_close(w, ow, RTOL, ATOL, f"w {prec}")
        assert torch.equal(w.argmax(-1), ow.argmax(-1))


def test_melpostnet2_golden_and_template_dims(golden_taco2, H):
    """MelPostnet2 (Conv1dFix residual blocks): the reference's own vectors at reduced dims, then
    config_template dims (num_mels 80, hidden 256, 3 layers) against the oracle in all three modes."""
    g, c, d = golden_taco2, golden_taco2["cases"], golden_taco2["meta"]["dims"]
    pn = H.make_postnet2(d["d_mel"], d["postnet_hidden"], d["postnet_layers"], g["post"])
    with torch.no_grad():
        H.assert_close(pn(c["infer/y"].cuda()).cpu(), c["infer/y_post"], RTOL, ATOL, "y_post")
        H.assert_close(pn(c["unit/post_y"].cuda()).cpu(), c["unit/post_out"], RTOL, ATOL, "unit")
    pw = O.random_postnet2_weights(80, 256, 3, seed=4)
    gen = torch.Generator().manual_seed(6)
    y = torch.randn(4, 97, 80, generator=gen)
    ref = O.mel_postnet2(y, pw, 3)
    pn = H.make_postnet2(80, 256, 3, pw)
    for mode, rt, at in (("f32", RTOL, ATOL), ("split_f16", RTOL, ATOL), ("bf16", 5e-2, 5e-2)):
        pn.precision = mode
        with torch.no_grad():
            out = pn(y.cuda()).cpu()
        print(f"MelPostnet2 {mode}: max abs err {float((out - ref).abs().max()):.3e} (|ref| max {float(ref.abs().max()):.2f})")
        H.assert_close(out, ref, rt, at, f"postnet2 {mode}")


def test_postnet_ljspeech_dims_vs_oracle(H):
    pw = O.random_postnet_weights(80, 512, 3, seed=9)
    g = torch.Generator().manual_seed(2)
    y = torch.randn(3, 57, 80, generator=g)
    ref = O.mel_postnet(y, pw, 3)
    pn = H.make_postnet(80, 512, 3, pw)
    with torch.no_grad():
        out = pn(y.cuda()).cpu()
    H.assert_close(out, ref, RTOL, ATOL, "postnet")


@pytest.mark.parametrize("B,T", [(5, 131), (7, 301)])  # 655 rows: 64x64 tiles; 2107 rows: the 128x128 tiles, ragged last tile
@pytest.mark.parametrize("mode,rtol,atol", [("split_f16", RTOL, ATOL), ("bf16", 3e-2, 3e-2)])
def test_postnet_low_precision_modes_vs_oracle(H, mode, rtol, atol, B, T):
    """Postnet on 16-bit MFMA operands.  split_f16 must meet the exact path's bar; bf16
    (BASELINE.json configs[2]: 'Postnet conv1d on MFMA bf16, mel tolerance vs CPU reported')
    carries 8 significand bits per operand - its measured error is printed and bounded."""
    pw = O.random_postnet_weights(80, 512, 3, seed=9)
    g = torch.Generator().manual_seed(2)
    y = torch.randn(B, T, 80, generator=g)
    ref = O.mel_postnet(y, pw, 3)
    pn = H.make_postnet(80, 512, 3, pw)
    pn.precision = mode
    with torch.no_grad():
        out = pn(y.cuda()).cpu()
    err = (out - ref).abs()
    print(f"postnet {mode}: max abs err {float(err.max()):.3e}, rel-to-max {float(err.max() / ref.abs().max()):.3e}, "
          f"mean abs err {float(err.mean()):.3e}")
    H.assert_close(out, ref, rtol, atol, f"postnet {mode}")


@pytest.mark.parametrize("B,T", [(5, 131), (3, 600), (1, 257)])  # utterance edges inside 256-row tiles; 600: the bench's length
def test_postnet_bf16_vs_its_own_rounding_emulated_on_the_cpu(H, B, T):
    """The bf16 Postnet against a CPU emulation of ITS arithmetic (operands rounded to bf16 between the layers, fp32 accumulation,
    fp32 BN + isru): what is left is summation order and bf16 roundings that flip - several times tighter than the
    bound against the exact oracle, so a tap that leaks across an utterance edge or a tile edge of the 256 x 256 conv kernel
    (csrc/conv256.hip: the hidden -> hidden layers at these dims) cannot hide in bf16's own error."""
    pw = O.random_postnet_weights(80, 512, 3, seed=9)
    g = torch.Generator().manual_seed(2)
    y = torch.randn(B, T, 80, generator=g)

    def rb(x):
        return x.to(torch.bfloat16).to(torch.float32)

    xc = rb(y).transpose(1, 2)
    for i in range(3):
        xc = torch.nn.functional.conv1d(xc, rb(pw[f"conv.{i}.0.weight"]), None, padding=2)
        inv = 1.0 / torch.sqrt(pw[f"conv.{i}.1.running_var"] + 1e-5)
        alpha = pw[f"conv.{i}.1.weight"] * inv
        beta = pw[f"conv.{i}.1.bias"] - pw[f"conv.{i}.1.running_mean"] * alpha
        xc = rb(O.isru(xc * alpha[None, :, None] + beta[None, :, None]))
    ref = y + torch.nn.functional.linear(xc.transpose(1, 2), rb(pw["fc_out.weight"]))
    pn = H.make_postnet(80, 512, 3, pw)
    pn.precision = "bf16"
    os.environ["TTSDEC_CONV256_FORCE"] = "1"  # (the library takes that kernel only where its tiles fill the chip: csrc/conv256.hip)
    try:
        with torch.no_grad():
            out = pn(y.cuda()).cpu()
    finally:
        del os.environ["TTSDEC_CONV256_FORCE"]
    err = float((out - ref).abs().max())
    print(f"postnet bf16 vs bf16 emulation, B = {B}, T = {T}: max abs err {err:.3e}")
    assert err < 8e-3, err  # (the shared tile and the 256-wide kernel both measure 2e-3 ... 5e-3 here; a leaked tap is >= 5e-2)


@pytest.mark.parametrize("B,T", [(5, 131), (3, 600), (1, 257)])
def test_postnet_fp32_on_the_256_wide_conv_kernel_vs_oracle(H, B, T):
    """The exact-fp32 form of csrc/conv256.hip (the hidden -> hidden layers of the headline's Postnet) against the oracle at the
    fp32 bar, forced at shapes the oracle finishes in seconds: utterance edges inside 256-row tiles, a ragged last tile."""
    pw = O.random_postnet_weights(80, 512, 3, seed=9)
    g = torch.Generator().manual_seed(2)
    y = torch.randn(B, T, 80, generator=g)
    ref = O.mel_postnet(y, pw, 3)
    pn = H.make_postnet(80, 512, 3, pw)
    os.environ["TTSDEC_CONV256_FORCE"] = "1"
    try:
        with torch.no_grad():
            out = pn(y.cuda()).cpu()
    finally:
        del os.environ["TTSDEC_CONV256_FORCE"]
    H.assert_close(out, ref, RTOL, ATOL, "postnet fp32 (conv256)")


def test_philox_mode_matches_oracle_masks(H):
    """On-device dropout: the same Philox function restated in the oracle gives the masks;
    the decode must match the oracle run with those masks injected."""
    dims = O.DecoderDims(d_mel=20, d_pre=36, d_ctx=40, h_att=72, h_dec=88)
    wts = O.random_decoder_weights(dims, seed=4)
    B, L, T, seed = 7, 11, 9, 0x1234567890ABCDEF
    mem = O.synthetic_memory(B, L, dims.d_ctx, lengths=[11, 11, 4, 9, 11, 2, 7])
    masks = torch.stack([O.philox_keep_masks(seed, t, B, dims.d_pre) for t in range(T)])
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T - 1, masks=masks)
    dec = H.make_decoder(dims, wts)
    dec.dropout_source, dec.dropout_seed = "philox", seed
    with torch.no_grad():
        y, s, w = dec(mem.cuda(), None, None, T - 1)
    H.assert_close(y.cpu(), oy, RTOL, ATOL, "y")
    H.assert_close(w.cpu(), ow, RTOL, ATOL, "w")


# --------------------------------------------------------------------------
# parity at the headline size and length (BASELINE.json configs[1] and [2]): B = 64 / 256, L = 120,
# T = 600 frames, on bench.py's own model and inputs, both arithmetic modes, injected masks and the
# on-device Philox masks.  SURVEY 7 H7: argmax equality over >= 600 steps x 256 utterances.
# --------------------------------------------------------------------------
_HEADLINE = {}


def _headline_case(B, mask_kind):
    """bench.py's workload (LJSpeech dims, init seed 42, ids seed 1234 through the encoder), the oracle's
    outputs for all 600 frames (computed once per case) and the masks that produced them."""
    key = (B, mask_kind)
    if key in _HEADLINE:
        return _HEADLINE[key]
    import bench
    import torch_tts_amd as T

    L, NF = 120, 600
    if "model" not in _HEADLINE:
        torch.manual_seed(42)
        _HEADLINE["model"] = T.build_tacotron(bench.LJSPEECH).eval().cuda()
    model = _HEADLINE["model"]
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(1, 40, (B, L), generator=g).cuda()
    lens = torch.full((B,), L, dtype=torch.long).cuda()
    with torch.no_grad():
        mem = torch.cat([model.encoder(ids[i : i + 64], lens[i : i + 64]) for i in range(0, B, 64)]).contiguous()
    if mask_kind == "injected":
        masks = O.synthetic_masks(NF, B, 256, seed=123)
    else:
        masks = torch.stack([O.philox_keep_masks(123, t, B, 256) for t in range(NF)])
    sd = {k: v.detach().cpu() for k, v in model.decoder.state_dict().items()}
    pw = {k: v.detach().cpu() for k, v in model.postnet.state_dict().items()}
    with torch.no_grad():
        oy, os_, ow = O.decode(sd, O.DecoderDims(), mem.cpu(), max_steps=NF - 1, masks=masks)
        opost = O.mel_postnet(oy, pw, 3)
    _HEADLINE[key] = (model, mem, masks, oy, os_, ow, opost)
    return _HEADLINE[key]


def _drift_report(name, a, b, block=100):
    """max relative error (bar: 1e-4 with a 1e-5 absolute floor) per block of steps, so drift is visible"""
    e = (a - b).abs() / (ATOL / RTOL + b.abs())
    per = [float(e[:, t : t + block].max()) for t in range(0, a.shape[1], block)]
    print(f"  {name}: max rel err per {block}-step block: " + " ".join(f"{v:.2e}" for v in per))
    return max(per)


@pytest.mark.parametrize("B", [256, 64])
@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_headline_600_frames_vs_oracle(H, B, prec):
    model, mem, masks, oy, os_, ow, opost = _headline_case(B, "injected")
    dec = model.decoder
    dec.precision = prec
    y, s, w, fired = H.run_decoder_with_masks(dec, mem, masks, max_steps=599)
    assert dec.engine(torch.device("cuda:0")).precision() == prec
    assert not fired and y.shape == oy.shape == (B, 600, 80)
    print(f"B={B} {prec}:")
    worst = max(_drift_report("y", y, oy), _drift_report("s", s, os_), _drift_report("w", w, ow))
    model.postnet.precision = prec
    with torch.no_grad():
        yp = model.postnet(y.cuda()).cpu()
    worst = max(worst, _drift_report("y_post", yp, opost))
    mism = int((w.argmax(-1) != ow.argmax(-1)).sum())
    print(f"  argmax mismatches: {mism} of {B * 600}")
    assert mism == 0, "attention argmax must be bit-exact on all 600 steps of every utterance"
    H.assert_close(y, oy, RTOL, ATOL, "y")
    H.assert_close(s, os_, RTOL, ATOL, "s")
    H.assert_close(w, ow, RTOL, ATOL, "w")
    H.assert_close(yp, opost, RTOL, ATOL, "y_post")
    assert worst <= 1.0


def test_headline_600_frames_philox_vs_oracle(H):
    """The configuration bench.py times by default: B = 256, split-fp16, on-device Philox dropout (seed 123)."""
    model, mem, masks, oy, os_, ow, opost = _headline_case(256, "philox")
    dec = model.decoder
    dec.precision, dec.dropout_source, dec.dropout_seed = "split_f16", "philox", 123
    with torch.no_grad():
        y, s, w = dec(mem, None, None, 599)
        model.postnet.precision = "split_f16"
        yp = model.postnet(y).cpu()
    y, s, w = y.cpu(), s.cpu(), w.cpu()
    print("B=256 split_f16 philox:")
    for n, a, b in (("y", y, oy), ("s", s, os_), ("w", w, ow), ("y_post", yp, opost)):
        _drift_report(n, a, b)
        H.assert_close(a, b, RTOL, ATOL, n)
    assert torch.equal(w.argmax(-1), ow.argmax(-1))


# --------------------------------------------------------------------------
# size-independent properties at BASELINE.json's full size (B=256, L=120, T=600)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_full_size_properties(H, prec):
    dims = O.DecoderDims()
    wts = O.random_decoder_weights(dims, seed=42)
    B, L, T = 256, 120, 600
    mem = O.synthetic_memory(B, L, dims.d_ctx, seed=1234).cuda()
    dec = H.make_decoder(dims, wts)
    dec.precision = prec
    dec.dropout_source, dec.dropout_seed = "philox", 123
    with torch.no_grad():
        y, s, w = dec(mem, None, None, T - 1)
        y2, s2, w2 = dec(mem, None, None, T - 1)
    assert y.shape == (B, T, 80) and s.shape == (B, T, 1) and w.shape == (B, T, L)
    assert torch.isfinite(y).all() and torch.isfinite(s).all()
    # run-to-run determinism, bit for bit
    assert torch.equal(y, y2) and torch.equal(w, w2) and torch.equal(s, s2)
    # attention.py:117-123: rows sum to 1; mass only stays or moves one slot forward
    assert float((w.sum(-1) - 1).abs().max()) < 1e-4
    pos = (w * torch.arange(L, device=w.device, dtype=w.dtype)).sum(-1)
    assert bool((pos[:, 1:] >= pos[:, :-1] - 1e-4).all()), "expected attention position must be non-decreasing"
    for k in range(4):
        assert float(w[:, k, k + 2 :].abs().max()) == 0.0, "support after k steps lies within [0, k+1]"


def test_shard_equivalence_bitwise(H):
    """Utterances never interact inside the step (decoder_cell.py:180-195 is row-wise), so
    decoding a batch in two shards must equal decoding it whole, bit for bit.
    Whole batch and shards must lie in one regime of the launch schedule (the same GEMM tiling and K order; the table in
    DESIGN.md section 6): split-fp16 65 .. 384 utterances (two launches per step: the query a job of the attention role) or more
    than 384; exact fp32 more than 128."""
    dims = O.DecoderDims()
    wts = O.random_decoder_weights(dims, seed=1)
    for prec, B in (("split_f16", 256), ("split_f16", 1024), ("f32", 768)):
        L, T = 64, 12
        mem = O.synthetic_memory(B, L, dims.d_ctx, seed=5).cuda()
        masks = O.synthetic_masks(T, B, dims.d_pre, seed=6)
        dec = H.make_decoder(dims, wts)
        dec.precision = prec
        h = B // 2
        y, s, w, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
        ya, sa, wa, _ = H.run_decoder_with_masks(dec, mem[:h], masks[:, :, :h].contiguous(), max_steps=T - 1)
        yb, sb, wb, _ = H.run_decoder_with_masks(dec, mem[h:], masks[:, :, h:].contiguous(), max_steps=T - 1)
        assert torch.equal(y, torch.cat([ya, yb])) and torch.equal(w, torch.cat([wa, wb])) and torch.equal(s, torch.cat([sa, sb])), (prec, B)


# --------------------------------------------------------------------------
# Encoder2 on the HIP path
# --------------------------------------------------------------------------
def test_encoder2_golden_and_ljspeech_dims(H):
    import json
    import os

    import numpy as np
    import torch_tts_amd as T

    gd = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gd, "encoder_small.npz"))
    meta = json.load(open(os.path.join(gd, "encoder_meta.json")))
    wts = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w/")}
    enc = T.Encoder2(meta["dims"]["alphabet"], dim_out=meta["dims"]["d_out"], dim_emb=meta["dims"]["d_emb"])
    missing, unexpected = enc.load_state_dict(wts, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    enc = enc.cuda().eval()
    assert enc.precision == "f32"  # (the reference's arithmetic is the default; split-fp16 is opt-in: ttsenc_set_precision)
    with torch.no_grad():
        for prec in ("f32", "split_f16"):
            enc.precision = prec
            for suf in ("", "2"):  # the reference's own vectors (ragged lengths; case 2: no utterance fills the padding)
                ids, lens, mem = (torch.from_numpy(g[n + suf]) for n in ("ids", "lengths", "memory"))
                out = enc(ids.cuda(), lens)
                assert out.shape == mem.shape
                H.assert_close(out.cpu(), mem, RTOL, ATOL, f"memory{suf} ({prec})")
                for b, n in enumerate(lens.tolist()):
                    if n < out.shape[1]:
                        assert float(out[b, n:].abs().max()) == 0.0, "padded rows must be exactly zero (rnn.py:126)"
        # LJSpeech dims against the oracle, and against the stock PyTorch-ROCm ops of the same module
        wl = O.random_encoder2_weights(40, 512, 512, seed=3)
        encl = T.Encoder2(40, dim_out=512, dim_emb=512)
        encl.load_state_dict(wl, strict=False)
        encl = encl.cuda().eval()
        gen = torch.Generator().manual_seed(7)
        B, L = 6, 47
        lens = torch.tensor([47, 33, 47, 1, 20, 46])
        ids = torch.randint(1, 40, (B, L), generator=gen)
        for b in range(B):
            ids[b, lens[b]:] = 0
        ref = O.encoder2(ids, lens, wl)
        for prec in ("split_f16", "f32"):
            encl.precision = prec
            out = encl(ids.cuda(), lens)
            H.assert_close(out.cpu(), ref, RTOL, ATOL, f"memory (LJSpeech dims, {prec})")
        encl.use_hip = False
        stock = encl(ids.cuda(), lens)
        H.assert_close(out.cpu(), stock.cpu(), RTOL, ATOL, "HIP vs stock ops")


def test_tacotron_forward_glue_sandra_style_config(H):
    """build_tacotron for a config-sandra / config_template style model: Taco2DecoderCell, r = 2,
    MelPostnet2 (postnet without type), encoder 256."""
    import torch_tts_amd as T

    cfg = {
        "text": {"alphabet": "abcdefghijklmnopqrstuvwxyz '.,?!"},
        "audio": {"num_mels": 80},
        "model": {
            "encoder": {"dim_emb": 64, "dim_out": 256},
            "decoder": {"type": "tacotron2", "r": 2, "dim_pre": 256, "dim_att": 256, "dim_rnn": [512, 512]},
            "postnet": {"dim_hidden": 256, "num_layers": 3},
        },
    }
    torch.manual_seed(0)
    model = T.build_tacotron(cfg).cuda().eval()
    assert isinstance(model.decoder.decoder_cell, T.Taco2DecoderCell) and isinstance(model.postnet, T.MelPostnet2)
    ids = torch.randint(1, 30, (3, 19)).cuda()
    lens = torch.tensor([19, 12, 19]).cuda()
    ids[1, 12:] = 0
    with torch.no_grad():
        y, y_post, s, out = model(ids, lens, max_steps=9)
    assert y.shape == (3, 20, 80) and y_post.shape == y.shape and s.shape == (3, 20, 1) and out["w"].shape == (3, 10, 19)
    assert torch.isfinite(y_post).all() and float((out["w"].sum(-1) - 1).abs().max()) < 1e-5


def test_tacotron_forward_glue(H):
    """Tacotron.forward / build_tacotron (tacotron.py:29-56,165-224) end to end on the GPU:
    shapes, dict keys, and y_post == postnet(y)."""
    import torch_tts_amd as T

    cfg = {
        "text": {"alphabet": "abcdefghijklmnopqrstuvwxyz '.,?!"},
        "audio": {"num_mels": 80},
        "model": {
            "encoder": {"dim_emb": 64, "dim_out": 512},
            "decoder": {"type": "tacotron2prod", "r": 1, "dim_pre": 256, "dim_att": 1024, "dim_rnn": [1024, 1024]},
            "postnet": {"type": "tacotron2", "dim_hidden": 512, "num_layers": 3},
        },
    }
    torch.manual_seed(0)
    model = T.build_tacotron(cfg).cuda().eval()
    ids = torch.randint(1, 30, (4, 21)).cuda()
    lens = torch.tensor([21, 17, 9, 21]).cuda()
    ids[1, 17:] = 0
    ids[2, 9:] = 0
    with torch.no_grad():
        y, y_post, s, out = model(ids, lens, max_steps=15)
        assert y.shape == (4, 16, 80) and y_post.shape == y.shape and s.shape == (4, 16, 1)
        assert out["w"].shape == (4, 16, 21) and float(out["kl_loss"]) == 0.0
        assert torch.equal(y_post, model.postnet(y))


# --------------------------------------------------------------------------
# the fused frame kernel (d_mel = 80, PreNet hidden 128 / 256) and the split-K projections:
# stop rule, teacher forcing, call boundaries
# --------------------------------------------------------------------------
def _stop_case(wts, dims, mem, masks, T, first=5):
    """(weights, threshold, k): the batch-global stop rule fires at step k, first <= k < T - 2.  The stop
    logits of a random model drift one way; if that is upwards the fc_stop sign is flipped."""
    for sign, first in ((1.0, first), (-1.0, first), (1.0, 2), (-1.0, 2)):
        w2 = {k: (v * sign if "fc_stop" in k else v) for k, v in wts.items()}
        _, s_all, _ = O.decode(w2, dims, mem, max_steps=T - 1, masks=masks, dropout="masks" if masks is not None else "off")
        m = s_all.reshape(s_all.shape[0], T, -1).amin(dim=(0, 2))  # per-step batch minimum
        run = torch.cummin(m, 0).values
        cand = [k for k in range(first, T - 2) if m[k] < run[k - 1]]
        if cand:
            k = cand[0]
            return w2, float((m[k] + run[k - 1]) / 2), k
    raise AssertionError("no mid-sequence record minimum of the stop logit; change the seed")


@pytest.mark.parametrize("prec", ["f32", "split_f16"])
@pytest.mark.parametrize("r,d_pre", [(1, 128), (2, 256)])
def test_frame_kernel_stop_teacher_and_chunks(H, prec, r, d_pre):
    dims = O.DecoderDims(d_mel=80, r=r, d_pre=d_pre, d_ctx=64, h_att=256, h_dec=192)
    wts = O.random_decoder_weights(dims, seed=21, nonzero_init_state=True)
    B, L, T = 37, 23, 44
    mem = O.synthetic_memory(B, L, dims.d_ctx, lengths=[23] * 30 + [1, 2, 5, 9, 14, 20, 23], seed=4)
    masks = O.synthetic_masks(T, B, dims.d_pre, seed=6)
    # (a) stop rule firing mid-graph: the frame is finished by the NEXT step's first kernel, the
    # batch-global rule must still be inclusive of the firing step and cut everything after it
    w2, thr, k = _stop_case(wts, dims, mem, masks, T)
    oy, os_, ow = O.decode(w2, dims, mem, max_steps=T - 1, masks=masks, stop_threshold=thr)
    assert ow.shape[1] == k + 1
    dec = H.make_decoder(dims, w2, stop_threshold=thr)
    dec.precision = prec
    y, s, w, fired = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    assert fired and w.shape[1] == k + 1
    H.assert_close(y, oy, RTOL, ATOL, "y stop")
    H.assert_close(s, os_, RTOL, ATOL, "s stop")
    H.assert_close(w, ow, RTOL, ATOL, "w stop")
    # (b) never firing: T = max_steps + 1, the last frame comes from the end-of-call launch
    dec2 = H.make_decoder(dims, wts)
    dec2.precision = prec
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T - 1, masks=masks)
    y, s, w, fired = H.run_decoder_with_masks(dec2, mem, masks, max_steps=T - 1)
    assert not fired and y.shape == oy.shape
    H.assert_close(y, oy, RTOL, ATOL, "y")
    H.assert_close(s, os_, RTOL, ATOL, "s")
    H.assert_close(w, ow, RTOL, ATOL, "w")
    H.assert_argmax(w, ow)
    # (c) teacher forcing with per-step flags through graph replay
    g = torch.Generator().manual_seed(9)
    Tx = 41 * r + (r - 1)
    x = torch.randn(B, Tx, 80, generator=g) * 0.5
    flags = (torch.rand(Tx // r - 1, generator=g) > 0.3).tolist()
    oy, os_, ow = O.decode(wts, dims, mem, masks=masks, x=x, teacher_flags=flags)
    y, s, w, _ = H.run_decoder_with_masks(dec2, mem, masks, x=x, flags=flags)
    H.assert_close(y, oy, RTOL, ATOL, "y teacher")
    H.assert_close(s, os_, RTOL, ATOL, "s teacher")
    H.assert_close(w, ow, RTOL, ATOL, "w teacher")
    # (d) unbounded decode in chunks: every call boundary hands the pending frame over
    w2, thr, k = _stop_case(wts, dims, mem, None, T)
    oy, os_, ow = O.decode(w2, dims, mem, max_steps=0, dropout="off", stop_threshold=thr)
    dec3 = H.make_decoder(dims, w2, stop_threshold=thr)
    dec3.precision = prec
    dec3.dropout_source = "off"
    dec3.chunk_steps = 4
    with torch.no_grad():
        y, s, w = dec3(mem.cuda(), None, None, 0)
    assert w.shape[1] == k + 1
    H.assert_close(y.cpu(), oy, RTOL, ATOL, "y chunks")
    H.assert_close(s.cpu(), os_, RTOL, ATOL, "s chunks")
    H.assert_close(w.cpu(), ow, RTOL, ATOL, "w chunks")


@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_frame_kernel_cell_step_at_ljspeech_prenet_dims(H, prec):
    """Taco2ProdDecoderCell.forward (ttsdec_cell_step) where the PreNet runs in the frame kernel
    (input frame handed in directly, nothing to finish) - against the oracle's single step."""
    dims = O.DecoderDims(d_mel=80, r=1, d_pre=256, d_ctx=64, h_att=128, h_dec=192)
    wts = O.random_decoder_weights(dims, seed=31, nonzero_init_state=True)
    dec = H.make_decoder(dims, wts)
    dec.precision = prec
    cell = dec.decoder_cell
    cell.dropout_source = "reference_rng"
    B, L = 35, 19
    mem = O.synthetic_memory(B, L, dims.d_ctx, lengths=[19] * 30 + [1, 4, 9, 13, 19], seed=8)
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, 1, dims.d_mel, generator=g)
    w = torch.rand(B, L, generator=g)
    w = w / w.sum(1, keepdim=True)
    ctx = torch.randn(B, dims.d_ctx, generator=g) * 0.3
    ha, ca = torch.randn(B, dims.h_att, generator=g) * 0.3, torch.randn(B, dims.h_att, generator=g) * 0.3
    hd, cd = torch.randn(B, dims.h_dec, generator=g) * 0.3, torch.randn(B, dims.h_dec, generator=g) * 0.3
    torch.manual_seed(5)
    m0, m1 = O.draw_prenet_masks(B, dims.d_pre, dims.d_pre)
    ox, (ow, octx, (oha, oca), (ohd, ocd)) = O.decoder_cell_step(x[:, -1], (w, ctx, (ha, ca), (hd, cd)), mem, wts, dims, torch.stack([m0, m1]))
    dev = "cuda:0"
    with torch.no_grad():
        torch.manual_seed(5)
        xd, cx, (w2, ctx2, ((ha2, ca2), (hd2, cd2))) = cell(
            x.to(dev), (w.to(dev), ctx.to(dev), ((ha.to(dev), ca.to(dev)), (hd.to(dev), cd.to(dev)))), mem.to(dev), None
        )
    H.assert_close(xd.cpu(), ox, RTOL, ATOL, "x_dec")
    H.assert_close(w2.cpu(), ow, RTOL, ATOL, "w")
    H.assert_close(ha2.cpu(), oha, RTOL, ATOL, "h_att")
    H.assert_close(cd2.cpu(), ocd, RTOL, ATOL, "c_dec")


# --------------------------------------------------------------------------
# round-2 additions: end-to-end glue against the reference's own Tacotron.forward vectors,
# bound (not locally packed) blobs, invalidation, range guard, argument rejection
# --------------------------------------------------------------------------
def test_tacotron_forward_end_to_end_vs_reference(golden, H):
    """ids -> Encoder2(HIP) -> Decoder(HIP, reference RNG replay) -> MelPostnet(HIP) against the vectors the
    reference's Tacotron.forward produced under the same torch.manual_seed (tacotron.py:29-56)."""
    import torch_tts_amd as T

    d, c, m = golden["meta"]["small_dims"], golden["cases"], golden["meta"]["e2e"]
    enc = T.Encoder2(golden["enc"]["emb.weight"].shape[0], dim_out=d["d_ctx"], dim_emb=golden["enc"]["emb.weight"].shape[1])
    dec = H.make_decoder(_small_dims(golden), golden["dec"], device="cpu")
    pn = T.MelPostnet(d["d_mel"], dim_hidden=d["postnet_hidden"], kernel_size=5, num_layers=d["postnet_layers"])
    model = T.Tacotron(enc, dec, pn)  # (its constructor re-initialises: load the fixture's weights afterwards)
    sd = {"encoder." + k: v for k, v in golden["enc"].items()}
    sd.update({"decoder." + k: v for k, v in golden["dec"].items()})
    sd.update({"postnet." + k: v for k, v in golden["post"].items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") or k.endswith("attention_module.bias") for k in missing), (missing, unexpected)
    model = model.cuda().eval()
    assert model.decoder.dropout_source == "reference_rng"
    with torch.no_grad():
        torch.manual_seed(m["seed"])
        y, y_post, s, out = model(golden["ids"].cuda(), golden["lengths"], max_steps=m["max_steps"])
        after = torch.rand(1)
    assert y.shape[1] == m["T"]
    H.assert_close(y.cpu(), c["e2e/y"], RTOL, ATOL, "y")
    H.assert_close(y_post.cpu(), c["e2e/y_post"], RTOL, ATOL, "y_post")
    H.assert_close(s.cpu(), c["e2e/s"], RTOL, ATOL, "s")
    H.assert_close(out["w"].cpu(), c["e2e/w"], RTOL, ATOL, "w")
    assert torch.equal(out["w"].cpu().argmax(-1), c["e2e/w"].argmax(-1))
    assert float(out["kl_loss"]) == float(c["e2e/kl_loss"]) == 0.0
    # the host generator is left where the reference leaves it: T steps x 2 Bernoulli draws
    torch.manual_seed(m["seed"])
    for _ in range(m["T"]):
        O.draw_prenet_masks(golden["ids"].shape[0], d["d_pre"], d["d_pre"])
    assert torch.equal(after, torch.rand(1))


@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_bound_blob_decodes_like_the_packing_handle(H, prec):
    """What non-source ranks do after the weight broadcast (distributed.broadcast_engine_weights): a second handle
    `bind`s a copy of the blob another handle packed, and must decode bit for bit like it."""
    from torch_tts_amd import _lib
    from torch_tts_amd.engine import Engine

    dims = O.DecoderDims()
    wts = O.random_decoder_weights(dims, seed=5, nonzero_init_state=True)
    B, L, T = 64, 40, 34
    mem = O.synthetic_memory(B, L, dims.d_ctx, seed=2).cuda()
    masks = O.synthetic_masks(T, B, dims.d_pre, seed=3)
    dec = H.make_decoder(dims, wts)
    dec.precision = prec
    ya, sa, wa, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    eng_a = dec.engine(torch.device("cuda:0"))
    eng_b = Engine(eng_a.dims, torch.device("cuda:0"))
    eng_b.bind(eng_a.blob.clone())  # (a copy: the bytes a broadcast delivers)
    eng_b.set_precision(prec)
    assert eng_b.precision() == prec
    y = torch.empty(B, T, 80, device="cuda")
    s = torch.empty(B, T, device="cuda")
    w = torch.empty(B, T, L, device="cuda")
    t_out = torch.zeros(2, dtype=torch.int32, device="cuda")
    eng_b.decode(mem, t_begin=0, n_steps=T, stop_threshold=-2.0, check_stop=True, dropout_mode=_lib.DROPOUT_MASKS,
                 masks=masks.cuda().contiguous(), seed=0, teacher=None, teacher_flags=None, y=y, s=s, w=w, t_out=t_out)
    assert t_out.tolist() == [T, 0]
    assert torch.equal(y.cpu(), ya) and torch.equal(s.cpu().unsqueeze(2), sa) and torch.equal(w.cpu(), wa)


def test_data_edits_need_invalidate_and_load_state_dict_repacks(H):
    """`param.data.copy_()` does not bump the version counter the blob key watches (the reference's checkpoint
    loader does exactly that, train_util.py:43): invalidate() makes the next forward repack; load_state_dict
    repacks by itself."""
    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=128)
    w1 = O.random_decoder_weights(dims, seed=1)
    w2 = O.random_decoder_weights(dims, seed=2)
    B, L, T = 4, 9, 6
    mem = O.synthetic_memory(B, L, dims.d_ctx).cuda()
    masks = O.synthetic_masks(T, B, dims.d_pre)
    dec = H.make_decoder(dims, w1)
    y1, _, _, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    o2, _, _ = O.decode(w2, dims, mem.cpu(), max_steps=T - 1, masks=masks)
    with torch.no_grad():
        for k, p in dec.state_dict().items():
            if k in w2:
                p.data.copy_(w2[k])
    dec.invalidate()
    y2, _, _, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    H.assert_close(y2, o2, RTOL, ATOL, "after .data edit + invalidate()")
    assert not torch.equal(y1, y2)
    dec.load_state_dict(w1, strict=False)  # hook -> repack without an explicit invalidate()
    y3, _, _, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    assert torch.equal(y3, y1)


def test_reference_style_checkpoint_loader_repacks(H):
    """The reference's own partial loader writes through param.data.copy_ (train_util.py:43), which the blob fingerprint cannot
    see; the drop-in loader (torch_tts_amd.train_util.load_state_dict) loads the same way and invalidates the packed weights."""
    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=192)
    w1 = O.random_decoder_weights(dims, seed=1, nonzero_init_state=True)
    w2 = O.random_decoder_weights(dims, seed=2, nonzero_init_state=True)
    mem = O.synthetic_memory(5, 9, dims.d_ctx, seed=3)
    masks = O.synthetic_masks(6, 5, dims.d_pre, seed=4)
    dec = H.make_decoder(dims, w1)
    y1, _, _, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=5)
    w2["no.such.parameter"] = torch.zeros(3)  # (skipped with a warning, like the reference)
    import torch_tts_amd as T

    T.train_util.load_state_dict(dec, w2)
    y2, s2, wt2, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=5)
    oy, os_, ow = O.decode({k: v for k, v in w2.items() if k != "no.such.parameter"}, dims, mem, max_steps=5, masks=masks)
    assert not torch.equal(y1, y2)
    H.assert_close(y2, oy, RTOL, ATOL, "y after the reference-style loader")
    H.assert_argmax(wt2, ow, "argmax after the reference-style loader")


def test_stop_rule_with_a_grid_larger_than_one_residency_wave(H):
    """B = 8192 -> 1024 frame-kernel workgroups (4 per CU): workgroups dispatched after the stop flag was lowered
    must still write their rows of the firing step (the frame kernel gates on t-1 <= stop_t)."""
    dims = O.DecoderDims(d_mel=80, r=1, d_pre=128, d_ctx=32, h_att=64, h_dec=64)
    wts = O.random_decoder_weights(dims, seed=21, nonzero_init_state=True)
    B, L, T = 8192, 6, 12
    mem = O.synthetic_memory(B, L, dims.d_ctx, seed=4)
    masks = O.synthetic_masks(T, B, dims.d_pre, seed=9)
    w2, thr, k = _stop_case(wts, dims, mem, masks, T, first=3)
    oy, os_, ow = O.decode(w2, dims, mem, max_steps=T - 1, stop_threshold=thr, masks=masks)
    assert oy.shape[1] == k + 1
    for prec in ("f32", "split_f16"):
        dec = H.make_decoder(dims, w2, stop_threshold=thr)
        dec.precision = prec
        for _ in range(3):
            y, s, w, fired = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
            assert fired and y.shape[1] == k + 1
            assert torch.isfinite(y).all() and torch.isfinite(s).all()
            H.assert_close(y, oy, RTOL, ATOL, f"y ({prec})")
            H.assert_close(s, os_, RTOL, ATOL, f"s ({prec})")


def test_split_f16_range_guard(H):
    """A weight at or beyond the fp16 range keeps the handle on exact fp32 (reported by precision()); an activation
    beyond it (a teacher frame of 7e4) is saturated, never inf/NaN, and reported by the call."""
    from torch_tts_amd import _lib

    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=128)
    wts = O.random_decoder_weights(dims, seed=3)
    B, L, T = 5, 7, 6
    mem = O.synthetic_memory(B, L, dims.d_ctx).cuda()
    masks = O.synthetic_masks(T, B, dims.d_pre)
    big = {k: v.clone() for k, v in wts.items()}
    big["decoder_cell.decoder_rnn.weight_hh"][3, 5] = 7.0e4
    dec = H.make_decoder(dims, big)
    dec.precision = "split_f16"
    assert dec.engine(torch.device("cuda:0")).precision() == "f32", "an out-of-range weight must keep the GEMMs on fp32"
    oy, _, _ = O.decode(big, dims, mem.cpu(), max_steps=T - 1, masks=masks)
    y, _, _, _ = H.run_decoder_with_masks(dec, mem, masks, max_steps=T - 1)
    H.assert_close(y, oy, RTOL, ATOL, "y with a 7e4 weight (fp32 path)")
    # activation out of range: teacher frame of 7e4 in split mode
    dec = H.make_decoder(dims, wts)
    dec.precision = "split_f16"
    eng = dec.engine(torch.device("cuda:0"))
    assert eng.precision() == "split_f16"
    x = torch.zeros(B, T, 80)
    x[2, 1, 7] = 7.0e4
    y = torch.empty(B, T, 80, device="cuda"); s = torch.empty(B, T, device="cuda"); w = torch.empty(B, T, L, device="cuda")
    t_out = torch.zeros(2, dtype=torch.int32, device="cuda")
    eng.decode(mem, t_begin=0, n_steps=T, stop_threshold=-2.0, check_stop=False, dropout_mode=_lib.DROPOUT_MASKS,
               masks=masks.cuda().contiguous(), seed=0, teacher=x.cuda(), teacher_flags=torch.ones(T, dtype=torch.uint8, device="cuda"),
               y=y, s=s, w=w, t_out=t_out)
    done, flags = t_out.tolist()
    assert done == T and flags & 2, "the saturation must be reported in T_out[1] bit 1"
    assert torch.isfinite(y).all() and torch.isfinite(s).all() and torch.isfinite(w).all()
    with pytest.raises(RuntimeError, match="fp16 range"), torch.no_grad():
        dec(mem, None, x.cuda())
    dec.precision = "f32"  # the exact path takes the same input without complaint and matches the oracle
    oy, _, _ = O.decode(wts, dims, mem.cpu(), masks=masks, x=x, p_no_forcing=None)
    yt, _, _, _ = H.run_decoder_with_masks(dec, mem, masks, x=x)
    H.assert_close(yt, oy, RTOL, ATOL, "teacher 7e4, fp32")


def test_argument_rejection_through_the_c_abi(H):
    import ctypes as C

    import torch_tts_amd as T
    from torch_tts_amd import _lib
    from torch_tts_amd.engine import Engine, EngineDims

    lib = _lib.load()
    h = C.c_void_p()
    assert lib.ttsdec_create(C.byref(EngineDims(d_ctx=2048).to_c()), C.byref(h)) == _lib.ERR_DIMS  # attention covers d_ctx <= 1024
    # Philox draws one keep bit per unit: only p = 0.5
    eng = Engine(EngineDims(p_dropout=0.25), torch.device("cuda:0"))
    eng.pack([None] * eng.num_weight_tensors())
    mem = torch.zeros(2, 3, 512, device="cuda")
    y = torch.empty(2, 2, 80, device="cuda"); s = torch.empty(2, 2, device="cuda"); w = torch.empty(2, 2, 3, device="cuda")
    with pytest.raises(_lib.TtsdecError) as ei:
        eng.decode(mem, t_begin=0, n_steps=2, stop_threshold=-2.0, check_stop=True, dropout_mode=_lib.DROPOUT_PHILOX, masks=None, seed=1,
                   teacher=None, teacher_flags=None, y=y, s=s, w=w, t_out=torch.zeros(2, dtype=torch.int32, device="cuda"))
    assert ei.value.code == _lib.ERR_INVALID_ARG
    # token ids outside the table: IndexError like nn.Embedding, never an out-of-bounds read
    enc = T.Encoder2(40, dim_out=64, dim_emb=32).cuda().eval()
    ids = torch.randint(1, 40, (2, 9)).cuda()
    ids[1, 4] = 40
    with pytest.raises(IndexError), torch.no_grad():
        enc(ids, torch.tensor([9, 9]))
    te = T.vits2.TextEncoder(50, 192, 192, 768, 2, 2, 3, 0.1).cuda().eval()
    bad = torch.randint(0, 50, (2, 7)).cuda()
    bad[0, 0] = -1
    with pytest.raises(IndexError), torch.no_grad():
        te(bad, torch.tensor([7, 7]).cuda())


# --------------------------------------------------------------------------
# every step order the library can run (the defaults pick one per batch size / mode; the rest are reachable through
# the tuning options of include/ttsdec.h, here preset through TTSDEC_OPTIONS, which every new handle reads) must give
# the reference's results
# --------------------------------------------------------------------------
_STEP_OPTIONS = [
    {}, {"overlap": 0}, {"overlap": 1}, {"overlap": 2}, {"graph": 0},
    {"overlap": 2, "graph": 0}, {"chunk_a": 0}, {"chunk_b": 0},
    {"chunk_a": 0, "chunk_b": 0, "overlap": 2}, {"proj_regw": 0},
    {"head_proj": 1}, {"head_proj": 1, "overlap": 1}, {"head_proj": 1, "graph": 0},
    {"head_proj": 0},
    {"query_role": 0}, {"query_role": 1, "overlap": 2}, {"query_role": 1, "overlap": 2, "head_proj": 0, "graph": 0}, {"query_role": 1, "chunk_a": 0},
    # the whole step as one launch (split-fp16, up to 256 utterances; elsewhere the option means level 2)
    {"overlap": 3, "query_role": 1}, {"overlap": 3, "query_role": 1, "graph": 0}, {"overlap": 3, "query_role": 1, "chunk_a": 0, "chunk_b": 0},
]


@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_all_step_orders_and_layouts_vs_oracle(H, prec, monkeypatch):
    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=192)  # (fused launches, chunked planes and the
    wts = O.random_decoder_weights(dims, seed=21, nonzero_init_state=True)    #  register-weight projection all apply)
    T_ = 18  # >= 15 steps: the captured-graph path (unless switched off)
    opts = _STEP_OPTIONS
    for B in (3, 40, 70):  # stand-alone small tile / 64x8 lean tile / 64x16 lean tile of the two-role launches
        mem = O.synthetic_memory(B, 11, dims.d_ctx, lengths=[11] * (B - 1) + [4], seed=7)
        masks = O.synthetic_masks(T_, B, dims.d_pre, seed=9)
        oy, os_, ow = O.decode(wts, dims, mem, max_steps=T_ - 1, masks=masks)
        for opt in opts:
            with monkeypatch.context() as mp:
                mp.setenv("TTSDEC_OPTIONS", ",".join(f"{k}={v}" for k, v in opt.items()))
                dec = H.make_decoder(dims, wts)  # a new module = a new handle, which reads the options
                dec.precision = prec
                eng = dec.engine(torch.device("cuda:0"))
                for k, v in opt.items():
                    assert eng.get_option(k) == v, (k, v)
                y, s, w, fired = H.run_decoder_with_masks(dec, mem, masks, max_steps=T_ - 1)
                if "overlap" in opt and "chunk_a" not in opt:  # the option really selects the launch sequence
                    from torch_tts_amd import _lib
                    names = set(eng.profile_step(mem.cuda(), 1, _lib.DROPOUT_OFF, None, 0))
                    lv = opt["overlap"]
                    if lv == 3 and prec == "split_f16":
                        assert names == {"step"}, (opt, names)
                    else:
                        assert "step" not in names, (opt, names)
                        fa = "prenet+lstm_att" in names or "proj+prenet+lstm_att" in names
                        assert fa == (lv >= 1) and any(n.endswith("attention+lstm_dec") for n in names) == (lv >= 2), (opt, names)
                        if opt.get("query_role") == 1:
                            assert "query+attention+lstm_dec" in names and "query" not in names, (opt, names)
                        if opt.get("head_proj") == 1 and prec == "split_f16":
                            assert "proj+prenet+lstm_att" in names and "proj" not in names, (opt, names)
            what = f"B={B} {prec} {opt}"
            assert not fired and y.shape == oy.shape, what
            H.assert_close(y, oy, RTOL, ATOL, "y " + what)
            H.assert_close(s, os_, RTOL, ATOL, "s " + what)
            H.assert_close(w, ow, RTOL, ATOL, "w " + what)
            H.assert_argmax(w, ow, "argmax " + what)


@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_two_role_launches_with_more_producers_than_the_chip_holds(H, prec, monkeypatch):
    """640 utterances with the two-role launches forced on (the defaults stop at 320): the attention role's 640 workgroups exceed
    the chip's 512 resident slots, so the grid dispatches in waves - producers first (lowest block ids, waiting for nothing), the
    LSTM role's workgroups as slots free up.  Same results as one role per launch and as the oracle; no time-out."""
    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=192)
    wts = O.random_decoder_weights(dims, seed=13, nonzero_init_state=True)
    B, T_ = 640, 5
    mem = O.synthetic_memory(B, 7, dims.d_ctx, lengths=[7] * (B - 3) + [5, 2, 1], seed=2)
    masks = O.synthetic_masks(T_, B, dims.d_pre, seed=8)
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T_ - 1, masks=masks)
    outs = {}
    for lv in (0, 1, 2):
        with monkeypatch.context() as mp:
            mp.setenv("TTSDEC_OPTIONS", f"overlap={lv}")
            dec = H.make_decoder(dims, wts)
            dec.precision = prec
            y, s, w, fired = H.run_decoder_with_masks(dec, mem, masks, max_steps=T_ - 1)
        assert not fired
        H.assert_close(y, oy, RTOL, ATOL, f"y overlap={lv} {prec}")
        H.assert_close(s, os_, RTOL, ATOL, f"s overlap={lv} {prec}")
        H.assert_argmax(w, ow, f"argmax overlap={lv} {prec}")
        outs[lv] = (y, s, w)
    # (no bitwise comparison between the levels: a gated LSTM walks its K segments in another order than the whole cell)
    # 330 utterances on the defaults: level 2 with the query as a job of the attention role's 330 workgroups (all resident, lowest
    # block ids), the decoder LSTM's workgroups arriving in two waves behind them
    B2 = 330
    mem2 = O.synthetic_memory(B2, 7, dims.d_ctx, lengths=[7] * (B2 - 2) + [3, 1], seed=12)
    masks2 = O.synthetic_masks(T_, B2, dims.d_pre, seed=18)
    oy2, os2, ow2 = O.decode(wts, dims, mem2, max_steps=T_ - 1, masks=masks2)
    dec = H.make_decoder(dims, wts)
    dec.precision = prec
    y, s, w, fired = H.run_decoder_with_masks(dec, mem2, masks2, max_steps=T_ - 1)
    assert not fired
    if prec == "split_f16":
        names = set(dec.engine(torch.device("cuda:0")).profile_step(mem2.cuda(), 1, 0, None, 0))
        assert "query+attention+lstm_dec" in names, names
    H.assert_close(y, oy2, RTOL, ATOL, f"y B=330 {prec}")
    H.assert_close(s, os2, RTOL, ATOL, f"s B=330 {prec}")
    H.assert_argmax(w, ow2, f"argmax B=330 {prec}")


def test_set_option_on_a_live_handle_switches_the_launch_sequence(H):
    """ttsdec_set_option between two calls on ONE handle (the captured graph is dropped): same results, other launches."""
    from torch_tts_amd import _lib
    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=192)
    wts = O.random_decoder_weights(dims, seed=4, nonzero_init_state=True)
    B, T_ = 96, 20
    mem = O.synthetic_memory(B, 9, dims.d_ctx, seed=3)
    masks = O.synthetic_masks(T_, B, dims.d_pre, seed=5)
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T_ - 1, masks=masks)
    dec = H.make_decoder(dims, wts)
    dec.precision = "split_f16"
    eng = dec.engine(torch.device("cuda:0"))
    seen = []
    for opt in ({"overlap": 2}, {"head_proj": 0}, {"overlap": 1}, {"overlap": 0}, {"overlap": -1, "head_proj": -1}):
        for k, v in opt.items():
            eng.set_option(k, v)
        y, s, w, fired = H.run_decoder_with_masks(dec, mem, masks, max_steps=T_ - 1)
        seen.append(tuple(eng.profile_step(mem.cuda(), 1, _lib.DROPOUT_OFF, None, 0)))
        H.assert_close(y, oy, RTOL, ATOL, f"y {opt}")
        H.assert_argmax(w, ow, f"argmax {opt}")
    assert any(n.endswith("attention+lstm_dec") for n in seen[0]) and "proj" in seen[1] and not any(n.endswith("attention+lstm_dec") for n in seen[3]), seen
    with pytest.raises(KeyError):
        eng.set_option("no_such_option", 1)


def test_role_timeout_at_the_default_spin_limit_is_bounded_in_time(H):
    """A call whose producer role never signals costs about ONE spin limit, not steps x gates of them (common.h role_poll: once
    Ctrl::range_err bit 1 is set every later gate of the call returns at once).  Default spin limit (~1-4 ms), a 256-step chunk
    (Decoder.forward's chunk_steps): the whole call, flag read-back included, stays far below a second.  Round 3's limit
    (2^17 polls per gate, no short-circuit) would have spun for tens of seconds here."""
    import time
    from torch_tts_amd import _lib
    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=192)
    wts = O.random_decoder_weights(dims, seed=4, nonzero_init_state=True)
    B, T_ = 70, 256
    memd = O.synthetic_memory(B, 9, dims.d_ctx, seed=3).cuda()
    for prec in ("f32", "split_f16"):
        dec = H.make_decoder(dims, wts)
        dec.precision = prec
        eng = dec.engine(torch.device("cuda:0"))
        eng.set_option("overlap", 2)
        y = torch.empty(B, T_, 80, device="cuda"); s = torch.empty(B, T_, device="cuda"); w = torch.empty(B, T_, 9, device="cuda")
        t_out = torch.zeros(2, dtype=torch.int32, device="cuda")

        def call():
            eng.decode(memd, t_begin=0, n_steps=T_, stop_threshold=-2.0, check_stop=True, dropout_mode=_lib.DROPOUT_OFF, masks=None,
                       seed=0, teacher=None, teacher_flags=None, y=y, s=s, w=w, t_out=t_out)
            torch.cuda.synchronize()

        call()  # (graph capture, code load)
        assert t_out.tolist()[1] & 4 == 0
        eng.set_option("debug_flags", 1)  # the frame role stays silent: every attention-LSTM workgroup runs into its gate
        call()  # (the option dropped the graph: capture again outside the timed call)
        t0 = time.perf_counter()
        call()
        dt = time.perf_counter() - t0
        assert t_out.tolist()[1] & 4, (prec, t_out.tolist())
        assert dt < 0.5, f"a timed-out {T_}-step call took {dt:.2f} s ({prec})"
        eng.set_option("debug_flags", 0)
        call()
        assert t_out.tolist() == [T_, 0]


def test_role_timeout_flags_the_call_and_the_module_falls_back(H):
    """The bounded spin of the two-role launches (common.h role_wait): a producer role that never signals (test hook
    TTSDEC_OPT_DEBUG_FLAGS) makes the consumers give up - T_out[1] bit 2, the grid drains - and Decoder.forward does
    not return that call's outputs: it switches the engine to one role per launch (option overlap = 0), warns and
    repeats the call."""
    from torch_tts_amd import _lib
    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=192)  # (h_dec + d_ctx = 256: the projection head role applies)
    wts = O.random_decoder_weights(dims, seed=4, nonzero_init_state=True)
    B, T_ = 70, 4
    mem = O.synthetic_memory(B, 9, dims.d_ctx, seed=3)
    masks = O.synthetic_masks(T_, B, dims.d_pre, seed=5)
    oy, os_, ow = O.decode(wts, dims, mem, max_steps=T_ - 1, masks=masks)
    ny, ns, nw = O.decode(wts, dims, mem, max_steps=T_ - 1, dropout="off")
    memd = mem.cuda()
    for prec in ("f32", "split_f16"):
        for bit in (1, 2, 4, 8):  # the frame role / the attention role / the projection head role / the attention LSTM's tiles stay silent
            if bit == 8 and prec == "f32":
                continue  # (the one-launch step exists in split-fp16 mode only)
            dec = H.make_decoder(dims, wts)
            dec.precision = prec
            eng = dec.engine(torch.device("cuda:0"))
            eng.set_option("overlap", 3 if bit == 8 else 2)
            if bit == 8:
                eng.set_option("query_role", 1)
            eng.set_option("spin_limit", 64)
            eng.set_option("debug_flags", bit)
            y = torch.empty(B, T_, 80, device="cuda"); s = torch.empty(B, T_, device="cuda"); w = torch.empty(B, T_, 9, device="cuda")
            t_out = torch.zeros(2, dtype=torch.int32, device="cuda")
            eng.decode(memd, t_begin=0, n_steps=T_, stop_threshold=-2.0, check_stop=True, dropout_mode=_lib.DROPOUT_MASKS, masks=masks.cuda().contiguous(),
                       seed=0, teacher=None, teacher_flags=None, y=y, s=s, w=w, t_out=t_out)
            assert t_out.tolist()[1] & 4, (prec, bit, t_out.tolist())
            dec.dropout_source = "off"
            with pytest.warns(RuntimeWarning, match="producer role"), torch.no_grad():
                y2, s2, w2 = dec(memd, None, None, max_steps=T_ - 1)
            assert eng.get_option("overlap") == 0 and eng.get_option("head_proj") == 0
            H.assert_close(y2.cpu(), ny, RTOL, ATOL, f"y after the fallback ({prec}, bit {bit})")
            H.assert_argmax(w2.cpu(), nw, "argmax after the fallback")
            # the hook off again, two-role launches back on: the same handle decodes correctly
            eng.set_option("debug_flags", 0)
            eng.set_option("spin_limit", -1)
            eng.set_option("overlap", 2)
            eng.set_option("head_proj", -1)
            y3, s3, w3, fired = H.run_decoder_with_masks(dec, mem, masks, max_steps=T_ - 1)
            H.assert_close(y3, oy, RTOL, ATOL, "y after a timed-out call")
            H.assert_argmax(w3, ow, "argmax after a timed-out call")
