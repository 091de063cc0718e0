"""GPU tests of the grad-enabled / training-mode device path (torch-tts_amd/autograd_path.py): SURVEY.md 8b's
"autograd must flow when x is given".  Forward values and gradients are checked against the CPU oracle (whose
functions are differentiable torch code), eval-mode results against the HIP hot path; the training-only
randomness (zoneout masks, energy noise, postnet dropout / batch statistics) is checked by its properties."""
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


def _replay_masks(seed, T, B, d_pre):
    torch.manual_seed(seed)
    return torch.stack([torch.stack(O.draw_prenet_masks(B, d_pre, d_pre)) for _ in range(T)])


def test_eval_mode_grad_enabled_forward_and_gradients(H):
    """eval() but grad enabled: the torch-op path must give the HIP path's values (same host-RNG replay of the PreNet
    masks) and the oracle's gradients."""
    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=96)
    wts = O.random_decoder_weights(dims, seed=11, nonzero_init_state=True)
    B, L, Tx, seed = 5, 9, 7, 31
    mem = O.synthetic_memory(B, L, dims.d_ctx, lengths=[9, 9, 4, 7, 1])
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, Tx, 80, generator=g) * 0.5
    dec = H.make_decoder(dims, wts)  # eval mode, parameters require grad
    torch.manual_seed(seed)
    y, s, w = dec(mem.cuda(), None, x.cuda(), 0, None)  # grad enabled -> autograd path
    assert y.requires_grad and y.shape == (B, Tx, 80)
    with torch.no_grad():
        torch.manual_seed(seed)
        yh, sh, wh = dec(mem.cuda(), None, x.cuda(), 0, None)  # HIP path
    H.assert_close(y.detach().cpu(), yh.cpu(), RTOL, ATOL, "y vs HIP path")
    H.assert_close(w.detach().cpu(), wh.cpu(), RTOL, ATOL, "w vs HIP path")
    # gradients against the oracle (CPU, same masks)
    masks = _replay_masks(seed, Tx, B, dims.d_pre)
    ow = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in wts.items()}
    oy, os_, oww = O.decode(ow, dims, mem, masks=masks, x=x, p_no_forcing=None)
    H.assert_close(y.detach().cpu(), oy.detach(), RTOL, ATOL, "y vs oracle")
    loss = (y * y).sum() + s.sum() + (w * w).sum()
    loss.backward()
    oloss = (oy * oy).sum() + os_.sum() + (oww * oww).sum()
    oloss.backward()
    sd = dict(dec.named_parameters())
    checked = 0
    for k in ("fc_mel.weight", "fc_stop.bias", "decoder_cell.decoder_rnn.weight_hh", "decoder_cell.attention_rnn.weight_ih",
              "decoder_cell.attention_module.query_layer.weight", "decoder_cell.pre_net.layers.0.weight", "decoder_cell.initial_decoder_h.1"):
        ga, gb = sd[k].grad.cpu(), ow[k].grad
        scale = float(gb.abs().max())
        assert scale > 0 and float((ga - gb).abs().max()) <= 2e-3 * scale, (k, float((ga - gb).abs().max()), scale)
        checked += 1
    assert checked == 7


def test_training_mode_semantics(H):
    """train(): zoneout picks old-vs-new per UNIT for the whole batch (rnn.py:26-35), the attention energies get
    noise (attention.py:111-112), the PreNet dropout stays on, backward works."""
    import torch_tts_amd as T

    dims = O.DecoderDims(d_mel=80, d_pre=128, d_ctx=64, h_att=128, h_dec=96)
    wts = O.random_decoder_weights(dims, seed=12, nonzero_init_state=True)
    dec = H.make_decoder(dims, wts).train()
    cell = dec.decoder_cell
    B, L = 6, 8
    mem = O.synthetic_memory(B, L, dims.d_ctx).cuda()
    # one zoneout cell on its own, p = 0.5
    rnn = cell.attention_rnn
    rnn.p_zoneout = 0.5
    hp, cp = torch.randn(B, 128, device="cuda"), torch.randn(B, 128, device="cuda")
    h, c = rnn(torch.randn(B, 128 + 64, device="cuda"), (hp, cp))
    kept = h == hp  # [B, H]
    assert bool((kept.all(0) | (~kept).all(0)).all()), "a unit is zoned out for every utterance or for none"
    frac = float(kept.all(0).float().mean())
    assert 0.25 < frac < 0.75, frac
    rnn.p_zoneout = 0.1
    # whole decoder, teacher forced
    x = torch.randn(B, 6, 80, device="cuda") * 0.5
    y1, s1, w1 = dec(mem, None, x, 0, 0.1)
    y2, s2, w2 = dec(mem, None, x, 0, 0.1)
    assert y1.shape == (B, 6, 80) and s1.shape == (B, 6, 1) and w1.shape == (B, 6, L)
    assert not torch.equal(w1, w2), "energy noise / dropout / zoneout must differ between two training passes"
    assert float((w1.sum(-1) - 1).abs().max()) < 1e-4
    (y1.square().mean() + s1.mean()).backward()
    grads = [p.grad for p in dec.parameters() if p.grad is not None]
    assert len(grads) >= 15 and all(bool(torch.isfinite(g_).all()) for g_ in grads)
    # free-running in training mode (no teacher) also works and honours max_steps
    with torch.no_grad():
        y3, _, _ = dec(mem, None, None, 4, None)
    assert y3.shape == (B, 5, 80)
    # Tacotron.forward in training mode end to end (tacotron.py:29-56 with x given), incl. the postnet
    cfg = {"text": {"alphabet": "abcdefghij"}, "audio": {"num_mels": 80},
           "model": {"encoder": {"dim_emb": 32, "dim_out": 64}, "decoder": {"type": "tacotron2prod", "r": 1, "dim_pre": 128, "dim_att": 128, "dim_rnn": [128, 96]},
                     "postnet": {"type": "tacotron2", "dim_hidden": 64, "num_layers": 2}}}
    torch.manual_seed(0)
    model = T.build_tacotron(cfg).cuda().train()
    ids = torch.randint(1, 10, (3, 7)).cuda()
    lens = torch.tensor([7, 5, 3])
    xt = torch.randn(3, 9, 80, device="cuda")
    rm0 = model.postnet.conv[0][1].running_mean.clone()
    yy, yp, ss, out = model(ids, lens, xt)
    assert yy.shape == (3, 9, 80) and yp.shape == yy.shape and out["w"].shape == (3, 9, 7)
    (yp - xt).abs().mean().backward()
    assert model.encoder.emb.weight.grad is not None and model.postnet.fc_out.weight.grad is not None
    assert not torch.equal(rm0, model.postnet.conv[0][1].running_mean), "BatchNorm runs on batch statistics in training"


def test_postnets_grad_enabled_eval_match_golden(golden, golden_taco2, H):
    """eval-mode postnets with grad enabled run as torch ops: same vectors as the reference; Conv1dFix as one conv1d."""
    d, c = golden["meta"]["small_dims"], golden["cases"]
    pn = H.make_postnet(d["d_mel"], d["postnet_hidden"], d["postnet_layers"], golden["post"])
    out = pn(c["infer/y"].cuda())
    assert out.requires_grad
    H.assert_close(out.detach().cpu(), c["infer/y_post"], RTOL, ATOL, "MelPostnet (torch ops)")
    g2 = golden_taco2
    d2 = g2["meta"]["dims"]
    pn2 = H.make_postnet2(d2["d_mel"], d2["postnet_hidden"], d2["postnet_layers"], g2["post"])
    cin, cout = g2["cases"]["unit/post_y"], g2["cases"]["unit/post_out"]
    out2 = pn2(cin.cuda())
    assert out2.requires_grad
    H.assert_close(out2.detach().cpu(), cout, RTOL, ATOL, "MelPostnet2 (torch ops, Conv1dFix as conv1d)")
    with torch.no_grad():
        hip = pn2(cin.cuda())
    H.assert_close(out2.detach().cpu(), hip.cpu(), RTOL, ATOL, "MelPostnet2 torch ops vs HIP path")
