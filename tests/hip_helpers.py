"""Shared helpers for the GPU parity tests: build the product modules (the drop-in
mirrors of the reference's classes) from an oracle-style weight dict."""
import torch

import torch_tts_amd as T
from oracle import tacotron_oracle as O


def make_decoder(dims: O.DecoderDims, wts, device="cuda:0", stop_threshold=-2.0):
    cell = T.Taco2ProdDecoderCell(dims.d_ctx, dims.d_mel, dims.r, [dims.h_att, dims.h_dec], dim_pre=dims.d_pre, dim_att=dims.h_att,
                                  p_zoneout=dims.p_zoneout)
    dec = T.Decoder(cell, dims.r, dims.d_mel, stop_threshold=stop_threshold)
    missing, unexpected = dec.load_state_dict(wts, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("attention_module.bias") for k in missing), missing
    return dec.to(device).eval()


def make_taco2_decoder(dims: O.DecoderDims, wts, device="cuda:0", stop_threshold=-2.0):
    cell = T.Taco2DecoderCell(dims.d_ctx, dims.d_mel, dims.r, [dims.h_att, dims.h_dec], dim_pre=dims.d_pre, p_zoneout=dims.p_zoneout)
    dec = T.Decoder(cell, dims.r, dims.d_mel, stop_threshold=stop_threshold)
    missing, unexpected = dec.load_state_dict(wts, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("attention_module.bias") for k in missing), missing
    return dec.to(device).eval()


def flat_masks(m0, m1):
    """Per step: layer-0 mask [B, 128] then layer-1 mask [B, d_pre], flattened (include/ttsdec.h)."""
    T_ = m0.shape[0]
    return torch.cat([m0.reshape(T_, -1), m1.reshape(T_, -1)], dim=1).contiguous()


def make_postnet(d_mel, hidden, layers, wts, device="cuda:0", k=5):
    pn = T.MelPostnet(d_mel, dim_hidden=hidden, kernel_size=k, num_layers=layers)
    missing, unexpected = pn.load_state_dict(wts, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("num_batches_tracked") for k in missing), missing
    return pn.to(device).eval()


def run_decoder_with_masks(dec, memory, masks, *, max_steps=0, x=None, flags=None, device="cuda:0"):
    """Drives ttsdec_decode directly with injected masks (and teacher flags), the way
    Decoder.forward does for one bounded call."""
    from torch_tts_amd import _lib

    memory = memory.to(device)
    B, L, _ = memory.shape
    r, dm = dec.r, dec.dim_mel
    eng = dec.engine(memory.device)
    if x is not None:
        n = x.shape[1] // r
        teacher = x[:, : n * r].to(device).contiguous()
        fl = torch.ones(n, dtype=torch.uint8) if flags is None else torch.cat([torch.as_tensor(flags, dtype=torch.uint8), torch.ones(1, dtype=torch.uint8)])
        fl = fl.to(device)
        check = False
    else:
        n = max_steps + 1
        teacher, fl, check = None, None, True
    y = torch.empty(B, n * r, dm, device=device)
    s = torch.empty(B, n * r, device=device)
    w = torch.empty(B, n, L, device=device)
    t_out = torch.zeros(2, dtype=torch.int32, device=device)
    mode = _lib.DROPOUT_MASKS if masks is not None else _lib.DROPOUT_OFF
    eng.decode(memory, t_begin=0, n_steps=n, stop_threshold=float(dec.stop_threshold), check_stop=check, dropout_mode=mode,
               masks=None if masks is None else masks[:n].to(device).contiguous(), seed=0, teacher=teacher, teacher_flags=fl,
               y=y, s=s, w=w, t_out=t_out)
    done, fired = t_out.tolist()
    return y[:, : done * r].cpu(), s[:, : done * r].unsqueeze(2).cpu(), w[:, :done].cpu(), bool(fired)


def make_postnet2(d_mel, hidden, layers, wts, device="cuda:0"):
    pn = T.MelPostnet2(d_mel, dim_hidden=hidden, num_layers=layers)
    missing, unexpected = pn.load_state_dict(wts, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("num_batches_tracked") for k in missing), missing
    return pn.to(device).eval()


def assert_close(a, b, rtol=1e-4, atol=1e-5, what=""):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: max abs err {float(err.max()):.3e}, {int(bad.sum())} of {bad.numel()} outside rtol={rtol} atol={atol}"


def assert_argmax(w, ow, what="attention argmax", tie=2e-6):
    """argmax(w) must equal argmax(ow) bit for bit - except on rows where the reference's own two largest weights
    differ by less than `tie` (absolute; the weights of a row sum to 1): there any fp32 evaluation order decides the
    index, the reference's included.  Prints what it tolerated."""
    a, b = w.argmax(-1), ow.argmax(-1)
    bad = a != b
    if not bool(bad.any()):
        return
    top2 = ow[bad].topk(2, dim=-1).values
    gap = (top2[:, 0] - top2[:, 1]).abs()
    err = (w[bad] - ow[bad]).abs().max(dim=-1).values
    print(f"{what}: {int(bad.sum())} of {bad.numel()} rows differ; reference top-2 gaps {gap.tolist()[:8]}, max |w - ow| there {err.tolist()[:8]}")
    assert bool((gap < tie).all()), f"{what}: argmax differs on a row whose top-2 gap is {float(gap.max()):.3e}"
