"""Autograd-capable device path of the decoder, for grad-enabled calls (training, or anything that
differentiates through the model): composable torch ops on the modules' own parameters, on the ROCm
device the tensors live on.  SURVEY.md section 8b: "autograd must flow when x is given - the kernel path
is inference-only, falling back to composable torch ops otherwise".

This is NOT the hot path (that is the HIP library, forward only) and NOT the oracle (tests/ only): it is
what `TacotronTask.train_forward` (tacotron/tacotron_lightning.py:50-99) needs to run on the drop-in.  It
follows the reference's training semantics:

  * PreNet dropout always on                                   modules/modules.py:37-41
  * LSTM zoneout: eval = blend, training = per-UNIT Bernoulli masks shared by the batch    modules/rnn.py:24-39
  * attention energies get unit Gaussian noise in training     modules/attention.py:111-112
  * teacher forcing with p_no_forcing                          decoder.py:61-66
  * Postnet BatchNorm in batch-statistics mode + dropout 0.1   modules/modules.py:178-184

Randomness: with ``dropout_source == "reference_rng"`` the PreNet keep-masks and the teacher-forcing coin flips
replay the reference's host-generator draws (rng.MaskStream), so an eval-mode grad-enabled forward equals the
HIP path and the reference under the same seed; the training-only draws (zoneout masks, energy noise, postnet
dropout) come from the device generator, as they do when the reference itself runs on a GPU.
CPU tensors are refused like everywhere else in this package."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

from .rng import MaskStream

Tensor = torch.Tensor


_WARNED = set()


def warn_eval_on_autograd_path(module) -> None:
    """An eval-mode module that lands here only because grad mode is on (its parameters require grad and the call is not
    under torch.no_grad()) gets correct results - but from step-by-step torch ops, orders of magnitude slower than the HIP
    path.  Say so once per class."""
    if not module.training and type(module).__name__ not in _WARNED:
        _WARNED.add(type(module).__name__)
        import warnings

        warnings.warn(f"{type(module).__name__} is in eval mode but grad is enabled, so it runs the differentiable torch-op path, not the "
                      "HIP kernels: wrap inference in torch.no_grad() (as the reference's run_inference_step does) for the fast path",
                      RuntimeWarning, stacklevel=3)


def _require_device(t: Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on a ROCm device (got {t.device}): this package has no CPU path")


def _isru(x: Tensor) -> Tensor:  # activations.py:9-10
    return x / torch.sqrt(1.0 + x * x)


def _isru_sigmoid(x: Tensor) -> Tensor:  # activations.py:5-6
    return 0.5 * (1.0 + _isru(0.5 * x))


def prenet(pre_net, x: Tensor, keep: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """two x [Linear -> ReLU -> dropout(p, always)]; keep = injected keep-masks of the two layers, else drawn on x's device"""
    p = float(pre_net.p_dropout)
    for i, layer in enumerate(pre_net.layers):
        x = torch.relu(layer(x))
        if keep is not None:
            x = x * keep[i].to(x.dtype) * (1.0 / (1.0 - p))
        else:
            x = torch.dropout(x, p, True)
    return x


def lstm_zoneout(cell, x: Tensor, hc: Tuple[Tensor, Tensor], training: bool) -> Tuple[Tensor, Tensor]:
    h_prev, c_prev = hc
    gates = F.linear(x, cell.weight_ih, cell.bias_ih) + F.linear(h_prev, cell.weight_hh, cell.bias_hh)
    gi, gf, gg, go = gates.chunk(4, dim=1)  # PyTorch gate order i, f, g, o
    c = torch.sigmoid(gf) * c_prev + torch.sigmoid(gi) * torch.tanh(gg)
    h = torch.sigmoid(go) * torch.tanh(c)
    pz = cell.p_zoneout
    if training:
        if pz:
            keep_old_h = torch.rand(cell.hidden_size, device=h.device) < pz
            keep_old_c = torch.rand(cell.hidden_size, device=c.device) < pz
            h = torch.where(keep_old_h, h_prev, h)
            c = torch.where(keep_old_c, c_prev, c)
    else:
        h = pz * h_prev + (1.0 - pz) * h
        c = pz * c_prev + (1.0 - pz) * c
    return h, c


def monotonic_attention(att, h: Tensor, w: Tensor, memory: Tensor, training: bool) -> Tensor:
    q = att.query_layer(h)
    e = torch.bmm(memory, q.unsqueeze(2)).squeeze(2)
    if training:
        e = e + att.sigmoid_noise * torch.randn_like(e)
    last = torch.zeros_like(e)
    last[:, -1] = 1.0
    e = e * (1.0 - last) + 1e4 * last  # e[:, -1] = 1e4, without an in-place write into the graph
    p0 = _isru_sigmoid(e)
    stay, move = w * p0, w * (1.0 - p0)
    return stay + F.pad(move[:, :-1], (1, 0))


def prod_cell_step(cell, x_frame: Tensor, state, memory: Tensor, keep=None):
    """Taco2ProdDecoderCell.forward, decoder_cell.py:180-195; x_frame [B, d_mel]"""
    w, ctx, (hc_att, hc_dec) = state
    tr = cell.training
    x_pre = prenet(cell.pre_net, x_frame, keep)
    hc_att = lstm_zoneout(cell.attention_rnn, torch.cat([x_pre, ctx], dim=1), hc_att, tr)
    w = monotonic_attention(cell.attention_module, hc_att[0], w, memory, tr)
    ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
    hc_dec = lstm_zoneout(cell.decoder_rnn, torch.cat([hc_att[0], ctx], dim=1), hc_dec, tr)
    return torch.cat([hc_dec[0], ctx], dim=1), ctx, (w, ctx, (hc_att, hc_dec))


def taco2_cell_step(cell, x_frame: Tensor, state, memory: Tensor, keep=None):
    """Taco2DecoderCell.forward, decoder_cell.py:110-140: context from the PREVIOUS weights feeds both stacked
    LSTMs; attention and output read cat[h0, h1, zeros]"""
    w, hcs = state
    tr = cell.training
    x = prenet(cell.pre_net, x_frame, keep)
    ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
    new_hcs, outs = [], []
    for rnn, hc in zip(cell.decoder_rnn_list, hcs):
        hc = lstm_zoneout(rnn, torch.cat([x, ctx], dim=1), hc, tr)
        new_hcs.append(hc)
        outs.append(hc[0])
        x = hc[0]
    x_dec = torch.cat(outs + [torch.zeros_like(ctx)], dim=1)
    w = monotonic_attention(cell.attention_module, x_dec, w, memory, tr)
    return x_dec, ctx, (w, new_hcs)


def cell_step(cell, x_frame, state, memory, keep=None):
    if hasattr(cell, "decoder_rnn_list"):
        return taco2_cell_step(cell, x_frame, state, memory, keep)
    return prod_cell_step(cell, x_frame, state, memory, keep)


def decoder_forward(dec, memory: Tensor, mmask, x: Optional[Tensor], max_steps: int, p_no_forcing: Optional[float]):
    """Decoder.forward (decoder.py:16-77) with autograd."""
    _require_device(memory, "memory")
    cell, r, dm = dec.decoder_cell, dec.r, dec.dim_mel
    B, L, _ = memory.shape
    state = cell.initial_state(B, L, memory.dtype, memory.device)
    y_t = torch.zeros(B, r, dm, dtype=memory.dtype, device=memory.device)
    x_split = None
    if x is not None:
        T = (x.shape[1] // r) * r
        x_split = x[:, :T, :].split(r, dim=1)
        if not x_split:
            raise ValueError("teacher input shorter than one decoder step")
    stream = None
    if dec.dropout_source == "reference_rng":
        stream = MaskStream(B, cell.dim_pre, cell.pre_net.p_dropout, p_no_forcing=p_no_forcing if x is not None else None,
                            teacher_steps=len(x_split) if x_split is not None else None, d_pre_hidden=getattr(cell, "dim_pre_hidden", None))
    ys: List[Tensor] = []
    ss: List[Tensor] = []
    ws: List[Tensor] = []
    step = 0
    # reference_rng: the keep-masks (and teacher-forcing coin flips) of a teacher-forced run are all drawn up front, in the
    # reference's order, and uploaded in ONE copy (a blocking host-to-device copy per step otherwise); free-running decoding
    # draws step by step (its length is not known)
    pre_masks = pre_flags = None
    if stream is not None and x_split is not None:
        pre_masks, pre_flags = stream.draw(len(x_split))
        pre_masks = pre_masks.reshape(len(x_split), -1).to(memory.device)
    while True:
        keep, forced = None, True
        if stream is not None:
            w0, w1 = stream.widths
            if pre_masks is not None:
                flat, forced = pre_masks[step], bool(pre_flags[step])
            else:
                masks, flags = stream.draw(1)
                flat, forced = masks.reshape(-1).to(memory.device), bool(flags[0])
            keep = (flat[: B * w0].view(B, w0), flat[B * w0 :].view(B, w1))
        d_t, _, state = cell_step(cell, y_t[:, -1, :], state, memory, keep)
        s_t = dec.fc_stop(d_t).unsqueeze(2)
        y_t = F.leaky_relu(dec.fc_mel(d_t), 0.01).view(-1, r, dm)
        ys.append(y_t)
        ss.append(s_t)
        ws.append(state[0])
        step += 1
        if x_split is not None:
            if step >= len(x_split):
                break
            if stream is None and p_no_forcing:
                # (decoder.py:65: the coin is flipped here, behind the break - one draw per step but the last, so the host
                # generator ends where the reference leaves it)
                forced = bool(torch.rand(1) > p_no_forcing)
            if forced:
                y_t = x_split[step - 1]
        elif bool(torch.any(s_t < dec.stop_threshold)) or (max_steps and step > max_steps):
            break
    return torch.cat(ys, dim=1), torch.cat(ss, dim=1), torch.stack(ws, dim=1)


def mel_postnet(post, x: Tensor) -> Tensor:
    """MelPostnet.forward, modules.py:178-184, on the module's own Conv1d / BatchNorm1d (so BatchNorm follows
    post.training: batch statistics + running-stat updates in training)."""
    _require_device(x, "x")
    h = x.transpose(1, 2)
    for block in post.conv:
        h = F.dropout(_isru(block(h)), p=0.1, training=post.training)
    return x + post.fc_out(h.transpose(1, 2))


def conv1d_fix(mod, x: Tensor) -> Tensor:
    """Conv1dFix (mps_fixes/mps_fixes.py:22-29) as one convolution.  The reference stacks k rolled copies of the padded
    input and multiplies by weight.view(out, -1): flat column n * C_in + c meets x[c, t + pad - n].  With pad = (k-1)/2
    that is conv1d with the kernel W'[o, c, tap] = flat[o, (k - 1 - tap) * C_in + c] - the same re-indexing the weight
    packer of the HIP path applies (csrc conv1dfix_pack_kernel)."""
    co, ci, k = mod.weight.shape
    w = mod.weight.reshape(co, k, ci).flip(1).permute(0, 2, 1)
    return F.conv1d(x, w, mod.bias, padding=mod.padding)


def mel_postnet2(post, x: Tensor) -> Tensor:
    """MelPostnet2.forward, modules.py:187-216: x + block(x) per layer; a block is (transpose) Conv1dFix - BN - LeakyReLU -
    Dropout(0.2), twice, then Conv1dFix (transpose).  Layer indices follow the reference's nn.Sequential (state-dict keys)."""
    _require_device(x, "x")
    for blk in post.layers:
        h = x.transpose(1, 2)
        h = blk[4](blk[3](blk[2](conv1d_fix(blk[1], h))))
        h = blk[8](blk[7](blk[6](conv1d_fix(blk[5], h))))
        x = x + conv1d_fix(blk[9], h).transpose(1, 2)
    return x
