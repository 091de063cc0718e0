"""Drop-in for the reference's ``decoder_cell.Taco2ProdDecoderCell``
(tacotron/decoder_cell.py:143-195): same constructor, attributes, parameter tree
(state-dict keys) and ``initial_state`` / ``forward`` signatures; one step runs as
HIP kernels through libttsdec (``ttsdec_cell_step``); training mode and grad-enabled calls run as torch ops on the
device (autograd_path.py)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from .engine import EngineCache, EngineDims, PackedWeightsMixin


def _needs_autograd(module, *tensors) -> bool:
    """training-mode semantics, or a graph to differentiate -> the torch-op device path (autograd_path.py)"""
    if module.training:
        return True
    return torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters()))


class PreNet(nn.Module):
    """tacotron/modules/modules.py:15-41 (keys layers.{0,1}.{weight,bias}).  Inside Decoder / the cells it runs fused in
    the frame kernel; called on its own it runs as torch ops on the device (always-on dropout from the device generator)."""

    def __init__(self, dim_input, dim_output, dim_hidden=256, p_dropout=0.5, always_dropout=False):
        super().__init__()
        self.p_dropout = p_dropout
        self.always_dropout = always_dropout
        self.layers = nn.ModuleList([nn.Linear(dim_input, dim_hidden), nn.Linear(dim_hidden, dim_output)])

    def forward(self, x):
        from . import autograd_path

        autograd_path._require_device(x, "x")
        return autograd_path.prenet(self, x)


class LSTMZoneoutCell(nn.LSTMCell):
    """tacotron/modules/rnn.py:19-39 (nn.LSTMCell keys and default init).  Fused into the LSTM kernels inside
    Decoder / the cells; called on its own it runs as torch ops on the device (eval blend / training zoneout masks)."""

    def __init__(self, input_size, hidden_size, bias=True, p_zoneout=None):
        super().__init__(input_size, hidden_size, bias=bias)
        self.p_zoneout = p_zoneout

    def forward(self, x, hidden):
        from . import autograd_path

        autograd_path._require_device(x, "x")
        return autograd_path.lstm_zoneout(self, x, hidden, self.training)


class StepwiseMonotonicAttention(nn.Module):
    """tacotron/modules/attention.py:96-126 (keys query_layer.weight, bias; ``bias`` is unused by the reference's forward,
    kept for checkpoint compatibility).  Fused into the attention kernel inside Decoder / the cells; called on its own
    it runs as torch ops on the device."""

    def __init__(self, dim_input, dim_context, sigmoid_noise=1.0):
        super().__init__()
        self.sigmoid_noise = sigmoid_noise
        self.query_layer = nn.Linear(dim_input, dim_context, bias=False)
        self.bias = nn.Parameter(torch.Tensor([1.0]))

    def forward(self, x, w, memory, cmask=None):
        from . import autograd_path

        autograd_path._require_device(memory, "memory")
        return autograd_path.monotonic_attention(self, x, w, memory, self.training)


class Taco2ProdDecoderCell(PackedWeightsMixin, nn.Module):
    def __init__(self, dim_ctx, dim_mel, r, dim_rnn, dim_pre=128, dim_att=128, p_zoneout=0.1):
        super().__init__()
        self._watch_state_dict_loads()
        dim_att_hidden, dim_dec_hidden = dim_rnn[0], dim_rnn[1]
        self.dim_output = dim_dec_hidden + dim_ctx
        self.dim_ctx, self.dim_mel, self.r, self.dim_pre = dim_ctx, dim_mel, r, dim_pre
        self.dim_pre_hidden = dim_pre
        self.p_zoneout = p_zoneout

        self.pre_net = PreNet(dim_mel, dim_pre, always_dropout=True, dim_hidden=dim_pre)
        self.attention_module = StepwiseMonotonicAttention(dim_att_hidden, dim_ctx)
        self.attention_rnn = LSTMZoneoutCell(dim_pre + dim_ctx, dim_att_hidden, p_zoneout=p_zoneout)
        self.decoder_rnn = LSTMZoneoutCell(dim_att_hidden + dim_ctx, dim_dec_hidden, p_zoneout=p_zoneout)
        self.initial_decoder_h = nn.ParameterList(
            [nn.Parameter(torch.zeros(1, dim_att_hidden)), nn.Parameter(torch.zeros(1, dim_dec_hidden))]
        )
        self.initial_decoder_c = nn.ParameterList(
            [nn.Parameter(torch.zeros(1, dim_att_hidden)), nn.Parameter(torch.zeros(1, dim_dec_hidden))]
        )
        self.initial_ctx_0 = torch.zeros(1, dim_ctx)

        # how the always-on PreNet dropout is sourced when the cell is driven directly:
        #   "reference_rng": replay of the reference's CPU generator draws (default)
        #   "philox":        on-device counter RNG keyed by (dropout_seed, step)
        #   "off"
        self.dropout_source = "reference_rng"
        self.dropout_seed = 0
        self._step_counter = 0
        self._engines = EngineCache()

    # ---- parameters in TTSDEC_W_* order (None for what the cell does not own) ----
    def weight_tensors(self) -> List[Optional[torch.Tensor]]:
        pn, at, ar, dr = self.pre_net, self.attention_module, self.attention_rnn, self.decoder_rnn
        return [
            pn.layers[0].weight, pn.layers[0].bias, pn.layers[1].weight, pn.layers[1].bias,
            at.query_layer.weight,
            ar.weight_ih, ar.weight_hh, ar.bias_ih, ar.bias_hh,
            dr.weight_ih, dr.weight_hh, dr.bias_ih, dr.bias_hh,
            self.initial_decoder_h[0], self.initial_decoder_h[1], self.initial_decoder_c[0], self.initial_decoder_c[1],
        ]

    def engine_dims(self) -> EngineDims:
        return EngineDims(
            d_mel=self.dim_mel, r=self.r, d_pre=self.dim_pre, d_ctx=self.dim_ctx,
            h_att=self.attention_rnn.hidden_size, h_dec=self.decoder_rnn.hidden_size,
            p_zoneout=float(self.p_zoneout or 0.0), p_dropout=float(self.pre_net.p_dropout),
        )

    def initial_state(self, batch_size, memory_size, dtype, device):
        """(w_0 one-hot at position 0, ctx_0 zeros, [(h,c) attention rnn, (h,c) decoder rnn])
        - decoder_cell.py:9-17,165-178."""
        w_0 = torch.zeros(batch_size, memory_size, dtype=dtype, device=device)
        w_0[:, 0] = 1.0
        ctx_0 = self.initial_ctx_0.to(dtype=dtype, device=device).expand(batch_size, -1)
        h_rnn_0 = [
            (h.to(dtype=dtype, device=device).expand(batch_size, -1), c.to(dtype=dtype, device=device).expand(batch_size, -1))
            for h, c in zip(self.initial_decoder_h, self.initial_decoder_c)
        ]
        return w_0, ctx_0, h_rnn_0

    def forward(self, x, dec_state, memory, mmask):
        """x [B, r, D_mel]; dec_state = (w, ctx, ((h_att,c_att),(h_dec,c_dec))); memory [B, L, D_ctx].
        Returns (x_dec, ctx_att, dec_state) like decoder_cell.py:180-195.  Inference only."""
        if not memory.is_cuda:
            raise RuntimeError("Taco2ProdDecoderCell runs on the HIP path only: move the module and inputs to a ROCm device")
        if _needs_autograd(self, x, memory):
            from . import autograd_path

            xin = x.flatten(1, 2)[:, -self.dim_mel :] if x.dim() == 3 else x
            autograd_path.warn_eval_on_autograd_path(self)
            return autograd_path.prod_cell_step(self, xin, (dec_state[0], dec_state[1], dec_state[2]), memory)
        w_att, ctx_att, (hc_att, hc_dec) = dec_state[0], dec_state[1], dec_state[2]
        B = memory.shape[0]
        eng = self._engines.get(self.engine_dims(), memory.device)
        tensors = self.weight_tensors() + [None, None, None, None]
        eng.ensure_packed(tensors)
        f = lambda t: t.to(torch.float32).contiguous().clone()
        w, ctx = f(w_att), f(ctx_att)
        h_att, c_att, h_dec, c_dec = f(hc_att[0]), f(hc_att[1]), f(hc_dec[0]), f(hc_dec[1])
        xin = x.flatten(1, 2)[:, -self.dim_mel :].to(torch.float32).contiguous() if x.dim() == 3 else x.contiguous()
        masks = None
        if self.dropout_source == "reference_rng":
            m = torch.empty(2, B, self.dim_pre)
            m[0].bernoulli_(1.0 - self.pre_net.p_dropout)
            m[1].bernoulli_(1.0 - self.pre_net.p_dropout)
            masks = m.to(torch.uint8).to(memory.device)
            mode = _lib.DROPOUT_MASKS
        elif self.dropout_source == "philox":
            mode = _lib.DROPOUT_PHILOX
        else:
            mode = _lib.DROPOUT_OFF
        x_dec = eng.cell_step(
            xin, memory.to(torch.float32).contiguous(), w, ctx, h_att, c_att, h_dec, c_dec, mode, masks,
            self.dropout_seed, self._step_counter,
        )
        self._step_counter += 1
        return x_dec, ctx, (w, ctx, ((h_att, c_att), (h_dec, c_dec)))


class Taco2DecoderCell(PackedWeightsMixin, nn.Module):
    """Drop-in for the reference's ``decoder_cell.Taco2DecoderCell`` (tacotron/decoder_cell.py:66-140),
    the cell of config-rdh / config-sandra / config_template: context from the PREVIOUS attention
    weights feeds two stacked zoneout LSTMs, the attention input and the cell output are
    cat[h0, h1, zeros_like(ctx)].  State: (w, [(h0, c0), (h1, c1)])."""

    def __init__(self, dim_ctx, dim_mel, r, dim_rnn, dim_pre=128, dim_att=128, p_zoneout=0.1):
        super().__init__()
        self._watch_state_dict_loads()
        if len(dim_rnn) != 2:
            raise ValueError("Taco2DecoderCell.forward indexes exactly two LSTM layers (decoder_cell.py:126)")
        self.dim_output = sum(dim_rnn) + dim_ctx
        self.dim_ctx, self.dim_mel, self.r, self.dim_pre = dim_ctx, dim_mel, r, dim_pre
        self.dim_pre_hidden = 128  # decoder_cell.py:74-76
        self.p_zoneout = p_zoneout
        self.pre_net = PreNet(dim_mel, dim_pre, always_dropout=True, p_dropout=0.5, dim_hidden=128)
        self.attention_module = StepwiseMonotonicAttention(sum(dim_rnn) + dim_ctx, dim_ctx)
        rnn_dims = [dim_pre] + list(dim_rnn)
        self.decoder_rnn_list = nn.ModuleList(
            [LSTMZoneoutCell(d_in + dim_ctx, d_h, p_zoneout=p_zoneout) for d_in, d_h in zip(rnn_dims[:-1], rnn_dims[1:])]
        )
        self.initial_decoder_h = nn.ParameterList([nn.Parameter(torch.zeros(1, d)) for d in dim_rnn])
        self.initial_decoder_c = nn.ParameterList([nn.Parameter(torch.zeros(1, d)) for d in dim_rnn])
        self.dropout_source = "reference_rng"
        self.dropout_seed = 0
        self._step_counter = 0
        self._engines = EngineCache()

    def weight_tensors(self) -> List[Optional[torch.Tensor]]:
        pn, at = self.pre_net, self.attention_module
        r0, r1 = self.decoder_rnn_list[0], self.decoder_rnn_list[1]
        return [
            pn.layers[0].weight, pn.layers[0].bias, pn.layers[1].weight, pn.layers[1].bias,
            at.query_layer.weight,
            r0.weight_ih, r0.weight_hh, r0.bias_ih, r0.bias_hh,
            r1.weight_ih, r1.weight_hh, r1.bias_ih, r1.bias_hh,
            self.initial_decoder_h[0], self.initial_decoder_h[1], self.initial_decoder_c[0], self.initial_decoder_c[1],
        ]

    def engine_dims(self) -> EngineDims:
        return EngineDims(
            d_mel=self.dim_mel, r=self.r, d_pre=self.dim_pre, d_ctx=self.dim_ctx,
            h_att=self.decoder_rnn_list[0].hidden_size, h_dec=self.decoder_rnn_list[1].hidden_size,
            p_zoneout=float(self.p_zoneout or 0.0), p_dropout=float(self.pre_net.p_dropout),
            cell_type=_lib.CELL_TACO2, d_pre_hidden=self.dim_pre_hidden,
        )

    def initial_state(self, batch_size, memory_size, dtype, device):
        """(w_0 one-hot at position 0, [(h, c) per LSTM]) - decoder_cell.py:96-108."""
        w_0 = torch.zeros(batch_size, memory_size, dtype=dtype, device=device)
        w_0[:, 0] = 1.0
        h_dec_0 = [
            (h.to(dtype=dtype, device=device).expand(batch_size, -1), c.to(dtype=dtype, device=device).expand(batch_size, -1))
            for h, c in zip(self.initial_decoder_h, self.initial_decoder_c)
        ]
        return w_0, h_dec_0

    def forward(self, x, dec_state, memory, mmask):
        """Returns (x_dec, ctx_att, (w, h_dec)) like decoder_cell.py:110-140.  Inference only."""
        if not memory.is_cuda:
            raise RuntimeError("Taco2DecoderCell runs on the HIP path only: move the module and inputs to a ROCm device")
        if _needs_autograd(self, x, memory):
            from . import autograd_path

            xin = x.flatten(1, 2)[:, -self.dim_mel :] if x.dim() == 3 else x
            autograd_path.warn_eval_on_autograd_path(self)
            return autograd_path.taco2_cell_step(self, xin, (dec_state[0], dec_state[1]), memory)
        w_in, h_dec = dec_state[0], dec_state[1]
        B = memory.shape[0]
        eng = self._engines.get(self.engine_dims(), memory.device)
        eng.ensure_packed(self.weight_tensors() + [None, None, None, None])
        f = lambda t: t.to(torch.float32).contiguous().clone()
        w = f(w_in)
        h0, c0, h1, c1 = f(h_dec[0][0]), f(h_dec[0][1]), f(h_dec[1][0]), f(h_dec[1][1])
        ctx = torch.empty(B, self.dim_ctx, dtype=torch.float32, device=memory.device)
        xin = x.flatten(1, 2)[:, -self.dim_mel :].to(torch.float32).contiguous() if x.dim() == 3 else x.contiguous()
        masks = None
        if self.dropout_source == "reference_rng":
            p = self.pre_net.p_dropout
            m0 = torch.empty(B, self.dim_pre_hidden).bernoulli_(1.0 - p).to(torch.uint8)
            m1 = torch.empty(B, self.dim_pre).bernoulli_(1.0 - p).to(torch.uint8)
            masks = torch.cat([m0.reshape(-1), m1.reshape(-1)]).to(memory.device)
            mode = _lib.DROPOUT_MASKS
        elif self.dropout_source == "philox":
            mode = _lib.DROPOUT_PHILOX
        else:
            mode = _lib.DROPOUT_OFF
        x_dec = eng.cell_step(
            xin, memory.to(torch.float32).contiguous(), w, ctx, h0, c0, h1, c1, mode, masks, self.dropout_seed, self._step_counter
        )
        self._step_counter += 1
        return x_dec, ctx, (w, [(h0, c0), (h1, c1)])
