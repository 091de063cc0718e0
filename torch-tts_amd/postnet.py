"""Drop-in for the reference's ``modules.modules.MelPostnet``
(tacotron/modules/modules.py:155-184): same constructor, state-dict keys
(conv.{i}.0.weight, conv.{i}.1.{weight,bias,running_mean,running_var}, fc_out.weight)
and ``forward(x[B,T,D_mel]) -> [B,T,D_mel]``; eval-mode forward runs as implicit-GEMM
conv kernels through ``ttsdec_postnet``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .engine import EngineCache, EngineDims


class MelPostnet(nn.Module):
    def __init__(self, dim_mel, dim_hidden=512, kernel_size=5, num_layers=3):
        super().__init__()
        padding = (kernel_size - 1) // 2
        conv_dims = [dim_mel] + [dim_hidden for _ in range(num_layers)]
        self.conv = nn.ModuleList(
            [
                nn.Sequential(
                    nn.Conv1d(ch_in, ch_out, kernel_size=kernel_size, padding=padding, bias=False),
                    nn.BatchNorm1d(ch_out),
                )
                for ch_in, ch_out in zip(conv_dims[:-1], conv_dims[1:])
            ]
        )
        self.fc_out = nn.Linear(dim_hidden, dim_mel, bias=False)
        self.dim_mel, self.dim_hidden, self.kernel_size, self.num_layers = dim_mel, dim_hidden, kernel_size, num_layers
        # "f32": exact fp32 matrix instruction; "split_f16": hi/lo fp16 planes (fp32-grade, faster);
        # "bf16": bf16 operands with fp32 accumulate (~3 significant digits)
        self.precision = "f32"
        self._engines = EngineCache()

    def weight_tensors(self):
        ts = [None] * _lib.W_DECODER_COUNT
        for layer in self.conv:
            conv, bn = layer[0], layer[1]
            ts += [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        ts.append(self.fc_out.weight)
        return ts

    def engine_dims(self) -> EngineDims:
        return EngineDims(
            d_mel=self.dim_mel, r=1, d_pre=4, d_ctx=4, h_att=4, h_dec=4, p_zoneout=0.0, p_dropout=0.0,
            postnet_layers=self.num_layers, postnet_hidden=self.dim_hidden, postnet_kernel=self.kernel_size,
            bn_eps=float(self.conv[0][1].eps),
        )

    def engine(self, device):
        eng = self._engines.get(self.engine_dims(), device)
        eng.ensure_packed(self.weight_tensors())
        return eng

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("MelPostnet runs on the HIP path only: move the module and input to a ROCm device")
        if self.training:
            raise NotImplementedError("MelPostnet on the HIP path is eval-mode only (BatchNorm running stats, no dropout)")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("autograd through the postnet is outside the HIP hot path: call under torch.no_grad()")
        prec = {"f32": _lib.POSTNET_F32, "bf16": _lib.POSTNET_BF16, "split_f16": _lib.POSTNET_SPLIT_F16}[self.precision]
        return self.engine(x.device).postnet(x.detach().to(torch.float32).contiguous(), prec)
