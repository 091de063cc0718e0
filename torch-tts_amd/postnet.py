"""Drop-in for the reference's ``modules.modules.MelPostnet``
(tacotron/modules/modules.py:155-184): same constructor, state-dict keys
(conv.{i}.0.weight, conv.{i}.1.{weight,bias,running_mean,running_var}, fc_out.weight)
and ``forward(x[B,T,D_mel]) -> [B,T,D_mel]``; eval-mode forward runs as implicit-GEMM
conv kernels through ``ttsdec_postnet``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .engine import EngineCache, EngineDims, PackedWeightsMixin


class MelPostnet(PackedWeightsMixin, nn.Module):
    def __init__(self, dim_mel, dim_hidden=512, kernel_size=5, num_layers=3):
        super().__init__()
        self._watch_state_dict_loads()
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
        if self.training or (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            from . import autograd_path  # training semantics (batch-statistics BatchNorm, dropout) or a graph: torch ops on the device

            autograd_path.warn_eval_on_autograd_path(self)
            return autograd_path.mel_postnet(self, x)
        prec = {"f32": _lib.POSTNET_F32, "bf16": _lib.POSTNET_BF16, "split_f16": _lib.POSTNET_SPLIT_F16}[self.precision]
        return self.engine(x.device).postnet(x.detach().to(torch.float32).contiguous(), prec)


class Conv1dFix(nn.Module):
    """Parameter holder for tacotron/mps_fixes/mps_fixes.py:6-29 (key ``weight`` [out, in, k])."""

    def __init__(self, in_channels, out_channels, kernel_size, padding, bias=True):
        super().__init__()
        self.padding, self.kernel_size = padding, kernel_size
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.zeros([out_channels, in_channels, kernel_size]))
        nn.init.xavier_uniform_(self.weight)
        if bias:
            self.bias = nn.Parameter(torch.zeros([out_channels]))
            nn.init.normal_(self.bias)
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        from . import autograd_path  # (inside MelPostnet2's eval forward it runs fused on the HIP path instead)

        return autograd_path.conv1d_fix(self, x)


class MelPostnet2(PackedWeightsMixin, nn.Module):
    """Drop-in for the reference's ``modules.modules.MelPostnet2`` (tacotron/modules/modules.py:187-216),
    selected when model.postnet.type is not "tacotron2" (tacotron.py:207-212): ``num_layers`` residual
    blocks of Conv1dFix(k=5)-BN-LeakyReLU-Dropout x2 + Conv1dFix.  Same state-dict keys
    (layers.{i}.{1,5,9}.weight, layers.{i}.{2,6}.*); eval-mode forward through ``ttsdec_postnet``."""

    def __init__(self, dim_in, dim_hidden=128, num_layers=3):
        super().__init__()
        self._watch_state_dict_loads()
        self.layers = nn.ModuleList(
            [
                nn.Sequential(
                    nn.Identity(),  # Transposition (no parameters)
                    Conv1dFix(dim_in, dim_hidden, kernel_size=5, padding=2, bias=False),
                    nn.BatchNorm1d(dim_hidden),
                    nn.LeakyReLU(),
                    nn.Dropout(0.2),
                    Conv1dFix(dim_hidden, dim_hidden, kernel_size=5, padding=2, bias=False),
                    nn.BatchNorm1d(dim_hidden),
                    nn.LeakyReLU(),
                    nn.Dropout(0.2),
                    Conv1dFix(dim_hidden, dim_in, kernel_size=5, padding=2, bias=False),
                    nn.Identity(),  # Transposition
                )
                for _ in range(num_layers)
            ]
        )
        self.dim_in, self.dim_hidden, self.num_layers = dim_in, dim_hidden, num_layers
        self.precision = "f32"
        self._engines = EngineCache()

    def weight_tensors(self):
        ts = [None] * _lib.W_DECODER_COUNT
        for layer in self.layers:
            for conv_i, bn_i in ((1, 2), (5, 6)):
                bn = layer[bn_i]
                ts += [layer[conv_i].weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
            ts.append(layer[9].weight)
        return ts

    def engine_dims(self) -> EngineDims:
        return EngineDims(
            d_mel=self.dim_in, r=1, d_pre=4, d_ctx=4, h_att=4, h_dec=4, p_zoneout=0.0, p_dropout=0.0,
            postnet_layers=self.num_layers, postnet_hidden=self.dim_hidden, postnet_kernel=5,
            bn_eps=float(self.layers[0][2].eps), postnet_type=_lib.POSTNET_TYPE_MEL2,
        )

    def engine(self, device):
        eng = self._engines.get(self.engine_dims(), device)
        eng.ensure_packed(self.weight_tensors())
        return eng

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("MelPostnet2 runs on the HIP path only: move the module and input to a ROCm device")
        if self.training or (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            from . import autograd_path

            autograd_path.warn_eval_on_autograd_path(self)
            return autograd_path.mel_postnet2(self, x)
        prec = {"f32": _lib.POSTNET_F32, "bf16": _lib.POSTNET_BF16, "split_f16": _lib.POSTNET_SPLIT_F16}[self.precision]
        return self.engine(x.device).postnet(x.detach().to(torch.float32).contiguous(), prec)
