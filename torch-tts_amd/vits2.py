"""Drop-ins for the VITS2 second hot path of the reference:

* ``TextEncoder``                         - vits2/models.py:330-380 (same constructor, parameters, state-dict keys;
                                             ``forward(x, x_lengths, g=None) -> (x, m, logs, x_mask)``)
* ``ResidualCouplingTransformersBlock``   - vits2/models.py:681-810 with ``transformer_flow_type="pre_conv"``
                                             (``forward(x, x_mask, g=None, reverse=True)``)

Both hold the reference's parameters (so checkpoints load) and run inference through the HIP library
(``ttsvits_*`` in include/ttsdec.h).  The library works on channel-last activations; the [B, C, T]
tensors of the reference API are transposed here.  Training / the forward (non-reverse) direction of
the flow / speaker conditioning are outside the path and raise."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib
from .engine import PackedWeightsMixin, _require_device, _stream, weights_fingerprint


class _LayerNorm(nn.Module):  # modules.LayerNorm: parameters gamma / beta
    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))


class _MultiHeadAttention(nn.Module):  # attentions.MultiHeadAttention parameter holder (attentions.py:181-232)
    def __init__(self, channels, out_channels, n_heads, p_dropout=0.0, window_size=None):
        super().__init__()
        assert channels % n_heads == 0
        self.n_heads, self.window_size, self.k_channels = n_heads, window_size, channels // n_heads
        self.conv_q = nn.Conv1d(channels, channels, 1)
        self.conv_k = nn.Conv1d(channels, channels, 1)
        self.conv_v = nn.Conv1d(channels, channels, 1)
        self.conv_o = nn.Conv1d(channels, out_channels, 1)
        if window_size is not None:
            std = self.k_channels**-0.5
            self.emb_rel_k = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * std)
            self.emb_rel_v = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * std)
        nn.init.xavier_uniform_(self.conv_q.weight)
        nn.init.xavier_uniform_(self.conv_k.weight)
        nn.init.xavier_uniform_(self.conv_v.weight)


class _FFN(nn.Module):  # attentions.FFN parameter holder (attentions.py:385-410)
    def __init__(self, in_channels, out_channels, filter_channels, kernel_size):
        super().__init__()
        self.conv_1 = nn.Conv1d(in_channels, filter_channels, kernel_size)
        self.conv_2 = nn.Conv1d(filter_channels, out_channels, kernel_size)


class Encoder(nn.Module):
    """attentions.Encoder parameter layout (attentions.py:14-75)."""

    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0.0, window_size=4, **kwargs):
        super().__init__()
        self.hidden_channels, self.filter_channels, self.n_heads = hidden_channels, filter_channels, n_heads
        self.n_layers, self.kernel_size, self.window_size = n_layers, kernel_size, window_size
        # attentions.py:41-52: a speaker-conditioned encoder adds spk_emb_linear(g) to the input of layer cond_layer_idx (2 unless given)
        self.gin_channels = int(kwargs.get("gin_channels", 0) or 0)
        self.cond_layer_idx = n_layers
        if self.gin_channels:
            self.spk_emb_linear = nn.Linear(self.gin_channels, hidden_channels)
            self.cond_layer_idx = kwargs.get("cond_layer_idx", 2)
            assert self.cond_layer_idx < n_layers, "cond_layer_idx should be less than n_layers"
        self.attn_layers = nn.ModuleList(_MultiHeadAttention(hidden_channels, hidden_channels, n_heads, window_size=window_size) for _ in range(n_layers))
        self.norm_layers_1 = nn.ModuleList(_LayerNorm(hidden_channels) for _ in range(n_layers))
        self.ffn_layers = nn.ModuleList(_FFN(hidden_channels, hidden_channels, filter_channels, kernel_size) for _ in range(n_layers))
        self.norm_layers_2 = nn.ModuleList(_LayerNorm(hidden_channels) for _ in range(n_layers))

    def weight_tensors(self) -> List[torch.Tensor]:
        out = []
        for i in range(self.n_layers):
            a, f = self.attn_layers[i], self.ffn_layers[i]
            out += [a.conv_q.weight, a.conv_q.bias, a.conv_k.weight, a.conv_k.bias, a.conv_v.weight, a.conv_v.bias, a.conv_o.weight, a.conv_o.bias]
            if self.window_size is not None:
                out += [a.emb_rel_k, a.emb_rel_v]
            out += [self.norm_layers_1[i].gamma, self.norm_layers_1[i].beta, f.conv_1.weight, f.conv_1.bias, f.conv_2.weight, f.conv_2.bias,
                    self.norm_layers_2[i].gamma, self.norm_layers_2[i].beta]
        return out


class VitsEngine:
    """One ttsvits handle on one device."""

    def __init__(self, dims: Dict[str, int], device: torch.device):
        self._lib = _lib.load()
        self.device = device
        self.dims = dict(gin_channels=0, cond_layer_idx=0)
        self.dims.update(dims)
        dims = self.dims
        h = C.c_void_p()
        d = _lib.VitsDims(*[int(dims[n]) for n, _ in _lib.VitsDims._fields_])
        with torch.cuda.device(device):  # the handle binds to the device current at create
            _lib.check(self._lib.ttsvits_create(C.byref(d), C.byref(h)), "ttsvits_create")
        self._h = h
        self.blob: Optional[torch.Tensor] = None
        self._fingerprint = None
        self._ws: Dict = {}
        self.status = torch.zeros(1, dtype=torch.int32, device=device)  # bit 0: an id outside the table (ttsvits_text_encoder)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ttsvits_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc, what):
        if rc != _lib.OK:
            raise _lib.TtsdecError(rc, what, self._lib.ttsvits_last_hip_error(self._h).decode())

    def num_weight_tensors(self) -> int:
        return int(self._lib.ttsvits_num_weight_tensors(self._h))

    def set_precision(self, mode: str) -> None:
        """"f32" (exact, the modules' default) or "split_f16": arithmetic of every GEMM (include/ttsdec.h ttsvits_set_precision)."""
        self._err(self._lib.ttsvits_set_precision(self._h, {"f32": _lib.PREC_F32, "split_f16": _lib.PREC_SPLIT_F16}[mode]), "ttsvits_set_precision")

    def precision(self) -> str:
        return {_lib.PREC_F32: "f32", _lib.PREC_SPLIT_F16: "split_f16"}[int(self._lib.ttsvits_get_precision(self._h))]

    def ensure_packed(self, tensors, key_tensors=None) -> None:
        """tensors: the list the C ABI expects, or a callable producing it (called only when a repack is
        needed); key_tensors: the parameters whose identity / version decide that (default: the list itself -
        derived tensors such as weight-normed weights are new objects on every call and must not be the key)."""
        fp = weights_fingerprint([t for t in (key_tensors if key_tensors is not None else tensors) if t is not None])
        if self.blob is not None and fp == self._fingerprint:
            return
        if callable(tensors):
            tensors = tensors()
        n = len(tensors)
        assert n == self.num_weight_tensors(), (n, self.num_weight_tensors())
        arr = (C.c_void_p * n)()
        keep = []
        for i, t in enumerate(tensors):
            if t is None:
                arr[i] = None
                continue
            _require_device(t, "vits2 weights")
            tc = t.detach().to(torch.float32).contiguous()
            keep.append(tc)
            arr[i] = tc.data_ptr()
        with torch.cuda.device(self.device):
            blob = torch.empty(self._lib.ttsvits_packed_bytes(self._h), dtype=torch.uint8, device=self.device)
            rc = self._lib.ttsvits_pack_weights(self._h, arr, n, blob.data_ptr(), _stream(self.device))
            torch.cuda.current_stream(self.device).synchronize()  # `keep` must outlive the packing kernels
        self._err(rc, "ttsvits_pack_weights")
        self.blob, self._fingerprint = blob, fp

    def _workspace(self, kind: str, nbytes: int) -> torch.Tensor:
        ws = self._ws.get(kind)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[kind] = ws
        return ws

    def _speaker(self, g: Optional[torch.Tensor], B: int) -> Optional[torch.Tensor]:
        """g as the C ABI takes it: [B, gin] fp32 from the reference's [B, gin, 1] (models.py: g = emb_g(sid).unsqueeze(-1))."""
        if g is None:
            return None
        gin = self.dims["gin_channels"]
        if not gin:
            raise ValueError("g was given to a module built with gin_channels = 0")
        _require_device(g, "g")
        if g.dim() == 3:
            if g.shape[2] != 1:
                raise NotImplementedError("a time-varying g [B, gin, T] is outside the HIP path (the reference's callers pass [B, gin, 1])")
            g = g[:, :, 0]
        if tuple(g.shape) != (B, gin):
            raise ValueError(f"g must be [B, gin_channels(, 1)] = [{B}, {gin}(, 1)], got {tuple(g.shape)}")
        return g.to(torch.float32).contiguous()

    def text_encoder(self, ids: torch.Tensor, lengths: torch.Tensor, g: Optional[torch.Tensor] = None):
        _require_device(ids, "ids")
        B, T = ids.shape
        g = self._speaker(g, B)
        ids = ids.to(torch.int64).contiguous()
        self.status.zero_()
        lens = lengths.to(device=self.device, dtype=torch.int32).contiguous()
        H, I = self.dims["hidden_channels"], self.dims["inter_channels"]
        x = torch.empty(B, T, H, device=self.device)
        m = torch.empty(B, T, I, device=self.device)
        logs = torch.empty(B, T, I, device=self.device)
        ws = self._workspace("te", self._lib.ttsvits_text_encoder_workspace_bytes(self._h, B, T))
        with torch.cuda.device(self.device):
            rc = self._lib.ttsvits_text_encoder(self._h, ids.data_ptr(), lens.data_ptr(), g.data_ptr() if g is not None else None, B, T,
                                                x.data_ptr(), m.data_ptr(), logs.data_ptr(),
                                                ws.data_ptr(), ws.numel(), _stream(self.device), self.status.data_ptr())
        self._err(rc, "ttsvits_text_encoder")
        # nn.Embedding raises IndexError on an id outside the table (models.py:370): the kernel clamps and reports through the status
        # word - one host sync here, behind the queued launches (no scan of the ids before them)
        if int(self.status) & 1:
            raise IndexError(f"token id out of range [0, {self.dims['n_vocab']}) in the text encoder's input")
        return x, m, logs

    def flow_reverse(self, z_cl: torch.Tensor, lengths: torch.Tensor, g: Optional[torch.Tensor] = None) -> torch.Tensor:
        """z_cl [B, T, inter] channel-last."""
        _require_device(z_cl, "z")
        B, T, _ = z_cl.shape
        g = self._speaker(g, B)
        z_cl = z_cl.to(torch.float32).contiguous()
        lens = lengths.to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty_like(z_cl)
        ws = self._workspace("flow", self._lib.ttsvits_flow_workspace_bytes(self._h, B, T))
        with torch.cuda.device(self.device):
            rc = self._lib.ttsvits_flow_reverse(self._h, z_cl.data_ptr(), lens.data_ptr(), g.data_ptr() if g is not None else None, B, T,
                                                out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                _stream(self.device))
        self._err(rc, "ttsvits_flow_reverse")
        return out


class _EngCache:
    def __init__(self):
        self.by_dev: Dict[int, VitsEngine] = {}

    def __getstate__(self):
        return {}

    def __setstate__(self, st):
        self.by_dev = {}

    def __deepcopy__(self, memo):
        return _EngCache()

    def get(self, dims, device) -> VitsEngine:
        key = device.index if device.index is not None else torch.cuda.current_device()
        eng = self.by_dev.get(key)
        if eng is None:
            eng = VitsEngine(dims, torch.device("cuda", key))
            self.by_dev[key] = eng
        return eng


_DEFAULT_FLOW = dict(flow_hidden=4, flow_kernel=1, flow_wn_layers=1, n_flows=0, flow_tf_layers=0, flow_tf_heads=1, flow_tf_kernel=1)


class TextEncoder(PackedWeightsMixin, nn.Module):
    def __init__(self, n_vocab, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, gin_channels=0):
        super().__init__()
        self._watch_state_dict_loads()
        self.n_vocab, self.out_channels, self.hidden_channels, self.filter_channels = n_vocab, out_channels, hidden_channels, filter_channels
        self.n_heads, self.n_layers, self.kernel_size, self.p_dropout, self.gin_channels = n_heads, n_layers, kernel_size, p_dropout, gin_channels
        self.emb = nn.Embedding(n_vocab, hidden_channels)
        nn.init.normal_(self.emb.weight, 0.0, hidden_channels**-0.5)
        self.encoder = Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, gin_channels=gin_channels)  # models.py:358-366
        self.proj = nn.Conv1d(hidden_channels, out_channels * 2, 1)
        self.precision = "f32"  # arithmetic of the GEMMs: "f32" (the reference's own: exact fp32, default) or "split_f16" (two fp16 planes, opt-in: ~1.8x faster)
        self._engines = _EngCache()

    def _dims(self):
        d = dict(n_vocab=self.n_vocab, inter_channels=self.out_channels, hidden_channels=self.hidden_channels, filter_channels=self.filter_channels,
                 n_heads=self.n_heads, n_layers=self.n_layers, kernel_size=self.kernel_size, window_size=self.encoder.window_size,
                 gin_channels=self.gin_channels, cond_layer_idx=self.encoder.cond_layer_idx if self.gin_channels else 0)
        d.update(_DEFAULT_FLOW)
        return d

    def forward(self, x, x_lengths, g=None):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the HIP text encoder is inference-only: call under torch.no_grad()")
        eng = self._engines.get(self._dims(), x.device)
        eng.set_precision(self.precision)
        spk = [self.encoder.spk_emb_linear.weight, self.encoder.spk_emb_linear.bias] if self.gin_channels else []
        eng.ensure_packed([self.emb.weight] + spk + self.encoder.weight_tensors() + [self.proj.weight, self.proj.bias])
        xo, m, logs = eng.text_encoder(x, x_lengths, g)
        T = x.shape[1]
        x_mask = (torch.arange(T, device=x.device)[None, :] < x_lengths.to(x.device)[:, None]).unsqueeze(1).to(xo.dtype)
        return xo.transpose(1, 2), m.transpose(1, 2), logs.transpose(1, 2), x_mask


class _WN(nn.Module):
    """modules.WN parameter layout (modules.py:133-183): weight-normalised convs (weight_g / weight_v)."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        if dilation_rate != 1:
            raise NotImplementedError("WN with dilation_rate != 1 is outside the HIP path")
        self.hidden_channels, self.n_layers, self.kernel, self.gin_channels = hidden_channels, n_layers, kernel_size, gin_channels
        self.in_layers, self.res_skip_layers = nn.ModuleList(), nn.ModuleList()
        if gin_channels != 0:  # modules.py:149-153
            self.cond_layer = nn.utils.weight_norm(nn.Conv1d(gin_channels, 2 * hidden_channels * n_layers, 1), name="weight")
        for i in range(n_layers):
            self.in_layers.append(nn.utils.weight_norm(nn.Conv1d(hidden_channels, 2 * hidden_channels, kernel_size, padding=(kernel_size - 1) // 2), name="weight"))
            rs = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(nn.utils.weight_norm(nn.Conv1d(hidden_channels, rs, 1), name="weight"))

    def weight_tensors(self):
        out = []
        if self.gin_channels != 0:
            out += [torch._weight_norm(self.cond_layer.weight_v, self.cond_layer.weight_g, 0), self.cond_layer.bias]
        for i in range(self.n_layers):
            for l in (self.in_layers[i], self.res_skip_layers[i]):
                w = torch._weight_norm(l.weight_v, l.weight_g, 0)  # effective weight g * v / ||v||
                out += [w, l.bias]
        return out


class ResidualCouplingTransformersLayer(nn.Module):
    """Parameter layout of models.py:436-505 (mean_only)."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0, gin_channels=0, mean_only=False):
        super().__init__()
        if not mean_only:
            raise NotImplementedError("only mean_only=True (what ResidualCouplingTransformersBlock builds) is on the HIP path")
        self.channels, self.hidden_channels, self.half_channels = channels, hidden_channels, channels // 2
        self.pre_transformer = Encoder(self.half_channels, self.half_channels, n_heads=2, n_layers=2, kernel_size=3, p_dropout=0.1, window_size=None)
        self.pre = nn.Conv1d(self.half_channels, hidden_channels, 1)
        self.enc = _WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels, p_dropout=p_dropout)
        # (present in the reference's state dict, unused by its forward: models.py:513-515)
        self.post_transformer = Encoder(hidden_channels, hidden_channels, n_heads=2, n_layers=2, kernel_size=3, p_dropout=0.1, window_size=None)
        self.post = nn.Conv1d(hidden_channels, self.half_channels, 1)
        self.post.weight.data.zero_()
        self.post.bias.data.zero_()

    def weight_tensors(self):
        return self.pre_transformer.weight_tensors() + [self.pre.weight, self.pre.bias] + self.enc.weight_tensors() + [self.post.weight, self.post.bias]


class _Flip(nn.Module):
    pass


class ResidualCouplingTransformersBlock(PackedWeightsMixin, nn.Module):
    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4, gin_channels=0,
                 use_transformer_flows=False, transformer_flow_type="pre_conv"):
        super().__init__()
        self._watch_state_dict_loads()
        if not use_transformer_flows or transformer_flow_type != "pre_conv":
            raise NotImplementedError("only use_transformer_flows=True with transformer_flow_type='pre_conv' (the ModelConfig default) is built")
        self.channels, self.hidden_channels, self.kernel_size, self.n_layers, self.n_flows = channels, hidden_channels, kernel_size, n_layers, n_flows
        self.gin_channels = gin_channels
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(ResidualCouplingTransformersLayer(channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                                                                gin_channels=gin_channels, mean_only=True))
            self.flows.append(_Flip())
        self.precision = "f32"  # arithmetic of the GEMMs: "f32" (the reference's own: exact fp32, default) or "split_f16" (two fp16 planes, opt-in: ~1.8x faster)
        self._engines = _EngCache()

    def _dims(self):
        return dict(n_vocab=1, inter_channels=self.channels, hidden_channels=4, filter_channels=4, n_heads=1, n_layers=0, kernel_size=1, window_size=0,
                    flow_hidden=self.hidden_channels, flow_kernel=self.kernel_size, flow_wn_layers=self.n_layers, n_flows=self.n_flows,
                    flow_tf_layers=2, flow_tf_heads=2, flow_tf_kernel=3, gin_channels=self.gin_channels, cond_layer_idx=0)

    def forward(self, x, x_mask, g=None, reverse=False):
        if not reverse:
            raise NotImplementedError("only the reverse (inference) direction is on the HIP path")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the HIP flow is inference-only: call under torch.no_grad()")
        eng = self._engines.get(self._dims(), x.device)
        eng.set_precision(self.precision)

        def tensors() -> List[Optional[torch.Tensor]]:
            # emb, (spk_emb_linear.{weight,bias},) proj.weight, proj.bias of the (absent) text encoder
            ts: List[Optional[torch.Tensor]] = [None] * (5 if self.gin_channels else 3)
            for i in range(self.n_flows):
                ts += self.flows[2 * i].weight_tensors()  # (materialises the weight-normed conv weights)
            return ts

        eng.ensure_packed(tensors, key_tensors=list(self.parameters()))
        lengths = x_mask[:, 0, :].sum(dim=1).round().to(torch.int32)  # sequence_mask is a prefix mask
        out = eng.flow_reverse(x.transpose(1, 2), lengths, g)
        return out.transpose(1, 2)
