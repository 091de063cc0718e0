"""Drop-in for the reference's ``encoder.Encoder2`` (tacotron/encoder.py:27-82): same constructor,
state-dict keys and ``forward(x, x_lengths) -> memory [B, max(lengths), dim_out]``.  In eval mode on
a ROCm device (no autograd) the forward runs through ``ttsenc_forward`` (embedding gather, implicit-
GEMM convs, one input-projection GEMM, packed-sequence LSTM steps); training keeps the stock
PyTorch ops so gradients flow (the encoder is outside the inference hot path)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib
from .engine import PackedWeightsMixin, _require_device, _stream, weights_fingerprint


class _ISRLU(nn.Module):
    def forward(self, x):  # activations.py:13-14
        return torch.where(x >= 0, x, x / torch.sqrt(1 + x * x))


class _BiDiLSTM(nn.Module):
    """Packed bidirectional LSTM wrapper with the reference's key names (rnn.py:112-127)."""

    def __init__(self, input_size, hidden_size, bias=True):
        super().__init__()
        self.rnn = nn.LSTM(input_size, hidden_size, batch_first=True, bias=bias, bidirectional=True)

    def forward(self, x, x_lengths, h0, c0):
        x = nn.utils.rnn.pack_padded_sequence(x, x_lengths.cpu(), batch_first=True, enforce_sorted=False)
        h0 = torch.cat(torch.chunk(h0, 2, dim=-1), dim=0).contiguous()
        c0 = torch.cat(torch.chunk(c0, 2, dim=-1), dim=0).contiguous()
        x, (h, _) = self.rnn(x, (h0, c0))
        x, _ = nn.utils.rnn.pad_packed_sequence(x, batch_first=True)
        return x, h


class EncoderEngine:
    """One ttsenc handle on one device."""

    def __init__(self, alphabet_size: int, d_emb: int, d_out: int, bn_eps: float, device: torch.device):
        self._lib = _lib.load()
        self.device = device
        h = C.c_void_p()
        dims = _lib.EncDims(alphabet_size, d_emb, d_out, 5, bn_eps)
        with torch.cuda.device(device):  # the handle binds to the device current at create
            _lib.check(self._lib.ttsenc_create(C.byref(dims), C.byref(h)), "ttsenc_create")
        self.alphabet_size = alphabet_size
        self._h = h
        self.d_out = d_out
        self.blob: Optional[torch.Tensor] = None
        self._fingerprint = None
        self._ws: Dict = {}
        self.status = torch.zeros(1, dtype=torch.int32, device=device)  # bit 0: an id outside the table (ttsenc_forward)
        self._status_pending = False

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ttsenc_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def ensure_packed(self, tensors) -> None:
        fp = weights_fingerprint(tensors)
        if self.blob is not None and fp == self._fingerprint:
            return
        n = len(tensors)
        arr = (C.c_void_p * n)()
        keep = []
        for i, t in enumerate(tensors):
            _require_device(t, "encoder weights")
            tc = t.detach().to(torch.float32).contiguous()
            keep.append(tc)
            arr[i] = tc.data_ptr()
        with torch.cuda.device(self.device):
            blob = torch.empty(self._lib.ttsenc_packed_bytes(self._h), dtype=torch.uint8, device=self.device)
            rc = self._lib.ttsenc_pack_weights(self._h, arr, n, blob.data_ptr(), _stream(self.device))
        if rc != _lib.OK:
            raise _lib.TtsdecError(rc, "ttsenc_pack_weights", self._lib.ttsenc_last_hip_error(self._h).decode())
        self.blob, self._fingerprint = blob, fp

    def set_precision(self, mode: str) -> None:
        """"f32" (exact, default) or "split_f16": arithmetic of the conv / input-projection GEMMs (include/ttsdec.h ttsenc_set_precision)."""
        _lib.check(self._lib.ttsenc_set_precision(self._h, {"f32": _lib.PREC_F32, "split_f16": _lib.PREC_SPLIT_F16}[mode]), "ttsenc_set_precision")

    def check_ids(self) -> None:
        """Raises IndexError if the last forward met a token id outside the table (nn.Embedding does, encoder.py:69).  Reads the
        device status word: ONE host sync, placed by the caller - Encoder2.forward right behind its launches, Tacotron.forward
        at the decoder's own sync (none extra)."""
        if self._status_pending:
            self._status_pending = False
            if int(self.status) & 1:
                raise IndexError(f"token id out of range [0, {self.alphabet_size}) in the encoder's last input")

    def forward(self, ids: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        _require_device(ids, "ids")
        B, L = ids.shape
        l_out = int(lengths.max())  # (the reference's lengths live on the host: pack_padded_sequence wants them there, rnn.py:120)
        ids = ids.to(torch.int64).contiguous()
        self.status.zero_()
        self._status_pending = True
        lens = lengths.to(device=self.device, dtype=torch.int32).contiguous()
        key = (B, L)
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            ws = torch.empty(self._lib.ttsenc_workspace_bytes(self._h, B, L), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        memory = torch.empty(B, l_out, self.d_out, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.ttsenc_forward(self._h, ids.data_ptr(), lens.data_ptr(), B, L, l_out, memory.data_ptr(), ws.data_ptr(),
                                          ws.numel(), _stream(self.device), self.status.data_ptr())
        if rc != _lib.OK:
            raise _lib.TtsdecError(rc, "ttsenc_forward", self._lib.ttsenc_last_hip_error(self._h).decode())
        return memory


class _EncCache:
    def __init__(self):
        self.by_dev: Dict[int, EncoderEngine] = {}

    def __getstate__(self):
        return {}

    def __setstate__(self, st):
        self.by_dev = {}

    def __deepcopy__(self, memo):
        return _EncCache()


class Encoder2(PackedWeightsMixin, nn.Module):
    def __init__(self, alphabet_size, dim_out=512, dim_emb=512):
        super().__init__()
        self._watch_state_dict_loads()
        self.dim_out, self.dim_emb = dim_out, dim_emb
        self.emb = nn.Embedding(alphabet_size, dim_emb, padding_idx=0)
        self.conv = nn.Sequential(
            nn.Conv1d(dim_emb, dim_emb, kernel_size=5, padding=2, bias=False), nn.BatchNorm1d(dim_emb), _ISRLU(),
            nn.Conv1d(dim_emb, dim_emb, kernel_size=5, padding=2, bias=False), nn.BatchNorm1d(dim_emb), _ISRLU(),
            nn.Conv1d(dim_emb, dim_emb, kernel_size=5, padding=2, bias=False), nn.BatchNorm1d(dim_emb, affine=False), _ISRLU(),
        )
        self.rnn = _BiDiLSTM(dim_emb * 2, dim_out // 2, bias=False)
        self.rnn_h0 = nn.Parameter(torch.zeros(1, 1, dim_out))
        self.rnn_c0 = nn.Parameter(torch.zeros(1, 1, dim_out))
        self.use_hip = True  # eval-mode forwards on a ROCm device go through libttsdec
        self.precision = "f32"  # convs / input projection: "f32" (exact, the reference's own: default) or "split_f16" (opt-in)
        # nn.Embedding raises IndexError on an id outside the table; the HIP path finds out from a device status word.  False:
        # forward reads it before returning (one host sync); True: the caller reads it later through check_ids()
        # (Tacotron.forward does, at the decoder's own sync)
        self.defer_id_check = False
        self._engines = _EncCache()

    def weight_tensors(self):
        c, r = self.conv, self.rnn.rnn
        return [
            self.emb.weight,
            c[0].weight, c[1].weight, c[1].bias, c[1].running_mean, c[1].running_var,
            c[3].weight, c[4].weight, c[4].bias, c[4].running_mean, c[4].running_var,
            c[6].weight, c[7].running_mean, c[7].running_var,
            r.weight_ih_l0, r.weight_hh_l0, r.weight_ih_l0_reverse, r.weight_hh_l0_reverse,
            self.rnn_h0, self.rnn_c0,
        ]

    def _stock_forward(self, x, x_lengths):
        x = self.emb(x)
        xc = self.conv(x.mT).mT
        x = torch.cat((xc, x), dim=2)
        x = nn.functional.dropout(x, p=0.1, training=self.training)
        B = x.shape[0]
        x, _ = self.rnn(x, x_lengths, self.rnn_h0.expand(-1, B, -1), self.rnn_c0.expand(-1, B, -1))
        return x

    def forward(self, x, x_lengths):
        hip_ok = (
            self.use_hip and x.is_cuda and not self.training
            and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
            and self.dim_emb % 4 == 0 and self.dim_out % 8 == 0
        )
        if not hip_ok:
            return self._stock_forward(x, x_lengths)
        idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
        eng = self._engines.by_dev.get(idx)
        if eng is None:
            eng = EncoderEngine(self.emb.num_embeddings, self.dim_emb, self.dim_out, float(self.conv[1].eps), torch.device("cuda", idx))
            self._engines.by_dev[idx] = eng
        eng.ensure_packed(self.weight_tensors())
        eng.set_precision(self.precision)
        memory = eng.forward(x, x_lengths)
        if not self.defer_id_check:
            eng.check_ids()  # (one sync, behind the queued launches; Tacotron.forward defers it to the decoder's sync)
        return memory

    def check_ids(self) -> None:
        """The deferred id range check of the last HIP forward (see defer_id_check)."""
        for eng in self._engines.by_dev.values():
            eng.check_ids()
