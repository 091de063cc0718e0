"""Host-side owner of one ttsdec handle: packs an nn.Module's parameters into the
kernel blob, holds workspaces (PyTorch caching allocator memory) and issues the
C-ABI calls on torch's current HIP stream."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

Tensor = torch.Tensor


@dataclass(frozen=True)
class EngineDims:
    d_mel: int = 80
    r: int = 1
    d_pre: int = 256
    d_ctx: int = 512
    h_att: int = 1024
    h_dec: int = 1024
    p_zoneout: float = 0.1
    p_dropout: float = 0.5
    postnet_layers: int = 0
    postnet_hidden: int = 512
    postnet_kernel: int = 5
    bn_eps: float = 1e-5
    cell_type: int = 0      # _lib.CELL_*
    d_pre_hidden: int = 0   # 0 = d_pre
    postnet_type: int = 0   # _lib.POSTNET_TYPE_*

    def to_c(self) -> _lib.Dims:
        return _lib.Dims(
            self.d_mel, self.r, self.d_pre, self.d_ctx, self.h_att, self.h_dec, self.p_zoneout, self.p_dropout,
            self.postnet_layers, self.postnet_hidden, self.postnet_kernel, self.bn_eps,
            self.cell_type, self.d_pre_hidden, self.postnet_type,
        )

    @property
    def pre_hidden(self) -> int:
        return self.d_pre_hidden or self.d_pre

    @property
    def cell_output(self) -> int:
        return self.h_att + self.h_dec + self.d_ctx if self.cell_type == _lib.CELL_TACO2 else self.h_dec + self.d_ctx


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_device(t: Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} must live on a ROCm device (got {t.device}): this package runs the decoder hot path "
            "only through its HIP kernels and has no CPU fallback"
        )


_EPOCH = [0]


def invalidate_packed_weights() -> None:
    """Forces every engine to repack its weight blob on next use.  Needed after edits that autograd's
    version counter does not see: ``param.data.copy_(...)`` / ``param.data.mul_(...)`` leave
    ``param._version`` unchanged (the reference's checkpoint loader does exactly that,
    tacotron/train_util.py:43; so do EMA swaps).  ``load_state_dict`` calls this by itself (hook)."""
    _EPOCH[0] += 1


def _invalidate_hook(module, incompatible_keys) -> None:  # module-level: stays picklable
    invalidate_packed_weights()


class PackedWeightsMixin:
    """For the nn.Modules whose parameters are mirrored in a packed device blob."""

    def invalidate(self) -> None:
        """Call after modifying parameters through ``.data`` (see invalidate_packed_weights)."""
        invalidate_packed_weights()

    def _watch_state_dict_loads(self) -> None:
        self.register_load_state_dict_post_hook(_invalidate_hook)


def weights_fingerprint(tensors: Sequence[Optional[Tensor]]) -> Tuple:
    """Key of a packed blob: changes when a parameter is replaced, moved, or modified in place through
    autograd-visible ops (``_version``), and when invalidate_packed_weights() was called.  In-place edits
    through ``.data`` are NOT visible here - call ``module.invalidate()`` after them."""
    return (_EPOCH[0],) + tuple((None if t is None else (t.data_ptr(), t._version, tuple(t.shape))) for t in tensors)


class Engine:
    """One ttsdec handle bound to one device.  Not thread-safe (the C ABI asks the
    caller to serialise calls per handle)."""

    def __init__(self, dims: EngineDims, device: Optional[torch.device]):
        self.dims = dims
        self.device = device
        self._lib = _lib.load()
        h = C.c_void_p()
        cd = dims.to_c()
        if device is not None:
            with torch.cuda.device(device):
                rc = self._lib.ttsdec_create(C.byref(cd), C.byref(h))
        else:
            rc = self._lib.ttsdec_create(C.byref(cd), C.byref(h))
        _lib.check(rc, "ttsdec_create")
        self._h = h
        self.blob: Optional[Tensor] = None
        self._fingerprint = None
        self._ws: Dict[Tuple[int, int], Tensor] = {}
        self._pws: Dict[Tuple[int, int], Tensor] = {}

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.ttsdec_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- host-only queries (work without a GPU) ----
    def num_weight_tensors(self) -> int:
        return self._lib.ttsdec_num_weight_tensors(self._h)

    def packed_bytes(self) -> int:
        return self._lib.ttsdec_packed_bytes(self._h)

    def workspace_bytes(self, B: int, L: int) -> int:
        return self._lib.ttsdec_workspace_bytes(self._h, B, L)

    def postnet_workspace_bytes(self, B: int, T: int) -> int:
        return self._lib.ttsdec_postnet_workspace_bytes(self._h, B, T)

    # ---- arithmetic of the LSTM gate GEMMs ----
    def set_precision(self, mode: str) -> None:
        """"f32" (exact fp32 matrix instruction) or "split_f16" (hi/lo fp16 planes, 3 products)."""
        code = {"f32": _lib.PREC_F32, "split_f16": _lib.PREC_SPLIT_F16}[mode]
        _lib.check(self._lib.ttsdec_set_precision(self._h, code), "ttsdec_set_precision")

    def precision(self) -> str:
        return {_lib.PREC_F32: "f32", _lib.PREC_SPLIT_F16: "split_f16"}[self._lib.ttsdec_get_precision(self._h)]

    # ---- tuning / measurement options (include/ttsdec.h TTSDEC_OPT_*; -1 = library default) ----
    def set_option(self, name: str, value: int) -> None:
        _lib.check(self._lib.ttsdec_set_option(self._h, _lib.option_ids()[name], int(value)), f"ttsdec_set_option({name})")

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        _lib.check(self._lib.ttsdec_get_option(self._h, _lib.option_ids()[name], C.byref(v)), f"ttsdec_get_option({name})")
        return int(v.value)

    # ---- weights ----
    def pack(self, tensors: Sequence[Optional[Tensor]]) -> Tensor:
        """tensors: in TTSDEC_W_* order (None = not owned by the calling module)."""
        n = self.num_weight_tensors()
        if len(tensors) != n:
            raise ValueError(f"expected {n} weight tensors, got {len(tensors)}")
        keep: List[Tensor] = []
        arr = (C.c_void_p * n)()
        for i, t in enumerate(tensors):
            if t is None:
                arr[i] = None
                continue
            _require_device(t, "weights")
            tc = t.detach()
            if tc.dtype != torch.float32 or not tc.is_contiguous():
                tc = tc.to(torch.float32).contiguous()
            keep.append(tc)
            arr[i] = tc.data_ptr()
        with torch.cuda.device(self.device):
            blob = torch.empty(self.packed_bytes(), dtype=torch.uint8, device=self.device)
            rc = self._lib.ttsdec_pack_weights(self._h, arr, n, blob.data_ptr(), _stream(self.device))
        _lib.check(rc, "ttsdec_pack_weights", self._h)
        self.blob = blob
        self._fingerprint = weights_fingerprint(tensors)
        del keep  # stream-ordered: the caching allocator keeps the memory valid for the enqueued copies
        return blob

    def ensure_packed(self, tensors: Sequence[Optional[Tensor]]) -> None:
        if self.blob is None or self._fingerprint != weights_fingerprint(tensors):
            self.pack(tensors)

    def bind(self, blob: Tensor) -> None:
        """Adopt a packed blob produced elsewhere (e.g. broadcast from rank 0)."""
        _require_device(blob, "blob")
        if blob.numel() * blob.element_size() != self.packed_bytes():
            raise ValueError("blob size does not match this engine's dims")
        with torch.cuda.device(self.device):
            torch.cuda.current_stream(self.device).synchronize()  # the blob must be complete: its header is read back
            _lib.check(self._lib.ttsdec_bind_weights(self._h, blob.data_ptr()), "ttsdec_bind_weights", self._h)
        self.blob = blob
        self._fingerprint = None

    # ---- workspaces ----
    def workspace(self, B: int, L: int) -> Tensor:
        key = (B, L)
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            ws = torch.empty(self.workspace_bytes(B, L), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def postnet_workspace(self, B: int, T: int) -> Tensor:
        key = (B, T)
        ws = self._pws.get(key)
        if ws is None:
            self._pws.clear()
            ws = torch.empty(self.postnet_workspace_bytes(B, T), dtype=torch.uint8, device=self.device)
            self._pws[key] = ws
        return ws

    # ---- compute ----
    def decode(
        self,
        memory: Tensor,
        *,
        t_begin: int,
        n_steps: int,
        stop_threshold: float,
        check_stop: bool,
        dropout_mode: int,
        masks: Optional[Tensor],
        seed: int,
        teacher: Optional[Tensor],
        teacher_flags: Optional[Tensor],
        y: Tensor,
        s: Tensor,
        w: Tensor,
        t_out: Tensor,
    ) -> None:
        """Enqueues n_steps decode steps; outputs land in y/s/w rows [0, n_steps)."""
        _require_device(memory, "memory")
        B, L, _ = memory.shape
        t_stride = w.shape[1]
        ws = self.workspace(B, L)
        with torch.cuda.device(self.device):
            rc = self._lib.ttsdec_decode(
                self._h, memory.data_ptr(), B, L, t_begin, n_steps, t_stride,
                float(stop_threshold), int(check_stop), dropout_mode, _ptr(masks), seed & 0xFFFFFFFFFFFFFFFF,
                _ptr(teacher), 0 if teacher is None else teacher.shape[1], _ptr(teacher_flags),
                y.data_ptr(), s.data_ptr(), w.data_ptr(), t_out.data_ptr(),
                ws.data_ptr(), ws.numel(), _stream(self.device),
            )
        _lib.check(rc, "ttsdec_decode", self._h)

    def postnet(self, y: Tensor, precision: int = _lib.POSTNET_F32) -> Tensor:
        _require_device(y, "y")
        B, T, _ = y.shape
        out = torch.empty_like(y)
        ws = self.postnet_workspace(B, T)
        with torch.cuda.device(self.device):
            rc = self._lib.ttsdec_postnet(
                self._h, y.data_ptr(), B, T, precision, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(self.device)
            )
        _lib.check(rc, "ttsdec_postnet", self._h)
        return out

    def cell_step(self, x, memory, w, ctx, h_att, c_att, h_dec, c_dec, dropout_mode, masks, seed, step) -> Tensor:
        _require_device(memory, "memory")
        B, L, _ = memory.shape
        x_dec = torch.empty(B, self.dims.cell_output, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, L)
        with torch.cuda.device(self.device):
            rc = self._lib.ttsdec_cell_step(
                self._h, x.data_ptr(), memory.data_ptr(), B, L, w.data_ptr(), ctx.data_ptr(), h_att.data_ptr(),
                c_att.data_ptr(), h_dec.data_ptr(), c_dec.data_ptr(), dropout_mode, _ptr(masks),
                seed & 0xFFFFFFFFFFFFFFFF, step, x_dec.data_ptr(), ws.data_ptr(), ws.numel(), _stream(self.device),
            )
        _lib.check(rc, "ttsdec_cell_step", self._h)
        return x_dec

    def profile_step(self, memory: Tensor, iters: int, dropout_mode: int, masks: Optional[Tensor], seed: int):
        """Mean per-kernel duration (ms) of one decode step; see ttsdec_profile_step."""
        B, L, _ = memory.shape
        d = self.dims
        y = torch.empty(B, 2 * d.r, d.d_mel, device=self.device)  # (the profiled step is step 1 of 2)
        s = torch.empty(B, 2 * d.r, device=self.device)
        w = torch.empty(B, 2, L, device=self.device)
        ws = self.workspace(B, L)
        ms = (C.c_float * 16)()
        names = (C.c_char_p * 16)()
        nk = C.c_int(0)
        with torch.cuda.device(self.device):
            rc = self._lib.ttsdec_profile_step(
                self._h, memory.data_ptr(), B, L, iters, dropout_mode, _ptr(masks), seed & 0xFFFFFFFFFFFFFFFF,
                y.data_ptr(), s.data_ptr(), w.data_ptr(), ws.data_ptr(), ws.numel(), _stream(self.device),
                ms, names, 16, C.byref(nk),
            )
        _lib.check(rc, "ttsdec_profile_step", self._h)
        return {names[i].decode(): float(ms[i]) for i in range(nk.value)}


    def profile_loop(self, memory: Tensor, n_steps: int, dropout_mode: int, masks: Optional[Tensor], seed: int):
        """Per-launch time (ms) INSIDE the replayed graph of the step loop and the loop's time per step; see
        ttsdec_profile_loop (n_steps: a multiple of 30, at least 60)."""
        B, L, _ = memory.shape
        d = self.dims
        y = torch.empty(B, n_steps * d.r, d.d_mel, device=self.device)
        s = torch.empty(B, n_steps * d.r, device=self.device)
        w = torch.empty(B, n_steps, L, device=self.device)
        t_out = torch.zeros(2, dtype=torch.int32, device=self.device)
        ws = self.workspace(B, L)
        ms = (C.c_float * 16)()
        names = (C.c_char_p * 16)()
        nk = C.c_int(0)
        step = C.c_float(0.0)
        with torch.cuda.device(self.device):
            rc = self._lib.ttsdec_profile_loop(
                self._h, memory.data_ptr(), B, L, n_steps, dropout_mode, _ptr(masks), seed & 0xFFFFFFFFFFFFFFFF,
                y.data_ptr(), s.data_ptr(), w.data_ptr(), t_out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(self.device),
                ms, names, 16, C.byref(nk), C.byref(step),
            )
        _lib.check(rc, "ttsdec_profile_loop", self._h)
        return {names[i].decode(): float(ms[i]) for i in range(nk.value)}, float(step.value)


class EngineCache:
    """Per-device engines of one module.  Lives in the module's __dict__ but is
    dropped on pickling / deepcopy, and shared (keyed by device) by the replicas
    nn.DataParallel makes (train_util.py:215 in the reference)."""

    def __init__(self):
        self._by_dev: Dict[int, Engine] = {}

    def get(self, dims: EngineDims, device: torch.device) -> Engine:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        e = self._by_dev.get(idx)
        if e is None or e.dims != dims:
            e = Engine(dims, torch.device("cuda", idx))
            self._by_dev[idx] = e
        return e

    def clear(self) -> None:
        for e in self._by_dev.values():
            e.close()
        self._by_dev.clear()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self._by_dev = {}

    def __deepcopy__(self, memo):
        return EngineCache()
