"""Drop-in for the reference's ``decoder.Decoder`` (tacotron/decoder.py:5-77): same
constructor, attributes (``decoder_cell r dim_mel stop_threshold fc_mel fc_stop``),
state-dict keys and ``forward(memory, mmask, x=None, max_steps=0, p_no_forcing=None)
-> (y, s, w)``.  The whole step loop runs on the GPU through ``ttsdec_decode``; the
host syncs once per chunk of steps (once per call when max_steps is given) instead of
once per step."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib
from .engine import EngineCache, EngineDims, PackedWeightsMixin
from .rng import MaskStream


class Decoder(PackedWeightsMixin, nn.Module):
    def __init__(self, decoder_cell, r, dim_mel, stop_threshold=-2.0):
        super().__init__()
        self._watch_state_dict_loads()
        self.decoder_cell = decoder_cell
        self.stop_threshold = stop_threshold
        self.r = r
        self.dim_mel = dim_mel
        self.fc_mel = nn.Linear(decoder_cell.dim_output, self.r * self.dim_mel)
        self.fc_stop = nn.Linear(decoder_cell.dim_output, self.r)

        # Source of the PreNet's always-on dropout (tacotron/modules/modules.py:40):
        #   "reference_rng": replay the reference's draws from torch's default CPU generator,
        #                    so results equal the reference's under the same torch.manual_seed
        #   "philox":        on-device counter RNG keyed by (dropout_seed, step, layer, b, unit)
        #   "off":           no dropout (not what the reference does)
        self.dropout_source = "reference_rng"
        self.dropout_seed = 0
        # Arithmetic of the LSTM gate GEMMs: "f32" (exact) or "split_f16" (see include/ttsdec.h)
        self.precision = "f32"
        self.chunk_steps = 256      # steps per launch batch when max_steps == 0 (unbounded decode)
        self.max_decoder_steps = 0  # optional hard cap for unbounded decode (0 = none, like the reference)
        self._engines = EngineCache()

    def weight_tensors(self):
        return self.decoder_cell.weight_tensors() + [
            self.fc_mel.weight, self.fc_mel.bias, self.fc_stop.weight, self.fc_stop.bias,
        ]

    def engine(self, device):
        """The packed-weights engine for `device` (repacks if parameters changed)."""
        eng = self._engines.get(self.decoder_cell.engine_dims(), device)
        eng.ensure_packed(self.weight_tensors())
        eng.set_precision(self.precision)
        return eng

    def forward(self, memory, mmask, x=None, max_steps: int = 0, p_no_forcing: float = None):
        # memory: B x L x D_enc, x: B x T x D_mel (teacher frames) or None
        if not memory.is_cuda:
            raise RuntimeError(
                "Decoder runs on the HIP path only: move the model and its inputs to a ROCm device (no CPU fallback)"
            )
        if self.training or (torch.is_grad_enabled() and (memory.requires_grad or any(p.requires_grad for p in self.parameters()))):
            # training semantics (zoneout sampling, energy noise) or a graph to differentiate: composable torch ops on
            # the device (autograd_path.py); the HIP library is the forward-only, eval-mode hot path
            from . import autograd_path

            autograd_path.warn_eval_on_autograd_path(self)
            return autograd_path.decoder_forward(self, memory, mmask, x, max_steps, p_no_forcing)
        device = memory.device
        memory = memory.detach().to(torch.float32).contiguous()
        B, L, _ = memory.shape
        r, dm = self.r, self.dim_mel
        eng = self.engine(device)

        teacher, total_steps = None, None
        if x is not None:
            Tx = (x.shape[1] // r) * r
            total_steps = Tx // r  # len(x_split), decoder.py:41-42,62
            if total_steps < 1:
                raise ValueError("teacher input shorter than one decoder step")
            teacher = x[:, :Tx, :].detach().to(device=device, dtype=torch.float32).contiguous()
        elif max_steps:
            total_steps = int(max_steps) + 1  # decoder.py:68-71: breaks when step > max_steps
        cap = total_steps if total_steps is not None else (self.max_decoder_steps or None)
        check_stop = x is None

        if self.dropout_source == "reference_rng":
            mode = _lib.DROPOUT_MASKS
            stream = MaskStream(B, self.decoder_cell.dim_pre, self.decoder_cell.pre_net.p_dropout,
                                p_no_forcing=p_no_forcing if x is not None else None,
                                teacher_steps=total_steps if x is not None else None,
                                d_pre_hidden=getattr(self.decoder_cell, "dim_pre_hidden", None))
        elif self.dropout_source == "philox":
            mode, stream = _lib.DROPOUT_PHILOX, None
        elif self.dropout_source == "off":
            mode, stream = _lib.DROPOUT_OFF, None
        else:
            raise ValueError(f"unknown dropout_source {self.dropout_source!r}")
        if x is not None and p_no_forcing and stream is None:
            # teacher-forcing coin flips still come from the host generator, like the reference
            stream = MaskStream(B, self.decoder_cell.dim_pre, 0.0, p_no_forcing=p_no_forcing, teacher_steps=total_steps)

        chunk = max(2, int(self.chunk_steps) + (int(self.chunk_steps) & 1))  # even: ttsdec_decode ties buffer parity to t_begin

        def run():
            """The chunked decode from step 0.  Returns None when a two-role launch timed out (its outputs are invalid)."""
            ys: List[torch.Tensor] = []
            ss: List[torch.Tensor] = []
            wsl: List[torch.Tensor] = []
            flags_all = torch.ones(total_steps if x is not None else 0, dtype=torch.uint8)
            t_out = torch.zeros(2, dtype=torch.int32, device=device)
            t = 0
            produced = 0
            while True:
                n = chunk if cap is None else min(cap - t, chunk if total_steps is None else cap - t)
                if n <= 0:
                    break
                masks_dev = None
                if stream is not None:
                    masks, flags = stream.draw(n)
                    if mode == _lib.DROPOUT_MASKS:
                        masks_dev = masks.to(device, non_blocking=False)
                    if x is not None:
                        flags_all[t : t + n] = flags
                flags_dev = flags_all.to(device) if teacher is not None else None
                y = torch.empty(B, n * r, dm, dtype=torch.float32, device=device)
                s = torch.empty(B, n * r, dtype=torch.float32, device=device)
                w = torch.empty(B, n, L, dtype=torch.float32, device=device)
                eng.decode(
                    memory, t_begin=t, n_steps=n, stop_threshold=float(self.stop_threshold), check_stop=check_stop,
                    dropout_mode=mode, masks=masks_dev, seed=int(self.dropout_seed), teacher=teacher,
                    teacher_flags=flags_dev, y=y, s=s, w=w, t_out=t_out,
                )
                done, flags = (int(v) for v in t_out.tolist())  # the one host sync of this chunk
                fired = flags & 1
                if flags & 4:
                    return None
                if flags & 2:
                    raise RuntimeError(
                        "split_f16 precision: an activation (input/teacher frame, PreNet output or context) exceeded the fp16 "
                        "range (|x| > 65504) and was saturated; set decoder.precision = 'f32' for such inputs"
                    )
                k = done - t
                ys.append(y[:, : k * r])
                ss.append(s[:, : k * r])
                wsl.append(w[:, :k])
                produced = done
                t += n
                if fired or (cap is not None and t >= cap):
                    break
            return ys, ss, wsl, produced

        out = run()
        if out is None:
            # A consumer role of a two-role launch gave up waiting for its producer role (bounded spin: the GPU is shared or
            # stalled so badly that the roles were not co-resident).  The state in the workspace is unusable, but a decode
            # call restarts from its inputs: switch this engine to one role per launch and run the call again.
            import warnings

            warnings.warn("decode step: a multi-role launch timed out waiting for its producer role; "
                          "this engine now runs one role per launch (options overlap = 0, head_proj = 0) and the call is repeated", RuntimeWarning)
            eng.set_option("overlap", 0)
            eng.set_option("head_proj", 0)  # (the projection would otherwise stay a role at the head of the frame kernel's launch)
            if stream is not None:
                stream.restart()
            out = run()
            if out is None:
                raise RuntimeError("decode step: a launch timed out waiting for its producer role even with one role per launch; results discarded")
        ys, ss, wsl, produced = out
        if stream is not None:
            stream.rewind_to(produced)
        y = ys[0] if len(ys) == 1 else torch.cat(ys, dim=1)
        s = ss[0] if len(ss) == 1 else torch.cat(ss, dim=1)
        w = wsl[0] if len(wsl) == 1 else torch.cat(wsl, dim=1)
        return y, s.unsqueeze(2), w  # B x T x D_mel, B x T x 1, B x T x L
