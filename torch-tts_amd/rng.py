"""Reference-compatible randomness for the decoder.

The reference's PreNet applies dropout(p=0.5) at inference too
(tacotron/modules/modules.py:40, always_dropout=True at decoder_cell.py:152), drawing
from torch's default CPU generator (its inference path pins device="cpu",
tacotron/inference.py:47).  Per decode step it makes two Bernoulli(1-p) draws of
shape [B, d_pre] in layer order, and in teacher mode with p_no_forcing one
``torch.rand(1)`` after every step but the last (tacotron/decoder.py:65).

``MaskStream`` replays exactly those draws on the host, so that under the same
``torch.manual_seed`` the HIP path consumes the same masks, makes the same
teacher-forcing decisions, and leaves the generator in the same state as the
reference would - including when the stop rule ends decoding early."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


class MaskStream:
    def __init__(self, B: int, d_pre: int, p_dropout: float = 0.5, p_no_forcing: Optional[float] = None,
                 teacher_steps: Optional[int] = None, generator: Optional[torch.Generator] = None,
                 d_pre_hidden: Optional[int] = None):
        self.B, self.d_pre, self.p = B, d_pre, p_dropout
        self.widths = (d_pre_hidden or d_pre, d_pre)  # PreNet layer 0 / layer 1 output widths
        self.p_no_forcing = p_no_forcing
        self.teacher_steps = teacher_steps  # total steps of the teacher-forced run, or None
        self.gen = generator
        self._states: List[torch.Tensor] = []  # generator state after each drawn step
        self._start = self._get_state()        # ... and before the first one (restart)
        self.steps_drawn = 0

    def _get_state(self) -> torch.Tensor:
        return self.gen.get_state() if self.gen is not None else torch.get_rng_state()

    def _set_state(self, st: torch.Tensor) -> None:
        if self.gen is not None:
            self.gen.set_state(st)
        else:
            torch.set_rng_state(st)

    def draw(self, n_steps: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Next n_steps of draws: keep-masks uint8, per step layer 0 [B, d_pre_hidden] followed by
        layer 1 [B, d_pre] (returned as [n_steps, 2, B, d_pre] when the widths are equal, else
        [n_steps, B*(d_pre_hidden + d_pre)]), and teacher flags uint8 [n_steps] (flags[i] != 0: the
        step after step i is fed the teacher frame; all ones when p_no_forcing is falsy)."""
        w0, w1 = self.widths
        masks = torch.empty(n_steps, self.B * (w0 + w1), dtype=torch.uint8)
        flags = torch.ones(n_steps, dtype=torch.uint8)
        tmps = (torch.empty(self.B, w0), torch.empty(self.B, w1))
        for i in range(n_steps):
            off = 0
            for layer in range(2):
                tmp = tmps[layer]
                tmp.bernoulli_(1.0 - self.p, generator=self.gen)
                masks[i, off : off + tmp.numel()] = tmp.to(torch.uint8).reshape(-1)
                off += tmp.numel()
            t = self.steps_drawn
            if self.teacher_steps is not None and self.p_no_forcing and t < self.teacher_steps - 1:
                u = torch.rand(1, generator=self.gen)
                flags[i] = 1 if bool(u > self.p_no_forcing) else 0
            self._states.append(self._get_state())
            self.steps_drawn += 1
        if w0 == w1:
            masks = masks.view(n_steps, 2, self.B, w1)
        return masks, flags

    def restart(self) -> None:
        """Back to before the first draw: the same draws come again (a decode call that is repeated)."""
        self._set_state(self._start)
        self._states.clear()
        self.steps_drawn = 0

    def rewind_to(self, steps_used: int) -> None:
        """Leave the generator as if only `steps_used` steps had ever been drawn."""
        if steps_used < self.steps_drawn and steps_used >= 1:
            self._set_state(self._states[steps_used - 1])
            self.steps_drawn = steps_used
            del self._states[steps_used:]
