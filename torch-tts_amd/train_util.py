"""Drop-in for the one function of the reference's ``tacotron/train_util.py`` that touches this package's contract:
``load_state_dict(model, state_dict)`` (train_util.py:23-45), the name-walking partial checkpoint loader that
``Trainer.load_checkpoint`` falls back to (train_util.py:205).

It writes through ``param.data.copy_`` - an edit autograd's version counter does not see - so a model that has already run a
HIP forward would keep decoding with the packed copy of its OLD weights.  This version loads the same way (every key it can
resolve, a warning for the rest) and then invalidates every packed blob (engine.invalidate_packed_weights), so the next
forward repacks.  ``nn.Module.load_state_dict`` needs none of this (the modules hook it).  Everything else in
train_util.py (Trainer, loss_loop) is training orchestration, outside the hot path."""
from __future__ import annotations

import logging

import torch

from .engine import invalidate_packed_weights

logger = logging.getLogger(__name__)


def load_state_dict(model: torch.nn.Module, state_dict) -> None:
    for key, value in state_dict.items():
        target = model
        try:
            for name in key.split("."):
                target = getattr(target, name)
            with torch.no_grad():
                target.data.copy_(value)
        except Exception as e:  # (the reference skips what it cannot set: a missing attribute, a shape mismatch)
            logger.warning(f"Did not set param {key}, skipping ({e})")
    invalidate_packed_weights()
