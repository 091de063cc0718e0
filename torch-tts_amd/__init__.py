"""MI355X-native drop-in for the Tacotron decoder hot path of kgoba/torch-tts.

Import as ``torch_tts_amd`` (the directory name ``torch-tts_amd`` is not a Python
identifier; the sibling ``torch_tts_amd/`` package aliases it)."""
from . import _lib  # noqa: F401
from .decoder import Decoder
from .decoder_cell import LSTMZoneoutCell, PreNet, StepwiseMonotonicAttention, Taco2DecoderCell, Taco2ProdDecoderCell
from .engine import Engine, EngineDims
from .postnet import Conv1dFix, MelPostnet, MelPostnet2
from .tacotron import Encoder2, Tacotron, build_tacotron, lengths_to_mask
from . import vits2  # noqa: F401  (TextEncoder, ResidualCouplingTransformersBlock)
from . import audio  # noqa: F401  (AudioFrontend.mel_inv / decode, m_rev, synth_audio)
from . import train_util  # noqa: F401  (load_state_dict: the reference's partial checkpoint loader + blob invalidation)

__all__ = [
    "Decoder", "Taco2ProdDecoderCell", "Taco2DecoderCell", "PreNet", "LSTMZoneoutCell", "StepwiseMonotonicAttention", "MelPostnet", "MelPostnet2",
    "Tacotron", "Encoder2", "build_tacotron", "lengths_to_mask", "Engine", "EngineDims",
]
