"""Model assembly with the reference's API (tacotron/tacotron.py:20-56,165-224):
``Tacotron(encoder, decoder, postnet, refencoder).forward(...) -> (y, y_post, s,
{"w", "kl_loss"})`` and ``build_tacotron(config)``.  The decoder and postnet are the
HIP-backed drop-ins of this package; so is the encoder in eval mode (torch-tts_amd/encoder.py;
it runs once per batch, before the hot path)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .decoder import Decoder
from .engine import PackedWeightsMixin
from .decoder_cell import Taco2DecoderCell, Taco2ProdDecoderCell
from .encoder import Encoder2
from .postnet import MelPostnet, MelPostnet2


def lengths_to_mask(lengths):
    """Boolean [B, max(lengths)] mask (tacotron/data/util.py:4-7)."""
    idx = torch.arange(int(lengths.max()), device=lengths.device)
    return idx.unsqueeze(0) < lengths.unsqueeze(1)


def weights_init(m):
    # tacotron.py:12-17
    if isinstance(m, (nn.Conv1d, nn.Linear)):
        if m.weight is not None:
            nn.init.xavier_normal_(m.weight, gain=1.5)
        if m.bias is not None:
            nn.init.zeros_(m.bias)


class Tacotron(PackedWeightsMixin, nn.Module):
    def __init__(self, encoder, decoder, postnet=None, refencoder=None):
        super().__init__()
        self.refencoder = refencoder
        self.encoder = encoder
        self.decoder = decoder
        self.postnet = postnet
        self.apply(weights_init)

    def fast_inference(self, seed: int = 0):
        """One switch for the fast configuration of the HIP path (inference only):
          * PreNet dropout drawn on the device (Philox, keyed by `seed`) instead of replaying the reference's CPU generator -
            the default `dropout_source = "reference_rng"` draws ~80 MB of masks on the host and uploads them per 256 x 600
            batch, which exists for bit-compatibility with the reference's RNG stream, not for speed;
          * split-fp16 arithmetic (fp32-class accuracy: 22 significand bits per operand, fp32 accumulate) for the decoder's
            the Postnet's and the text encoder's GEMMs instead of exact fp32 matrix instructions (about 0.6 of the step time at
            256 utterances).
        Returns self; `reference_compatible()` switches back."""
        self.decoder.dropout_source, self.decoder.dropout_seed, self.decoder.precision = "philox", int(seed), "split_f16"
        for m in (self.postnet, self.encoder):
            if m is not None and hasattr(m, "precision"):
                m.precision = "split_f16"
        return self

    def reference_compatible(self):
        """The defaults: the reference's RNG stream for the always-on PreNet dropout, exact fp32 arithmetic."""
        self.decoder.dropout_source, self.decoder.precision = "reference_rng", "f32"
        for m in (self.postnet, self.encoder):
            if m is not None and hasattr(m, "precision"):
                m.precision = "f32"
        return self

    def forward(self, cond, cond_lengths, x=None, x_lengths=None, xref=None, xref_lengths=None, max_steps: int = 0):
        defer = isinstance(self.encoder, Encoder2) and not self.training
        if defer:  # the encoder's id range check rides on the decoder's sync below instead of a sync of its own
            prev, self.encoder.defer_id_check = self.encoder.defer_id_check, True
        try:
            memory = self.encoder(cond, cond_lengths)
        finally:
            if defer:
                self.encoder.defer_id_check = prev
        kl_loss = torch.scalar_tensor(0)
        if xref is not None and self.refencoder is not None:
            style_embed, style_loss_dict = self.refencoder(xref, xref_lengths)
            memory = memory + style_embed
            if "kl" in style_loss_dict:
                kl_loss = style_loss_dict["kl"].mean()
        mmask = lengths_to_mask(cond_lengths)
        y, s, w = self.decoder(memory, mmask, x, max_steps, p_no_forcing=0.1)
        if defer:
            self.encoder.check_ids()  # (the decoder has just synchronised: the status word is there)
        y_post = self.postnet(y) if self.postnet else y
        return y, y_post, s, {"w": w, "kl_loss": kl_loss}


def build_tacotron(config):
    """config dict -> Tacotron, as tacotron.py:165-224 for the LJSpeech-style configs
    (decoder type other than tacotron1/tacotron2 -> Taco2ProdDecoderCell; postnet type
    "tacotron2" -> MelPostnet)."""
    text_config, audio_config = config["text"], config["audio"]
    decoder_config, encoder_config = config["model"]["decoder"], config["model"]["encoder"]
    if decoder_config["type"] == "tacotron1":
        raise NotImplementedError("Taco1DecoderCell is dead code in the reference (SURVEY.md section 2) and is not provided")
    decoder_cell_class = Taco2DecoderCell if decoder_config["type"] == "tacotron2" else Taco2ProdDecoderCell
    decoder_cell = decoder_cell_class(
        encoder_config["dim_out"], audio_config["num_mels"], r=decoder_config["r"], dim_rnn=decoder_config["dim_rnn"],
        dim_pre=decoder_config["dim_pre"], dim_att=decoder_config["dim_att"],
    )
    decoder = Decoder(decoder_cell, decoder_config["r"], audio_config["num_mels"])
    alphabet_size = 1 + len(text_config["alphabet"])
    if "phonemes" in text_config:
        alphabet_size += len(text_config["phonemes"])
    encoder = Encoder2(alphabet_size, dim_out=encoder_config["dim_out"], dim_emb=encoder_config["dim_emb"])
    postnet_config = config["model"].get("postnet")
    postnet = None
    if postnet_config:
        if postnet_config.get("type") == "tacotron2":
            postnet = MelPostnet(audio_config["num_mels"], dim_hidden=postnet_config["dim_hidden"], num_layers=postnet_config["num_layers"])
        else:
            postnet = MelPostnet2(audio_config["num_mels"], dim_hidden=postnet_config["dim_hidden"], num_layers=postnet_config["num_layers"])
    if config["model"].get("style_encoder"):
        raise NotImplementedError("style encoder (VAE) is outside the hot path and not provided")
    return Tacotron(encoder, decoder, postnet=postnet, refencoder=None)
