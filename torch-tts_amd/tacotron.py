"""Model assembly with the reference's API (tacotron/tacotron.py:20-56,165-224):
``Tacotron(encoder, decoder, postnet, refencoder).forward(...) -> (y, y_post, s,
{"w", "kl_loss"})`` and ``build_tacotron(config)``.  The decoder and postnet are the
HIP-backed drop-ins of this package; the encoder runs once per batch, is outside the
hot path and stays stock PyTorch-ROCm ops."""
from __future__ import annotations

import torch
import torch.nn as nn

from .decoder import Decoder
from .decoder_cell import Taco2DecoderCell, Taco2ProdDecoderCell
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


class Encoder2(nn.Module):
    """Text encoder producing ``memory`` (tacotron/encoder.py:27-82): embedding ->
    3 x (conv5 + BN + ISRLU) -> concat with the embedding -> BiLSTM.  Stock PyTorch."""

    def __init__(self, alphabet_size, dim_out=512, dim_emb=512):
        super().__init__()
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

    def forward(self, x, x_lengths):
        x = self.emb(x)
        xc = self.conv(x.mT).mT
        x = torch.cat((xc, x), dim=2)
        x = nn.functional.dropout(x, p=0.1, training=self.training)
        B = x.shape[0]
        x, _ = self.rnn(x, x_lengths, self.rnn_h0.expand(-1, B, -1), self.rnn_c0.expand(-1, B, -1))
        return x


class Tacotron(nn.Module):
    def __init__(self, encoder, decoder, postnet=None, refencoder=None):
        super().__init__()
        self.refencoder = refencoder
        self.encoder = encoder
        self.decoder = decoder
        self.postnet = postnet
        self.apply(weights_init)

    def forward(self, cond, cond_lengths, x=None, x_lengths=None, xref=None, xref_lengths=None, max_steps: int = 0):
        memory = self.encoder(cond, cond_lengths)
        kl_loss = torch.scalar_tensor(0)
        if xref is not None and self.refencoder is not None:
            style_embed, style_loss_dict = self.refencoder(xref, xref_lengths)
            memory = memory + style_embed
            if "kl" in style_loss_dict:
                kl_loss = style_loss_dict["kl"].mean()
        mmask = lengths_to_mask(cond_lengths)
        y, s, w = self.decoder(memory, mmask, x, max_steps, p_no_forcing=0.1)
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
