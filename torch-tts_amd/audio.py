"""Mel -> waveform step after the decoder path: the inference half of the reference's
``AudioFrontend`` (tacotron/data/audio.py:25-76: ``mel_inv``, ``decode``), ``m_rev`` (data/dataset.py:183-184)
and ``synth_audio`` (inference.py:13-22).

The reference builds these from torchaudio (``InverseMelScale``, ``GriffinLim``, ``DB_to_amplitude``,
``amplitude_to_DB``; torchaudio >= 2.2.1 per tacotron/requirements.txt), which is not installed here, so the
published algorithms are restated on plain torch ops and run on whatever device the tensors live on (FFTs go
to rocFFT through ``torch.stft`` / ``torch.istft``): this step is I/O + FFT, there is no hand-written kernel
in it.  PARITY UNPINNED: without torchaudio no reference vectors can be produced, and the reference's
Griffin-Lim starts from a random phase (``rand_init=True``); tests check the algebraic properties instead."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class AudioFrontendConfig:  # data/audio.py:9-22
    sample_rate: int = 16000
    hop_length: int = 256
    win_length: int = 768
    num_mels: int = 80
    fmin: int = 50
    fmax: int = 7600

    def from_json(self, json):
        for key in json:
            self.__setattr__(key, json[key])
        return self


def m_fwd(x):  # data/dataset.py:179-180
    return torch.clip((x + 100) / 100, min=0)


def m_rev(x):  # data/dataset.py:183-184
    return (x * 100) - 100


def _hz_to_mel_slaney(f: float) -> float:
    f_sp = 200.0 / 3
    if f < 1000.0:
        return f / f_sp
    return 1000.0 / f_sp + math.log(f / 1000.0) / (math.log(6.4) / 27.0)


def _mel_to_hz_slaney(m: torch.Tensor) -> torch.Tensor:
    f_sp = 200.0 / 3
    min_log_mel = 1000.0 / f_sp
    logstep = math.log(6.4) / 27.0
    return torch.where(m >= min_log_mel, 1000.0 * torch.exp(logstep * (m - min_log_mel)), f_sp * m)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """Triangular mel filterbank [n_freqs, n_mels], Slaney scale and Slaney area normalisation
    (what MelScale / InverseMelScale(mel_scale="slaney", norm="slaney") use, data/audio.py:34-51)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_pts = torch.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2, dtype=torch.float64)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0)
    fb = fb * (2.0 / (f_pts[2 : n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb.to(torch.float32)


def db_to_amplitude(x: torch.Tensor, ref: float, power: float) -> torch.Tensor:
    return ref * torch.pow(torch.pow(10.0, 0.1 * x), power)


def amplitude_to_db(x: torch.Tensor, multiplier: float, amin: float, db_multiplier: float) -> torch.Tensor:
    return multiplier * torch.log10(torch.clamp(x, min=amin)) - multiplier * db_multiplier


def griffinlim(specgram: torch.Tensor, n_fft: int, hop_length: int, win_length: int, power: float = 2.0, n_iter: int = 32,
               momentum: float = 0.99, rand_init: bool = True, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Fast Griffin-Lim (Perraudin et al. 2013) as in torchaudio.functional.griffinlim: specgram [..., n_freqs, frames]
    (power spectrogram for power=2) -> waveform [..., time]."""
    window = torch.hann_window(win_length, device=specgram.device, dtype=specgram.dtype)
    shape = specgram.shape
    spec = specgram.reshape(-1, shape[-2], shape[-1]).pow(1.0 / power)
    mom = momentum / (1 + momentum)
    if rand_init:
        re = torch.rand(spec.shape, generator=generator, device=spec.device if generator is None or generator.device.type != "cpu" else "cpu")
        im = torch.rand(spec.shape, generator=generator, device=re.device)
        angles = torch.complex(re, im).to(spec.device)
    else:
        angles = torch.full(spec.shape, 1.0, dtype=torch.complex64, device=spec.device)
    tprev = torch.zeros_like(angles)
    for _ in range(n_iter):
        inverse = torch.istft(spec * angles, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window)
        rebuilt = torch.stft(inverse, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, center=True,
                             pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        angles = rebuilt
        if momentum:
            angles = angles - tprev * mom
        angles = angles / (angles.abs() + 1e-16)
        tprev = rebuilt
    wave = torch.istft(spec * angles, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window)
    return wave.reshape(shape[:-2] + wave.shape[-1:])


class AudioFrontend:
    """Inference half of data/audio.py's AudioFrontend: ``mel_inv`` (mel dB -> linear dB) and ``decode``
    (linear dB -> waveform by Griffin-Lim)."""

    def __init__(self, config: AudioFrontendConfig, device: Optional[torch.device] = None):
        self.config = config
        self.n_fft = config.win_length
        self.n_freqs = self.n_fft // 2 + 1
        self.fb = melscale_fbanks(self.n_freqs, float(config.fmin), float(config.fmax), config.num_mels, config.sample_rate)
        if device is not None:
            self.fb = self.fb.to(device)

    def stft_to_mels(self, D: torch.Tensor) -> torch.Tensor:  # MelScale: [..., n_freqs, T] -> [..., n_mels, T]
        return torch.matmul(D.transpose(-1, -2), self.fb.to(D.device)).transpose(-1, -2)

    def mels_to_stft(self, M: torch.Tensor) -> torch.Tensor:
        """InverseMelScale (torchaudio >= 2.1): least-squares solution of fb^T D = M, clamped at zero."""
        fbT = self.fb.to(M.device).transpose(-1, -2)  # [n_mels, n_freqs]
        lead = M.shape[:-2]
        m2 = M.reshape(-1, M.shape[-2], M.shape[-1])
        # driver "gels" is torchaudio's default and the only one the GPU backend offers; LAPACK's default
        # (gelsy) returns nothing useful here because the filterbank has all-zero rows above f_max
        sol = torch.linalg.lstsq(fbT.unsqueeze(0).expand(m2.shape[0], -1, -1).contiguous(), m2, driver="gels").solution
        return torch.relu(sol).reshape(lead + sol.shape[-2:])

    def mel_inv(self, M_db: torch.Tensor) -> torch.Tensor:  # data/audio.py:73-76
        M = db_to_amplitude(M_db.mT, 1, 1)
        D = self.mels_to_stft(M)
        return amplitude_to_db(D, 10, 1e-12, 0)

    def decode(self, D_db: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:  # data/audio.py:69-71
        D = db_to_amplitude(D_db, 1, 1)
        return griffinlim(D, self.n_fft, self.config.hop_length, self.n_fft, power=2.0, generator=generator)


def synth_audio(y: torch.Tensor, audio_frontend: AudioFrontend, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """inference.py:13-22: y [B, T, n_mels] (the model's normalised mel) -> waves [B, time], each peak-normalised."""
    wave = []
    for y_i in y:
        D_db = audio_frontend.mel_inv(m_rev(y_i))
        w = audio_frontend.decode(D_db, generator=generator)
        wave.append(w / w.abs().max())
    return torch.stack(wave)
