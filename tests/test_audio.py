"""Mel -> waveform step (torch-tts_amd/audio.py).  Parity is unpinned (torchaudio, which the
reference builds this step from, is not installed here and its Griffin-Lim starts from a random phase), so these
tests check the algebra: the Slaney filterbank, the least-squares mel inversion, Griffin-Lim's convergence."""
import math

import pytest
import torch

import torch_tts_amd as T

A = T.audio


def _sig(sr=22050, n=22050):
    t = torch.arange(n) / sr
    return 0.5 * torch.sin(2 * math.pi * 440 * t) + 0.3 * torch.sin(2 * math.pi * 1200 * t) + 0.1 * torch.sin(2 * math.pi * 3100 * t)


def _frontend(device=None):
    return A.AudioFrontend(A.AudioFrontendConfig(sample_rate=22050, hop_length=256, win_length=1024, num_mels=80, fmin=0, fmax=8000), device)


def test_slaney_filterbank_shape_and_normalisation():
    fe = _frontend()
    fb = fe.fb
    assert fb.shape == (513, 80) and float(fb.min()) == 0.0
    # triangles: every filter has one peak, neighbours overlap, nothing above f_max
    assert bool((fb.argmax(0)[1:] > fb.argmax(0)[:-1]).all())
    assert float(fb[int(8000 / (22050 / 2) * 512) + 2:].abs().max()) == 0.0
    # Slaney area normalisation: integral of each triangle over Hz is 1 (to the bin resolution)
    hz_per_bin = (22050 // 2) / 512
    area = fb.sum(0) * hz_per_bin
    assert float((area[5:] - 1).abs().max()) < 0.08
    assert float(A.m_rev(A.m_fwd(torch.tensor([-80.0, -3.0]))).sub(torch.tensor([-80.0, -3.0])).abs().max()) < 1e-5


def test_mel_inversion_is_a_least_squares_solution_and_griffinlim_converges():
    fe = _frontend()
    x = _sig()
    w = torch.hann_window(1024)
    S = torch.stft(x, 1024, 256, 1024, w, return_complex=True).abs().pow(2)
    M = fe.stft_to_mels(S)
    D = fe.mels_to_stft(M)
    assert D.shape == S.shape and float(D.min()) >= 0.0
    assert float((fe.stft_to_mels(D) - M).abs().max() / M.abs().max()) < 0.15  # (the clamp at zero costs a few percent)
    g = torch.Generator().manual_seed(0)
    wave = A.griffinlim(S, 1024, 256, 1024, generator=g)
    S2 = torch.stft(wave, 1024, 256, 1024, w, return_complex=True).abs()
    n = min(S.shape[-1], S2.shape[-1])
    conv = float((S2[:, :n] - S[:, :n].sqrt()).norm() / S[:, :n].sqrt().norm())
    assert conv < 0.15, conv
    g2 = torch.Generator().manual_seed(0)
    assert torch.equal(wave, A.griffinlim(S, 1024, 256, 1024, generator=g2))  # deterministic under a seeded generator
    # the reference's chain: model mel (normalised dB) -> waveform, peak-normalised
    y = A.m_fwd(A.amplitude_to_db(M, 10, 1e-12, 0).mT).unsqueeze(0)
    out = A.synth_audio(y, fe, generator=g)
    assert out.dim() == 2 and out.shape[0] == 1 and abs(float(out.abs().max()) - 1.0) < 1e-6


@pytest.mark.gpu
def test_synth_audio_runs_on_the_gpu_and_matches_the_cpu_mel_inversion():
    fe_c, fe_g = _frontend(), _frontend(torch.device("cuda:0"))
    x = _sig()
    S = torch.stft(x, 1024, 256, 1024, torch.hann_window(1024), return_complex=True).abs().pow(2)
    M_db = A.amplitude_to_db(fe_c.stft_to_mels(S), 10, 1e-12, 0).mT
    dc = fe_c.mel_inv(M_db)
    dg = fe_g.mel_inv(M_db.cuda())
    # same least-squares problem on both devices (different LAPACK / rocSOLVER paths): compare in the linear domain
    lc, lg = A.db_to_amplitude(dc, 1, 1), A.db_to_amplitude(dg.cpu(), 1, 1)
    assert float((lc - lg).abs().max() / lc.abs().max()) < 1e-2
    y = A.m_fwd(M_db).unsqueeze(0).cuda()
    out = A.synth_audio(y, fe_g)
    assert out.is_cuda and bool(torch.isfinite(out).all()) and abs(float(out.abs().max()) - 1.0) < 1e-6
