"""Mel front-end oracle (librosa is absent: self-oracle, parity unpinned) -- internal cross-checks."""
import numpy as np
import torch

from oracle import mel_ref
from vectorquantizedcpc_amd import synth


def _wave(n, name="w"):
    u = synth.uniform01("wave/" + name, n)
    t = np.arange(n) / 16000.0
    return (0.3 * np.sin(2 * np.pi * 220 * t) + 0.1 * np.sin(2 * np.pi * 1850 * t + 1.0) + 0.05 * (u - 0.5)).astype(np.float32)


def test_stft_matches_torch_stft():
    y = mel_ref.preemphasis(_wave(4000) / 3.0)
    mine = mel_ref.stft_mag(y)
    ref = torch.stft(torch.from_numpy(y), n_fft=2048, hop_length=160, win_length=400,
                     window=torch.hann_window(400, periodic=True, dtype=torch.float64), center=True,
                     pad_mode="reflect", return_complex=True).abs().numpy()
    assert mine.shape == ref.shape == (1025, 1 + 4000 // 160)
    assert np.abs(mine - ref).max() <= 1e-9


def test_mel_filterbank_properties():
    w = mel_ref.mel_filterbank()
    assert w.shape == (80, 1025) and w.dtype == np.float32 and (w >= 0).all()
    freqs = np.linspace(0, 8000, 1025)
    centers = (w * freqs).sum(1) / w.sum(1)
    assert np.all(np.diff(centers) > 0) and centers[0] > 50 and centers[-1] < 8000
    assert w[:, freqs < 50].sum() == 0                      # fmin = 50 Hz (config.py:107)
    # Slaney area normalisation: every triangle integrates to ~1 Hz^-1 * Hz
    area = w.sum(1) * (freqs[1] - freqs[0])
    assert np.abs(area - 1.0).max() < 0.05
    assert abs(float(mel_ref.hz_to_mel_slaney(1000.0)) - 15.0) < 1e-9
    assert abs(float(mel_ref.mel_to_hz_slaney(mel_ref.hz_to_mel_slaney(4321.0))) - 4321.0) < 1e-6


def test_wave_to_mel_range_and_shape():
    mel = mel_ref.wave_to_mel(_wave(16000))
    assert mel.shape == (80, 101) and mel.dtype == np.float32
    assert mel.max() <= 1.0 + 20 * np.log10(2048) / 80 and mel.max() - mel.min() <= 1.0 + 1e-6   # top_db 80 -> span <= 1
    # gain invariance (the peak normalisation of preprocess.py:62)
    assert np.abs(mel_ref.wave_to_mel(_wave(16000) * 0.1) - mel).max() <= 1e-5
