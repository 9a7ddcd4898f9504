"""CPU restatement of the reference's mel front-end (TEST INFRASTRUCTURE, like the rest of oracle/).

Reference: ``preprocess.py:16-17`` (pre-emphasis = ``scipy.signal.lfilter([1, -a], [1], x)``),
``preprocess.py:53-75`` / ``convert.py:54-70`` (peak-normalise to 0.999, ``librosa.feature.melspectrogram``
with n_fft 2048 / win 400 / hop 160 / 80 mels / fmin 50 / power 1, ``librosa.amplitude_to_db(top_db=80)``,
``/ top_db + 1``), parameters ``config.py:103-112``.

librosa (^0.8.0, ``pyproject.toml:18``) is not installed here and cannot be fetched: **parity unpinned**.
The functions below restate librosa 0.8's published algorithm (centered STFT with reflect padding and a
periodic Hann window zero-padded to n_fft; Slaney mel filterbank, area-normalised; amplitude_to_db with
amin 1e-5, ref 1.0, top_db clipping against the utterance maximum) in float64 numpy, as the reference runs it
(scipy's lfilter returns float64, so librosa computes in float64).  ``tests/test_mel_cpu.py`` cross-checks the
STFT against ``torch.stft``.
"""
import numpy as np


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-12) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=16000, n_fft=2048, n_mels=80, fmin=50.0, fmax=None):
    """librosa.filters.mel(htk=False, norm='slaney') -> (n_mels, 1 + n_fft//2) float32."""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def preemphasis(x, a=0.97):
    x = np.asarray(x, dtype=np.float64)
    y = x.copy()
    y[1:] -= a * x[:-1]
    return y


def stft_mag(y, n_fft=2048, hop=160, win=400):
    """|librosa.stft(y, center=True, pad_mode='reflect', window='hann')| -> (1 + n_fft//2, 1 + len//hop)."""
    y = np.asarray(y, dtype=np.float64)
    w = np.zeros(n_fft)
    lpad = (n_fft - win) // 2
    w[lpad: lpad + win] = hann_periodic(win)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return np.abs(np.fft.rfft(yp[idx] * w[None, :], axis=1)).T


def wave_to_mel(wave, sr=16000, n_fft=2048, n_mels=80, hop=160, win=400, fmin=50.0, preemph=0.97, top_db=80.0):
    """``preprocess.py:53-75`` -> float32 (n_mels, T) normalised log-mel."""
    wave = np.asarray(wave, dtype=np.float64)
    ws = wave / np.abs(wave).max() * 0.999
    S = stft_mag(preemphasis(ws, preemph), n_fft, hop, win)
    mel = mel_filterbank(sr, n_fft, n_mels, fmin).astype(np.float64) @ S
    logspec = 10.0 * np.log10(np.maximum(1e-10, mel * mel))          # amplitude_to_db: amin 1e-5, ref 1.0
    logspec = np.maximum(logspec, logspec.max() - top_db)
    return (logspec / top_db + 1.0).astype(np.float32)
