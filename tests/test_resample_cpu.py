"""oracle/resample_ref.py: the float64 restatement of librosa's kaiser_best resampling (convert.py:54-56 ->
librosa.load(sr=...)).  PARITY UNPINNED: resampy / librosa are absent and the reference holds no audio fixture, so the
restatement is checked through properties of the published algorithm: output length, identity at equal rates,
pass-band fidelity of tones, rejection above the new Nyquist, and agreement with scipy's polyphase resampler
(a different Kaiser-windowed sinc design) inside the common pass band."""
import numpy as np
import scipy.signal as ss

from oracle import resample_ref as R


def tone(f, sr, n):
    return np.sin(2 * np.pi * f * np.arange(n) / sr)


def test_lengths_and_identity():
    x = tone(300, 44100, 4411)
    assert R.resample(x, 44100, 16000).shape == (int(np.ceil(4411 * 16000 / 44100)),)
    assert R.resample(x, 8000, 16000).shape == (8822,)
    assert np.array_equal(R.resample(x, 16000, 16000), x)
    w, nb = R.sinc_window(**R.KAISER_BEST)
    assert nb == 512 and w.shape == (64 * 512 + 1,) and abs(w[0] - R.KAISER_BEST["rolloff"]) < 1e-15 and abs(w[-1]) < 1e-7


def test_passband_tones_and_stopband():
    for sr0, sr1, tol in ((44100, 16000, 6e-3), (48000, 16000, 6e-3), (22050, 16000, 6e-3), (8000, 16000, 1e-5)):
        n = sr0 // 8
        for f in (200.0, 1000.0, 3000.0):
            y = R.resample(tone(f, sr0, n), sr0, sr1)
            want = tone(f, sr1, len(y))
            assert np.abs(y[300:-300] - want[300:-300]).max() <= tol, (sr0, sr1, f)   # resampy's integer index step
    y = R.resample(tone(12000.0, 44100, 6000), 44100, 16000)                            # above the new Nyquist: removed
    assert np.sqrt(np.mean(y[300:-300] ** 2)) < 1e-3


def test_close_to_scipy_polyphase_in_the_passband():
    rng = np.random.default_rng(13)
    x = ss.lfilter(*ss.butter(6, 0.2), rng.standard_normal(9600))                       # band-limited noise (< 0.2 x 24 kHz)
    y = R.resample(x, 48000, 16000)
    z = ss.resample_poly(x, 1, 3, window=("kaiser", 14.769656459379492))
    m = min(len(y), len(z))
    assert np.abs(y[300:m - 300] - z[300:m - 300]).max() <= 2e-2 * np.abs(z).max()
