"""CPU restatement of the loudness steps of ``convert.py`` (TEST INFRASTRUCTURE, like the rest of oracle/).

Reference call sites: ``convert.py:50`` (``pyloudnorm.Meter(sr)``), ``:57`` (``meter.integrated_loudness(wav)`` on
the input), ``:79`` (on the generated audio), ``:80`` (``pyloudnorm.normalize.loudness(output, output_loudness,
ref_loudness)``).

pyloudnorm (``^0.1.0``, ``pyproject.toml:20``) is a third-party dependency that is not installed here and cannot be
fetched: **parity unpinned** against it.  The functions restate its published algorithm (ITU-R BS.1770-4 gated
integrated loudness as pyloudnorm 0.1 implements it for mono input): the "K-weighting" pair of biquads derived at
the stream's own rate (high shelf +4 dB, Q 1/sqrt(2), 1500 Hz; high pass Q 0.5, 38 Hz; RBJ-style formulas),
``scipy.signal.lfilter`` (direct form II transposed), 400 ms blocks with 75 % overlap whose sample bounds are
``int(T_g * (j * step) * rate)`` / ``int(T_g * (j * step + 1) * rate)``, block energy ``sum(x^2) / (T_g * rate)``,
absolute gate -70 LUFS, relative gate -10 LU, ``-0.691 + 10 log10(mean)``.  What anchors it is the standard's own
conformance point (``tests/test_loudness_cpu.py``): a 0 dBFS 997 Hz sine on one front channel reads -3.01 LKFS within
the 0.1 LU conformance tolerance (-3.05 here: the derived high-pass has unit passband gain where the BS.1770 table's
numerator [1, -2, 1] has +0.04 dB), -20 dB of gain moves the reading by exactly -20 LU, and at 48 kHz the derived
shelf coefficients agree with the BS.1770 table to 1e-4.  All arithmetic float64.
"""
import numpy as np
from scipy.signal import lfilter

BLOCK_S = 0.400
OVERLAP = 0.75
GAMMA_ABS = -70.0


def biquad(kind: str, G: float, Q: float, fc: float, rate: float):
    """-> (b[3], a[3]) normalised by a0."""
    A = 10.0 ** (G / 40.0)
    w0 = 2.0 * np.pi * (fc / rate)
    alpha = np.sin(w0) / (2.0 * Q)
    c = np.cos(w0)
    if kind == "high_shelf":
        b0 = A * ((A + 1) + (A - 1) * c + 2 * np.sqrt(A) * alpha)
        b1 = -2 * A * ((A - 1) + (A + 1) * c)
        b2 = A * ((A + 1) + (A - 1) * c - 2 * np.sqrt(A) * alpha)
        a0 = (A + 1) - (A - 1) * c + 2 * np.sqrt(A) * alpha
        a1 = 2 * ((A - 1) - (A + 1) * c)
        a2 = (A + 1) - (A - 1) * c - 2 * np.sqrt(A) * alpha
    elif kind == "high_pass":
        b0 = (1 + c) / 2
        b1 = -(1 + c)
        b2 = (1 + c) / 2
        a0 = 1 + alpha
        a1 = -2 * c
        a2 = 1 - alpha
    else:
        raise ValueError(kind)
    return np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0


def k_weighting(rate: float):
    """The two stages of the meter's default filter class, in the order they are applied."""
    return [biquad("high_shelf", 4.0, 1.0 / np.sqrt(2.0), 1500.0, rate), biquad("high_pass", 0.0, 0.5, 38.0, rate)]


def k_filter(x, rate: float) -> np.ndarray:
    y = np.asarray(x, dtype=np.float64)
    for b, a in k_weighting(rate):
        y = lfilter(b, a, y)
    return y


def block_bounds(n_samples: int, rate: float):
    """-> (l[], u[]) sample bounds of the gating blocks (u clipped to the signal, as numpy slicing clips)."""
    step = 1.0 - OVERLAP
    T = n_samples / rate
    n_blocks = int(np.round((T - BLOCK_S) / (BLOCK_S * step)) + 1)
    j = np.arange(n_blocks)
    l = np.array([int(BLOCK_S * (k * step) * rate) for k in j], dtype=np.int64)
    u = np.array([int(BLOCK_S * (k * step + 1) * rate) for k in j], dtype=np.int64)
    return l, np.minimum(u, n_samples)


def block_energies(x, rate: float) -> np.ndarray:
    if len(x) < BLOCK_S * rate:
        raise ValueError("Audio must have length greater than the block size.")
    y = k_filter(x, rate)
    l, u = block_bounds(len(y), rate)
    return np.array([(1.0 / (BLOCK_S * rate)) * np.sum(np.square(y[a:b])) for a, b in zip(l, u)])


def gate(z: np.ndarray) -> float:
    """Two-stage gating of the block energies of a mono signal -> LUFS (``-inf`` when every block is gated out)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        l = -0.691 + 10.0 * np.log10(z)
        keep = l >= GAMMA_ABS
        gamma_r = -0.691 + 10.0 * np.log10(np.mean(z[keep])) - 10.0 if keep.any() else np.nan
        keep = (l > gamma_r) & (l > GAMMA_ABS)
        z_avg = np.mean(z[keep]) if keep.any() else 0.0
        return float(-0.691 + 10.0 * np.log10(z_avg))


def integrated_loudness(x, rate: float) -> float:
    """``Meter(rate).integrated_loudness(x)`` for mono ``x``."""
    return gate(block_energies(x, rate))


def normalize_loudness(x, input_loudness: float, target_loudness: float) -> np.ndarray:
    """``pyloudnorm.normalize.loudness``: one gain for the whole signal."""
    gain = np.power(10.0, (target_loudness - input_loudness) / 20.0)
    return gain * np.asarray(x, dtype=np.float64)
