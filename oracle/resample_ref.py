"""CPU restatement of the resampling the reference gets from ``librosa.load(path, sr=...)`` (``convert.py:54-56``) --
TEST INFRASTRUCTURE, like the rest of oracle/.

librosa ^0.8 resamples with ``res_type='kaiser_best'``, i.e. resampy's band-limited sinc interpolation (J. O. Smith,
"Digital Audio Resampling"): a half sinc window sampled at 2**precision points per zero crossing, tapered by a Kaiser
window, linearly interpolated between table entries; for down-sampling the filter is stretched and scaled by the rate
ratio.  resampy and librosa are not installed here and cannot be fetched: **parity unpinned**.  The filter
constants below (64 zero crossings, precision 9, Kaiser beta 14.769656459379492, roll-off 0.9475937167399596) are
resampy's published ``kaiser_best`` design, restated from its documentation -- they cannot be checked against the
package's stored table offline.  Everything in float64, as resampy computes on librosa's float32 input upcast.
Version assumption: the output clock follows resampy 0.2.2's interpolation loop (what librosa ^0.8 pins as >= 0.2.2), which
advances the input time by SEQUENTIAL additions `time_register += time_increment` -- not `t * time_increment` as resampy >= 0.3's
rewritten kernel does; the two differ in the last bits of the interpolation weights on long signals.
"""
import numpy as np

KAISER_BEST = {"num_zeros": 64, "precision": 9, "beta": 14.769656459379492, "rolloff": 0.9475937167399596}


def sinc_window(num_zeros, precision, beta, rolloff):
    """resampy.filters.sinc_window with a Kaiser taper: right half of the interpolation filter."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def resample(x, sr_orig, sr_new, filt=KAISER_BEST):
    """librosa.resample(x, sr_orig, sr_new, res_type='kaiser_best', fix=True, scale=False) for a mono signal."""
    x = np.asarray(x, dtype=np.float64)
    if sr_orig == sr_new:
        return x.copy()
    ratio = float(sr_new) / float(sr_orig)
    n_out = int(np.ceil(x.shape[-1] * ratio))                  # librosa's n_samples; resampy yields int(n * ratio), fix_length pads
    n_res = int(x.shape[-1] * ratio)
    win, num_table = sinc_window(**filt)
    win = win.copy()
    if ratio < 1:
        win *= ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * num_table)
    nwin = win.shape[0]
    n_orig = x.shape[0]
    y = np.zeros(n_out)
    time_register = 0.0
    for t in range(n_res):
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        if i_max > 0:
            idx = offset + index_step * np.arange(i_max)
            y[t] += np.dot(win[idx] + eta * delta[idx], x[n - np.arange(i_max)])
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        if k_max > 0:
            idx = offset + index_step * np.arange(k_max)
            y[t] += np.dot(win[idx] + eta * delta[idx], x[n + 1 + np.arange(k_max)])
        time_register += time_increment
    return y
