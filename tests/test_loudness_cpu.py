"""The loudness oracle (oracle/loudness_ref.py) against what the standard itself publishes.

pyloudnorm is absent (parity unpinned against it); ITU-R BS.1770 gives a conformance point and a coefficient table,
and the gating has properties that hold whatever the filter: these anchor the restatement."""
import numpy as np
import pytest

from oracle import loudness_ref as L


def _sine(rate, f=997.0, dur=5.0, amp=1.0):
    return amp * np.sin(2 * np.pi * f * np.arange(int(rate * dur)) / rate)


@pytest.mark.parametrize("rate", [48000, 44100, 16000])
def test_bs1770_conformance_sine(rate):
    """0 dBFS 997 Hz sine on one front channel: -3.01 LKFS, conformance tolerance 0.1 LU."""
    assert abs(L.integrated_loudness(_sine(rate), rate) - (-3.01)) < 0.1
    assert abs(L.integrated_loudness(_sine(rate, amp=0.1), rate) - L.integrated_loudness(_sine(rate), rate) + 20.0) < 1e-9


def test_shelf_coefficients_at_48k_match_the_bs1770_table():
    (b, a), (b2, a2) = L.k_weighting(48000)
    assert np.allclose(b, [1.53512485958697, -2.69169618940638, 1.19839281085285], atol=2e-4)
    assert np.allclose(a, [1.0, -1.69065929318241, 0.73248077421585], atol=2e-4)
    # high pass: same poles as the table; the numerator is normalised to unit passband gain
    assert np.allclose(a2, [1.0, -1.99004745483398, 0.99007225036621], atol=1e-4)
    assert np.allclose(b2 / b2[0], [1.0, -2.0, 1.0], atol=1e-12)


def test_block_bounds_and_short_audio():
    l, u = L.block_bounds(32000, 16000)
    assert len(l) == 17 and l[1] == 1600 and u[0] == 6400 and u[-1] == 32000
    l, u = L.block_bounds(32960, 16000)                       # rounds up to a block that runs past the end: clipped
    assert len(l) == 18 and u[-1] == 32960 and l[-1] == 27200
    assert len(L.block_bounds(6400, 16000)[0]) == 1
    with pytest.raises(ValueError):
        L.integrated_loudness(np.zeros(6399), 16000)


def test_gating():
    assert L.integrated_loudness(np.zeros(16000), 16000) == -np.inf           # every block under the absolute gate
    rate = 16000
    loud, quiet = _sine(rate, dur=3.0, amp=0.5), _sine(rate, dur=3.0, amp=0.5e-2)     # 40 dB apart: relative gate drops the tail
    both = L.integrated_loudness(np.concatenate([loud, quiet]), rate)
    assert abs(both - L.integrated_loudness(loud, rate)) < 0.5        # an ungated mean would sit 3 dB lower
    x = _sine(rate, dur=2.0, amp=0.3)
    y = L.normalize_loudness(x, L.integrated_loudness(x, rate), -30.0)
    assert abs(L.integrated_loudness(y, rate) + 30.0) < 1e-9
