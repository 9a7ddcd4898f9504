"""-m gpu: HIP loudness meter / normaliser vs the float64 oracle (pyloudnorm absent: self-oracle, parity unpinned).

Tolerances: block energies 1e-10 relative and LUFS 1e-9 dB (both sides fp64; the chunked scan and the
summation order differ from scipy's serial filter only in rounding); normalised samples within 1 fp32 ulp
(the device's fp64 pow may differ from numpy's in the last place before the product is rounded to fp32)."""
import numpy as np
import pytest
import torch

import vectorquantizedcpc_amd.loudness as pyloudnorm            # the drop-in name convert.py uses
from oracle import loudness_ref as L
from vectorquantizedcpc_amd import synth

pytestmark = pytest.mark.gpu


def _speechlike(n, name, amp=0.3):
    u = synth.uniform01("loud/" + name, n)
    t = np.arange(n) / 16000.0
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 1.7 * t + len(name))
    return (amp * env * (np.sin(2 * np.pi * 180 * t) + 0.4 * np.sin(2 * np.pi * 2310 * t + 0.3)) + 0.05 * (u - 0.5)).astype(np.float32)


def test_single_utterance_like_convert_py():
    wav = _speechlike(32000, "a")
    meter = pyloudnorm.Meter(16000)                               # convert.py:50
    got = meter.integrated_loudness(wav)                          # numpy in, float out (convert.py:57)
    want = L.integrated_loudness(wav, 16000)
    assert isinstance(got, float) and abs(got - want) < 1e-9, (got, want)
    out = pyloudnorm.normalize.loudness(torch.from_numpy(wav).cuda(), got, -27.5)      # convert.py:80
    ref = L.normalize_loudness(wav, want, -27.5).astype(np.float32)
    assert np.abs(out.cpu().numpy() - ref).max() <= np.spacing(np.abs(ref).max())
    assert abs(L.integrated_loudness(out.cpu().numpy(), 16000) + 27.5) < 1e-5


def test_block_energies_ragged_batch():
    lens = [6400, 6401, 32000, 32960, 20017, 160000]               # one block; odd; 2 s; block past the end; ragged; 10 s
    waves = [_speechlike(n, "b%d" % i, amp=0.05 + 0.1 * i) for i, n in enumerate(lens)]
    batch = torch.zeros(len(lens), max(lens))
    for i, w in enumerate(waves):
        batch[i, : len(w)] = torch.from_numpy(w)
    lufs, z = pyloudnorm.Meter(16000).integrated_loudness(batch.cuda(), lengths=lens, return_blocks=True)
    assert lufs.dtype == torch.float64 and lufs.shape == (len(lens),)
    for i, w in enumerate(waves):
        zw = L.block_energies(w, 16000)
        assert z[i].shape == zw.shape
        assert np.abs(z[i].cpu().numpy() / zw - 1).max() < 1e-10, i
        assert abs(lufs[i].item() - L.gate(zw)) < 1e-9, i


def test_gates_and_refusals():
    meter = pyloudnorm.Meter(16000)
    assert meter.integrated_loudness(torch.zeros(16000).cuda()) == -np.inf       # everything under the absolute gate
    loud, quiet = _speechlike(48000, "c", 0.4), _speechlike(48000, "d", 0.4) * 1e-2
    both = np.concatenate([loud, quiet]).astype(np.float32)
    assert abs(meter.integrated_loudness(both) - L.integrated_loudness(both, 16000)) < 1e-9     # relative gate in play
    with pytest.raises(ValueError):
        meter.integrated_loudness(torch.zeros(6399).cuda())                       # shorter than one block
    with pytest.raises(ValueError):
        meter.integrated_loudness(torch.zeros(2, 8000).cuda(), lengths=[8000, 100])
    for rate in (48000, 44100):                                                    # coefficients derived per rate
        x = np.sin(2 * np.pi * 997 * np.arange(rate * 2) / rate).astype(np.float32)
        assert abs(pyloudnorm.Meter(rate).integrated_loudness(x) - L.integrated_loudness(x, rate)) < 1e-9


def test_match_loudness_batch():
    lens = [9000, 16000, 12345]
    wavs = [torch.from_numpy(_speechlike(n, "e%d" % i, 0.2)).cuda() for i, n in enumerate(lens)]
    target = [-23.0, -31.0, -18.5]
    out = pyloudnorm.match_loudness(wavs, target)
    for w, o, t in zip(wavs, out, target):
        assert o.shape == w.shape
        assert abs(L.integrated_loudness(o.cpu().numpy(), 16000) - t) < 1e-5
