"""-m gpu: HIP mel front-end vs the float64 numpy oracle (librosa absent: self-oracle, parity unpinned).
Tolerance: 2e-4 absolute on the normalised log-mel (fp32 400-term DFT sums and fp32 log10 vs float64)."""
import numpy as np
import pytest
import torch

from oracle import mel_ref
from vectorquantizedcpc_amd import preprocess, synth

pytestmark = pytest.mark.gpu


def _wave(n, name):
    u = synth.uniform01("wave/" + name, n)
    t = np.arange(n) / 16000.0
    f0 = 110 + 40 * (len(name) % 5)
    return (0.3 * np.sin(2 * np.pi * f0 * t) + 0.12 * np.sin(2 * np.pi * 1850 * t + 1.0) + 0.06 * (u - 0.5)).astype(np.float32)


def test_single_utterance_matches_oracle():
    w = _wave(16000, "a")
    got = preprocess.wave_to_mel(w).cpu().numpy()                     # numpy in, like the reference's function
    want = mel_ref.wave_to_mel(w)
    assert got.shape == want.shape == (80, 101)
    assert np.abs(got - want).max() <= 2e-4, float(np.abs(got - want).max())
    assert np.abs(preprocess.wave_to_mel(torch.from_numpy(w * 0.25).cuda()).cpu().numpy() - got).max() <= 2e-4   # gain invariant


def test_ragged_batch_and_odd_lengths():
    lens = [4000, 2777, 801, 16000]
    waves = [_wave(n, "b%d" % i) for i, n in enumerate(lens)]
    batch = torch.zeros(len(lens), max(lens))
    for i, w in enumerate(waves):
        batch[i, : len(w)] = torch.from_numpy(w)
    got = preprocess.wave_to_mel(batch.cuda(), lengths=lens).cpu().numpy()
    assert got.shape == (4, 80, 101)
    for i, w in enumerate(waves):
        want = mel_ref.wave_to_mel(w)
        T = want.shape[1]
        assert T == 1 + lens[i] // 160
        assert np.abs(got[i, :, :T] - want).max() <= 2e-4, (i, float(np.abs(got[i, :, :T] - want).max()))
        assert not got[i, :, T:].any()


def test_front_end_feeds_the_encoder():
    import vectorquantizedcpc_amd as V
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
    enc.load_state_dict(synth.encoder_state_dict())
    enc = enc.cuda().eval()
    mel = preprocess.wave_to_mel(torch.from_numpy(_wave(32000, "c")).cuda())
    z, c, idx = enc.encode(mel[None])
    assert idx.shape == (1, (201 - 2) // 2 + 1)
