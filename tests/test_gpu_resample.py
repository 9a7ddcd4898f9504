"""-m gpu: the HIP resampler (vqcpc_resampler_*, preprocess.resample) against the float64 oracle.
PARITY UNPINNED (resampy / librosa absent): oracle/resample_ref.py restates the published kaiser_best algorithm.
Tolerance: the kernel computes the output time as t * increment, resampy accumulates it -- the two differ by rounding
only, so 2e-6 absolute on signals of unit amplitude (fp32 output)."""
import numpy as np
import pytest
import torch

from oracle import resample_ref as R
from vectorquantizedcpc_amd import io, preprocess, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sr_in,sr_out", [(44100, 16000), (48000, 16000), (22050, 16000), (8000, 16000), (16000, 22050)])
def test_resample_matches_oracle(sr_in, sr_out):
    n = 3000
    x = (synth.uniform01(f"rs/{sr_in}", n) * 2 - 1).astype(np.float32)
    x = np.convolve(x, np.ones(5) / 5, mode="same").astype(np.float32)
    got = preprocess.resample(torch.from_numpy(x).cuda(), sr_in, sr_out).cpu().numpy()
    want = R.resample(x, sr_in, sr_out)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.abs(got - want).max() <= 2e-6


def test_ragged_batch_and_load_wav(tmp_path):
    lens = [2000, 777, 1]
    x = np.zeros((3, 2000), np.float32)
    for b, n in enumerate(lens):
        x[b, :n] = (synth.uniform01(f"rsb/{b}", n) - 0.5).astype(np.float32)
    got = preprocess.resample(torch.from_numpy(x).cuda(), 44100, 16000, lengths=lens).cpu().numpy()
    assert got.shape == (3, int(np.ceil(2000 * 16000 / 44100)))
    for b, n in enumerate(lens):
        want = R.resample(x[b, :n], 44100, 16000)
        assert np.abs(got[b, : len(want)] - want).max() <= 2e-6 and not got[b, len(want):].any()
    from scipy.io import wavfile
    wavfile.write(tmp_path / "a.wav", 22050, x[0])
    w = io.load_wav(tmp_path / "a", 16000)
    assert w.shape == (int(np.ceil(2000 * 16000 / 22050)),) and not w.is_cuda
    assert np.abs(w.numpy() - R.resample(x[0], 22050, 16000)).max() <= 2e-6
    wavfile.write(tmp_path / "b.wav", 16000, x[0])
    assert torch.equal(io.load_wav(tmp_path / "b", 16000), torch.from_numpy(x[0]))
