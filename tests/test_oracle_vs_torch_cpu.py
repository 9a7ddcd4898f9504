"""The C oracle against torch's own CPU operators on random shapes (hypothesis).

tests/golden pins the oracle to the reference on a handful of shapes; these properties pin each
building block to the ATen/MKL/oneDNN kernel the reference calls, bit for bit, over many shapes.
They hold for the CPU build the golden vectors were made with (AVX512 host, MKL sgemm); on another
CPU capability the reference itself would sum differently, so the module is skipped there.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from hypothesis import given, settings, strategies as st

import oracle

pytestmark = pytest.mark.skipif(torch.backends.cpu.get_cpu_capability() != "AVX512" or not torch.backends.mkl.is_available(),
                                reason="bit patterns are those of the AVX512 + MKL PyTorch CPU build")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


@settings(max_examples=25, deadline=None)
@given(st.integers(16, 300), st.sampled_from([64, 512]), st.integers(0, 2 ** 31 - 1), st.booleans())
def test_linear_matches_mkl_sgemm(M, N, seed, with_bias):
    """nn.Linear(512, N) for M >= 16 rows: K split 256 + 256, (bias + c0) + c1."""
    a, w = rnd(seed, M, 512), rnd(seed + 1, N, 512, scale=0.05)
    b = rnd(seed + 2, N) if with_bias else None
    want = F.linear(torch.from_numpy(a), torch.from_numpy(w), None if b is None else torch.from_numpy(b)).numpy()
    assert np.array_equal(bits(oracle.linear(a, w, b)), bits(want))


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 200), st.integers(0, 2 ** 31 - 1), st.floats(0.1, 10.0), st.floats(-3.0, 3.0))
def test_layernorm_matches_aten(M, seed, scale, shift):
    x = rnd(seed, M, 512, scale=scale) + np.float32(shift)
    g, b = rnd(seed + 1, 512) + 1, rnd(seed + 2, 512, scale=0.1)
    want = F.layer_norm(torch.from_numpy(x), (512,), torch.from_numpy(g), torch.from_numpy(b), 1e-5).numpy()
    assert np.array_equal(bits(oracle.layernorm(x, g, b, relu=False)), bits(want))


@settings(max_examples=20, deadline=None)
@given(st.integers(1, 4), st.integers(4, 300), st.integers(0, 2 ** 31 - 1))
def test_conv_matches_aten_dispatch(B, T, seed):
    """Both CPU back-ends of nn.Conv1d(80, 512, 4, 2, 1), chosen like ATen does, any T parity
    (a single output frame, T < 4 at B = 1, goes through MKL's gemv instead: outside the contract)."""
    x, w = rnd(seed, B, 80, T), rnd(seed + 1, 512, 80, 4, scale=0.05)
    want = F.conv1d(torch.from_numpy(x), torch.from_numpy(w), None, 2, 1).transpose(1, 2).contiguous().numpy()
    got = oracle.conv1d_k4s2(x, w)
    assert got.shape == want.shape == (B, (T - 2) // 2 + 1, 512)
    assert np.array_equal(bits(got), bits(want))


@settings(max_examples=20, deadline=None)
@given(st.integers(16, 400), st.sampled_from([64, 512, 1024]), st.integers(0, 2 ** 31 - 1), st.sampled_from([1.0 / 512, 0.3, 1.5]))
def test_vq_matches_addmm_argmin(N, M, seed, cb_scale):
    """VQEmbeddingEMA.encode (model.py:103-115): distances, first-index argmin, gather."""
    x = rnd(seed, N, 64, scale=0.4)
    E = ((np.random.default_rng(seed + 1).random((M, 64)) * 2 - 1) * cb_scale).astype(np.float32)
    xt, Et = torch.from_numpy(x), torch.from_numpy(E)
    d = torch.addmm(torch.sum(Et ** 2, dim=1) + torch.sum(xt ** 2, dim=1, keepdim=True), xt, Et.t(), alpha=-2.0, beta=1.0)
    idx = torch.argmin(d, dim=-1)
    q, got_idx, db, _ = oracle.vq_encode(x, E)
    assert np.array_equal(got_idx, idx.numpy())
    assert np.array_equal(bits(db), bits(d.min(dim=1).values.numpy()))
    assert np.array_equal(bits(q), bits(F.embedding(idx, Et).numpy()))


def test_linear_small_m_is_a_different_mkl_path():
    """Documented limit of the bit-exactness contract: below 16 rows MKL sums in another order."""
    a, w = rnd(5, 8, 512), rnd(6, 512, 512, scale=0.05)
    want = F.linear(torch.from_numpy(a), torch.from_numpy(w)).numpy()
    got = oracle.linear(a, w)
    assert np.abs(got - want).max() <= 1e-5 and not np.array_equal(bits(got), bits(want))
