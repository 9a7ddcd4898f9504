"""The CPU oracle against the golden vectors made from the reference (tools/gen_golden.py).

Encoder / VQ half: BIT-EXACT (SHA-256 of the fp32 bit patterns of every stage, all code
indices, all argmin distances) for calls of >= 16 output frames -- the regime in which the
reference's MKL sgemm blocks K as restated in oracle/vqcpc_oracle.c.  The 8-row case is
kept to document the reference's own small-M path: indices still equal, activations within
1e-6.  Context `c` (LSTM) is compared at 1e-6, not bitwise.
"""
import hashlib
import os

import numpy as np
import pytest

import oracle
from vectorquantizedcpc_amd import synth

CASES = ["c1_init", "c2_init", "c2_random_data", "ragged_3x32", "tiny_1x16", "odd_2x33", "long_1x300",
         "edge_1x32", "edge_1x34", "edge_1x62", "edge_2x16"]      # edge_*: 16..31 rows, the edge of the bit-exact contract
_cache = {}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(name, golden_dir):
    if name not in _cache:
        g = np.load(os.path.join(golden_dir, f"encoder_{name}.npz"))
        B, T = (int(v) for v in g["case"])
        sd = synth.encoder_state_dict(ln_affine=str(g["ln_affine"]), codebook=str(g["codebook"]))
        mel = synth.mel(name, B, T).numpy()
        _cache[name] = (g, sd, mel, oracle.encoder_encode(sd, mel))
    return _cache[name]


def spread_rows(a):
    a2 = a.reshape(-1, a.shape[-1])
    return a2[:: max(1, a2.shape[0] // 4)][:4]


@pytest.mark.parametrize("name", CASES)
def test_indices_bit_exact(name, golden_dir):
    g, _, _, r = run_case(name, golden_dir)
    assert np.array_equal(r["indices"], g["indices"].astype(np.int64))
    assert r["indices"].dtype == np.int64


@pytest.mark.parametrize("name", [c for c in CASES if c != "tiny_1x16"])
def test_activations_bit_exact(name, golden_dir):
    g, sd, mel, r = run_case(name, golden_dir)
    conv = oracle.conv1d_k4s2(mel, sd["conv.weight"].numpy())
    assert sha(conv) == str(g["sha_conv"])
    assert sha(r["z_pre"]) == str(g["sha_z_pre"])
    assert sha(r["z"]) == str(g["sha_z"])
    assert np.array_equal(r["d_best"], g["d_best"])
    assert np.array_equal(r["d_second"], g["d_second"])


def test_stage_chain_bit_exact(golden_dir):
    """Every intermediate of the seg-FC stack (model.py:46-55) for one case."""
    g, sd, mel, _ = run_case("ragged_3x32", golden_dir)
    x = oracle.conv1d_k4s2(mel, sd["conv.weight"].numpy())
    x = x.reshape(-1, 512)
    pre = oracle.layernorm(x, sd["encoder.0.weight"], sd["encoder.0.bias"], relu=False)
    assert sha(pre) == str(g["sha_enc0"])
    x = np.maximum(pre, 0)
    for lin, ln in ((2, 3), (5, 6), (8, 9), (11, 12)):
        y = oracle.linear(x, sd[f"encoder.{lin}.weight"].numpy())
        assert sha(y) == str(g[f"sha_enc{lin}"])
        pre = oracle.layernorm(y, sd[f"encoder.{ln}.weight"], sd[f"encoder.{ln}.bias"], relu=False)
        assert sha(pre) == str(g[f"sha_enc{ln}"])     # hook fires before the in-place ReLU
        x = np.maximum(pre, 0)


def test_small_m_case_documented(golden_dir):
    g, _, _, r = run_case("tiny_1x16", golden_dir)
    assert np.abs(spread_rows(r["z_pre"]) - g["rows_z_pre"]).max() <= 1e-6


@pytest.mark.parametrize("name", CASES)
def test_context_and_forward_stats(name, golden_dir):
    g, sd, _, r = run_case(name, golden_dir)
    assert np.abs(spread_rows(r["c"]) - g["rows_c"]).max() <= 1e-6
    zst, loss, ppl = oracle.vq_forward_stats(r["z_pre"], r["z"], r["indices"], 512)
    assert abs(loss - g["loss"]) <= 1e-6 * max(1.0, abs(g["loss"]))
    assert abs(ppl - g["perplexity"]) <= 1e-4 * g["perplexity"]
    assert np.abs(spread_rows(zst) - g["rows_z_fwd"]).max() <= 1e-6


def test_conv_dispatch_rule():
    """ATen's use_mkldnn rule decides the summation order (see vqcpc_oracle.c)."""
    L = oracle.lib()
    assert L.orc_conv_mode(1, 80, 200) == 1 and L.orc_conv_mode(1, 80, 256) == 1
    assert L.orc_conv_mode(1, 80, 258) == 2 and L.orc_conv_mode(2, 80, 16) == 2


def test_vq_first_index_wins_ties():
    """torch.argmin semantics (model.py:112): identical codebook rows -> lowest index."""
    E = np.tile(synth.uniform01("tie", 64).astype(np.float32)[None], (8, 1))
    x = synth.uniform01("tiex", 5 * 64).astype(np.float32).reshape(5, 64)
    _, idx, db, ds = oracle.vq_encode(x, E)
    assert (idx == 0).all() and np.array_equal(db, ds)


def test_philox_known_answer():
    """Random123 known-answer vectors for Philox4x32-10."""
    assert oracle.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert oracle.philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert oracle.philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)
    ctr = np.array([[0, 0, 0, 0], [0xFFFFFFFF] * 4], np.uint32)
    assert tuple(int(v) for v in synth.philox4x32_10(ctr[:1], (0, 0))[0]) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)


def test_sampling_noise_is_finite_for_every_word():
    """The protocol's uniform, ((w >> 9) + 0.5) * 2^-23, is exact in fp32 and strictly inside (0, 1): the
    extreme Philox words give finite Gumbel noise (a 24-bit form rounds to 1.0f at the top word -> +inf,
    which would win the argmax whatever the logits are).  C oracle and torch port use the same conversion."""
    from oracle import torch_ref
    words = np.array([0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFE00, 0xFFFFFFFF], np.uint32)
    c = np.array([oracle.noise_from_word(int(w)) for w in words], np.float32)
    t = torch_ref.noise_from_words(words).numpy()
    assert np.isfinite(c).all() and np.isfinite(t).all()
    assert np.abs(c - t).max() <= 2e-6                       # libm vs ATen logf
    u_max = (np.float32(0xFFFFFFFF >> 9) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    u_min = np.float32(0.5) * np.float32(1.0 / 8388608.0)
    assert float(u_max) == 1.0 - 2.0 ** -24 and float(u_min) == 2.0 ** -24
    assert c[0] < -2.5 and 16.0 < c[-1] < 17.0               # -log(-log(2^-24)) = -2.81, -log(-log(1 - 2^-24)) = 16.6
    assert np.all(np.diff(c) >= 0)                           # monotone in the word


def test_preprocess_functions_match_the_reference(golden_dir):
    """preprocess.npz = outputs of the reference's OWN preemphasis / mulaw_encode / mulaw_decode (preprocess.py:16-35,
    tools/gen_golden.py::preprocess_fixture_from_reference): the oracle's mu-law table, the package's numpy drop-ins and the
    oracle's pre-emphasis stage (first stage of wave_to_mel) are checked against them, not against a restated formula."""
    from oracle import mel_ref
    from vectorquantizedcpc_amd import preprocess as pp
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    assert "reference preprocess.py" in str(g["source"])
    dec = g["mulaw_decode_256"]
    assert dec.dtype == np.float64 and dec.shape == (256,)
    got = np.array([oracle.mulaw_decode(int(v)) for v in range(256)])
    assert np.array_equal(got, dec.astype(np.float32))             # the sample loop emits the fp32 rounding of the reference's fp64
    y = 2.0 * np.arange(256) / 255.0 - 1.0
    assert np.array_equal(pp.mulaw_decode(y, 256), dec)
    assert np.array_equal(pp.mulaw_encode(g["mulaw_encode_in"], 256), g["mulaw_encode_out"])
    assert np.array_equal(pp.mulaw_encode(g["mulaw_encode_in"].astype(np.float32), 256), g["mulaw_encode_out_f32_in"])
    assert g["mulaw_encode_out"].min() == 0 and g["mulaw_encode_out"].max() == 255
    pre = mel_ref.preemphasis(g["preemph_in"].astype(np.float64), 0.97)
    assert np.abs(pre - g["preemph_out"]).max() <= 1e-12           # scipy.signal.lfilter([1, -a], [1]) in float64


def test_mulaw_decode_matches_reference_formula():
    """preprocess.py:30-35 evaluated in float64 numpy, as the reference does."""
    s = np.arange(256)
    y = 2.0 * s / 255.0 - 1.0
    ref = np.sign(y) / 255.0 * ((1 + 255.0) ** np.abs(y) - 1)
    got = np.array([oracle.mulaw_decode(int(v)) for v in s])
    assert np.array_equal(got, ref.astype(np.float32))
    assert got[0] == -1.0 and got[255] == 1.0


def test_vocoder_glue_matches_fixture(golden_dir):
    """network_vocoder.py:69-77 layout: [:64] code emb twice per code, [64:] speaker emb.  The fixture is what the
    REFERENCE's own Vocoder.generate / Vocoder.forward handed to a capture stub standing in for rnnms
    (tools/gen_golden.py::glue_fixture_from_reference)."""
    g = np.load(os.path.join(golden_dir, "vocoder_glue.npz"))
    assert "reference network_vocoder.py" in str(g["source"])
    sd = synth.vocoder_state_dict()
    import ctypes as C
    z, spk = g["z"], g["speaker"]
    out = np.empty(g["series"].shape, np.float32)
    ce = sd["code_embedding.weight"].numpy()
    se = sd["speaker_embedding.weight"].numpy()
    oracle.lib().orc_vocoder_glue(z.ctypes.data_as(C.c_void_p), spk.ctypes.data_as(C.c_void_p), z.shape[0], z.shape[1],
                                  ce.ctypes.data_as(C.c_void_p), 64, se.ctypes.data_as(C.c_void_p), 64,
                                  out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, g["series"])


def test_vocoder_selforacle_regression(golden_dir):
    """Self-oracle (parity unpinned): the committed vectors pin the spec against drift."""
    g = np.load(os.path.join(golden_dir, "vocoder_selforacle.npz"))
    sd = synth.vocoder_state_dict()
    r = oracle.vocoder_generate(sd, g["z1"], int(g["spk1"]), seed=synth.SEED, utterance=1, n_steps=640, want_logits=True)
    assert np.array_equal(r["samples"], g["samples1"].astype(np.int64))
    assert np.abs(r["logits"][::64] - g["logits1"]).max() <= 1e-6
    assert np.array_equal(r["wav"], g["wav1"])
