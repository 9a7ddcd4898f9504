"""ctypes binding of ``oracle/vqcpc_oracle.c`` (numpy in, numpy out)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvqcpc_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with ``oracle/Makefile`` (gcc only)."""
    src = os.path.join(_HERE, "vqcpc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libvqcpc_oracle.so"])
    return _SO


class _EncW(C.Structure):
    _fields_ = [("conv_w", C.c_void_p), ("ln_g", C.c_void_p * 5), ("ln_b", C.c_void_p * 5),
                ("fc_w", C.c_void_p * 4), ("out_w", C.c_void_p), ("out_b", C.c_void_p),
                ("codebook", C.c_void_p), ("w_ih", C.c_void_p), ("w_hh", C.c_void_p),
                ("b_ih", C.c_void_p), ("b_hh", C.c_void_p),
                ("in_channels", C.c_int), ("channels", C.c_int), ("n_embeddings", C.c_int),
                ("z_dim", C.c_int), ("c_dim", C.c_int)]


class _VocW(C.Structure):
    _fields_ = [("code_emb", C.c_void_p), ("spk_emb", C.c_void_p),
                ("p_wih", (C.c_void_p * 2) * 2), ("p_whh", (C.c_void_p * 2) * 2),
                ("p_bih", (C.c_void_p * 2) * 2), ("p_bhh", (C.c_void_p * 2) * 2),
                ("ar_emb", C.c_void_p), ("ar_wih", C.c_void_p), ("ar_whh", C.c_void_p),
                ("ar_bih", C.c_void_p), ("ar_bhh", C.c_void_p),
                ("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p),
                ("n_codes", C.c_int), ("dz", C.c_int), ("n_spk", C.c_int), ("ds", C.c_int),
                ("Hp", C.c_int), ("dl", C.c_int), ("de", C.c_int), ("Hr", C.c_int), ("Hf", C.c_int),
                ("n_cls", C.c_int), ("upsample", C.c_int), ("bits", C.c_int)]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_sumsq64.restype = C.c_float
        _lib.orc_sample_noise.restype = C.c_float
        _lib.orc_noise_from_word.restype = C.c_float
        _lib.orc_noise_from_word.argtypes = [C.c_uint32]
        _lib.orc_sample_noise.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.orc_sample_from_logits.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib.orc_mulaw_decode.restype = C.c_float
        _lib.orc_mulaw_decode.argtypes = [C.c_int, C.c_int]
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _np(t):
    return _f32(t.detach().cpu().numpy() if hasattr(t, "detach") else t)


# --------------------------------------------------------------------------- encoder half
def conv1d_k4s2(x, w, mode=0):
    """model.py:43/:65 -- (B,C,T),(O,C,4) -> rows (B, T/2, O).  mode: see vqcpc_oracle.c."""
    x, w = _f32(x), _f32(w)
    B, Cc, T = x.shape
    O = w.shape[0]
    y = np.empty((B, (T - 2) // 2 + 1, O), np.float32)
    rc = lib().orc_conv1d_k4s2(_p(x), B, Cc, T, _p(w), O, int(mode), _p(y))
    assert rc == 0
    return y


def layernorm(x, g, b, relu=True, eps=1e-5):
    x, g, b = _f32(x), _f32(g), _f32(b)
    N = x.shape[-1]
    y = np.empty_like(x)
    rc = lib().orc_layernorm(_p(x), x.size // N, N, _p(g), _p(b), C.c_float(eps), int(relu), _p(y))
    assert rc == 0
    return y


def linear(a, w, bias=None):
    a, w = _f32(a), _f32(w)
    K = a.shape[-1]
    N = w.shape[0]
    bias = None if bias is None else _f32(bias)
    y = np.empty(a.shape[:-1] + (N,), np.float32)
    rc = lib().orc_linear(_p(a), a.size // K, K, _p(w), N, _p(bias), _p(y))
    assert rc == 0, "unsupported K for the MKL blocking rule"
    return y


def sumsq64(v):
    v = _f32(v)
    assert v.shape == (64,)
    return np.float32(lib().orc_sumsq64(_p(v)))


def vq_encode(x, E):
    """model.py:103-115 -> (q, idx, d_best, d_second)."""
    x, E = _f32(x), _f32(E)
    D = x.shape[-1]
    N = x.size // D
    idx = np.empty(N, np.int64)
    q = np.empty((N, D), np.float32)
    db = np.empty(N, np.float32)
    ds = np.empty(N, np.float32)
    rc = lib().orc_vq_encode(_p(x), N, D, _p(E), E.shape[0], _p(idx), _p(q), _p(db), _p(ds))
    assert rc == 0
    return q.reshape(x.shape), idx.reshape(x.shape[:-1]), db, ds


def vq_forward_stats(x, q, idx, M):
    """model.py:147-153 -> (z_st, loss, perplexity)."""
    x, q, idx = _f32(x), _f32(q), _i64(idx)
    D = x.shape[-1]
    N = x.size // D
    zst = np.empty_like(x)
    loss = C.c_float()
    ppl = C.c_float()
    lib().orc_vq_forward_stats(_p(x), _p(q), _p(idx), N, D, M, _p(zst), C.byref(loss), C.byref(ppl))
    return zst, np.float32(loss.value), np.float32(ppl.value)


def lstm(x, wih, whh, bih, bhh):
    x = _f32(x)
    B, T, D = x.shape
    H = whh.shape[1]
    out = np.empty((B, T, H), np.float32)
    lib().orc_lstm(_p(x), B, T, D, H, _p(_f32(wih)), _p(_f32(whh)), _p(_f32(bih)), _p(_f32(bhh)), _p(out))
    return out


def _enc_struct(sd):
    keep = []

    def g(k):
        a = _np(sd[k])
        keep.append(a)
        return a.ctypes.data

    w = _EncW()
    w.conv_w = g("conv.weight")
    for i, n in enumerate((0, 3, 6, 9, 12)):
        w.ln_g[i] = g(f"encoder.{n}.weight")
        w.ln_b[i] = g(f"encoder.{n}.bias")
    for i, n in enumerate((2, 5, 8, 11)):
        w.fc_w[i] = g(f"encoder.{n}.weight")
    w.out_w, w.out_b = g("encoder.14.weight"), g("encoder.14.bias")
    w.codebook = g("codebook.embedding")
    w.w_ih, w.w_hh = g("rnn.weight_ih_l0"), g("rnn.weight_hh_l0")
    w.b_ih, w.b_hh = g("rnn.bias_ih_l0"), g("rnn.bias_hh_l0")
    cw = _np(sd["conv.weight"])
    w.in_channels, w.channels = cw.shape[1], cw.shape[0]
    cb = _np(sd["codebook.embedding"])
    w.n_embeddings, w.z_dim = cb.shape
    w.c_dim = _np(sd["rnn.weight_hh_l0"]).shape[1]
    return w, keep


def encoder_encode(sd, mel, want_c=True, conv_mode=0):
    """Encoder.encode (model.py:59-70) from a state_dict -> dict of numpy outputs."""
    mel = _f32(mel)
    B, _, T = mel.shape
    w, keep = _enc_struct(sd)
    To = (T - 2) // 2 + 1
    N = B * To
    zp = np.empty((B, To, w.z_dim), np.float32)
    zq = np.empty_like(zp)
    idx = np.empty((B, To), np.int64)
    c = np.empty((B, To, w.c_dim), np.float32) if want_c else None
    db = np.empty(N, np.float32)
    ds = np.empty(N, np.float32)
    rc = lib().orc_encoder_encode(C.byref(w), _p(mel), B, T, int(conv_mode), _p(zp), _p(zq), _p(idx), _p(c), _p(db), _p(ds))
    assert rc == 0
    del keep
    return {"z_pre": zp, "z": zq, "indices": idx, "c": c, "d_best": db, "d_second": ds}


# --------------------------------------------------------------------------- vocoder half
def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def sample_noise(seed, utterance, t, k):
    """Gumbel noise of class k, sample t, utterance `utterance` (see orc_sample_noise)."""
    return np.float32(lib().orc_sample_noise(int(seed), int(utterance), int(t), int(k)))


def noise_from_word(w):
    """Gumbel noise of one raw Philox word (the uniform conversion of the sampling protocol)."""
    return np.float32(lib().orc_noise_from_word(int(w) & 0xFFFFFFFF))


def mulaw_decode(s, bits=8):
    return np.float32(lib().orc_mulaw_decode(int(s), int(bits)))


def sample_from_logits(logits, seed, utterance, t):
    """Gumbel-max draw -> (class, scores[n])."""
    lg = _f32(logits)
    sc = np.empty(lg.size, np.float32)
    k = lib().orc_sample_from_logits(_p(lg), lg.size, int(seed), int(utterance), int(t), _p(sc))
    return int(k), sc


def _voc_struct(sd, upsample=160, bits=8):
    keep = []

    def g(k):
        a = _np(sd[k])
        keep.append(a)
        return a.ctypes.data

    w = _VocW()
    w.code_emb, w.spk_emb = g("code_embedding.weight"), g("speaker_embedding.weight")
    for layer in range(2):
        for d, suf in enumerate(("", "_reverse")):
            w.p_wih[layer][d] = g(f"rnnms.prenet.weight_ih_l{layer}{suf}")
            w.p_whh[layer][d] = g(f"rnnms.prenet.weight_hh_l{layer}{suf}")
            w.p_bih[layer][d] = g(f"rnnms.prenet.bias_ih_l{layer}{suf}")
            w.p_bhh[layer][d] = g(f"rnnms.prenet.bias_hh_l{layer}{suf}")
    w.ar_emb = g("rnnms.ar.embedding.weight")
    w.ar_wih, w.ar_whh = g("rnnms.ar.rnn.weight_ih_l0"), g("rnnms.ar.rnn.weight_hh_l0")
    w.ar_bih, w.ar_bhh = g("rnnms.ar.rnn.bias_ih_l0"), g("rnnms.ar.rnn.bias_hh_l0")
    w.fc1_w, w.fc1_b = g("rnnms.ar.fc1.weight"), g("rnnms.ar.fc1.bias")
    w.fc2_w, w.fc2_b = g("rnnms.ar.fc2.weight"), g("rnnms.ar.fc2.bias")
    ce, se = _np(sd["code_embedding.weight"]), _np(sd["speaker_embedding.weight"])
    w.n_codes, w.dz = ce.shape
    w.n_spk, w.ds = se.shape
    w.Hp = _np(sd["rnnms.prenet.weight_hh_l0"]).shape[1]
    w.dl = 2 * w.Hp
    w.n_cls, w.de = _np(sd["rnnms.ar.embedding.weight"]).shape
    w.Hr = _np(sd["rnnms.ar.rnn.weight_hh_l0"]).shape[1]
    w.Hf = _np(sd["rnnms.ar.fc1.weight"]).shape[0]
    w.upsample, w.bits = upsample, bits
    return w, keep


def vocoder_condition(sd, z, speaker, upsample=160):
    """Prenet output for one utterance: (2*Tc, dl)."""
    z = _i64(z)
    w, keep = _voc_struct(sd, upsample)
    out = np.empty((2 * z.size, w.dl), np.float32)
    lib().orc_vocoder_condition(C.byref(w), _p(z), C.c_int64(int(speaker)), z.size, _p(out))
    del keep
    return out


def vocoder_generate(sd, z, speaker, seed, utterance=0, n_steps=None, inputs=None,
                     want_logits=False, upsample=160, bits=8):
    """Vocoder.generate for ONE utterance (project spec) -> dict(samples, wav, logits).

    ``inputs``: teacher forcing -- the previous-sample fed at step t (Vocoder.forward semantics)."""
    z = _i64(z)
    w, keep = _voc_struct(sd, upsample, bits)
    total = upsample * 2 * z.size
    n_steps = total if n_steps is None else min(int(n_steps), total)
    samples = np.empty(n_steps, np.int64)
    wav = np.empty(n_steps, np.float32)
    logits = np.empty((n_steps, w.n_cls), np.float32) if want_logits else None
    inputs = None if inputs is None else _i64(inputs)
    if inputs is not None:
        assert inputs.size >= n_steps
    rc = lib().orc_vocoder_generate(C.byref(w), _p(z), C.c_int64(int(speaker)), z.size,
                                    C.c_uint64(int(seed)), C.c_uint32(int(utterance)), n_steps,
                                    _p(inputs), _p(samples), _p(wav), _p(logits))
    assert rc == 0
    del keep
    return {"samples": samples, "wav": wav, "logits": logits}
