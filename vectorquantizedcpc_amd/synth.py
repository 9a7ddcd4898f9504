"""Seeded synthetic weights and inputs for the VQ-CPC inference path.

The reference ships neither checkpoints nor data (``.gitignore:110-115`` of the
reference), so tests and ``bench.py`` use random-init weights of the reference's
architecture and synthetic mel/code inputs.  Everything comes from ONE
counter-based generator (Philox4x32-10, seed 13 = the reference's global seed,
``config.py:13``) implemented here in numpy, so the same tensors are rebuilt
bit-for-bit anywhere from just ``(seed, name)`` -- fixtures carry no weights.

Distributions follow the reference initialisers: Conv/Linear
U(+-1/sqrt(fan_in)), LayerNorm 1/0 (or a perturbed affine for stricter tests),
codebook U(+-1/512) (``model.py:96-98``), LSTM/GRU U(+-1/sqrt(H)), embeddings a
12-uniform Irwin-Hall stand-in for N(0,1) (exact arithmetic, no libm).
"""
import zlib

import numpy as np
import torch

SEED = 13

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr: np.ndarray, key) -> np.ndarray:
    """Vectorised Philox4x32-10: ctr (N,4) uint32, key (k0,k1) -> (N,4) uint32."""
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & _MASK
        n3 = p0 & _MASK
        c = [n0, n1, n2, n3]
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack(c, axis=1).astype(np.uint32)


def uniform01(name: str, n: int, seed: int = SEED) -> np.ndarray:
    """n float64 uniforms in [0,1) with 24 random bits each, stream keyed by name."""
    stream = zlib.crc32(name.encode()) & 0xFFFFFFFF
    nblk = (n + 3) // 4
    ctr = np.zeros((nblk, 4), np.uint32)
    ctr[:, 0] = np.arange(nblk, dtype=np.uint64) & 0xFFFFFFFF
    ctr[:, 1] = (np.arange(nblk, dtype=np.uint64) >> 32).astype(np.uint32)
    ctr[:, 2] = stream
    words = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).reshape(-1)[:n]
    return (words >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)


def _uniform(name, shape, bound, seed):
    n = int(np.prod(shape))
    v = (uniform01(name, n, seed) * 2.0 - 1.0) * bound
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def _normalish(name, shape, seed):
    n = int(np.prod(shape))
    v = uniform01(name, 12 * n, seed).reshape(n, 12).sum(axis=1) - 6.0
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def randint(name: str, shape, high: int, seed: int = SEED) -> torch.Tensor:
    n = int(np.prod(shape))
    v = np.floor(uniform01(name, n, seed) * high).astype(np.int64)
    return torch.from_numpy(np.minimum(v, high - 1).reshape(shape))


def mel(name: str, B: int, T: int, n_mels: int = 80, seed: int = SEED) -> torch.Tensor:
    """Synthetic log-mel batch (B, n_mels, T), U(0,1) = the range ``preprocess.py:73-74`` yields."""
    return torch.from_numpy(uniform01("mel/" + name, B * n_mels * T, seed)
                            .astype(np.float32).reshape(B, n_mels, T))


def encoder_state_dict(seed: int = SEED, in_channels=80, channels=512, n_embeddings=512,
                       z_dim=64, c_dim=256, ln_affine: str = "init", codebook: str = "init"):
    """Random-init ``Encoder.state_dict()`` (keys and shapes of ``model.py:43-57``).

    ln_affine: "init" (gamma=1, beta=0) or "random" (gamma U(0.5,1.5), beta U(+-0.1)).
    codebook:  "init" U(+-1/512) (``model.py:96-98``) or "data" U(+-1.5) (data-scale regime).
    """
    sd = {}
    sd["conv.weight"] = _uniform("enc/conv.weight", (channels, in_channels, 4), (in_channels * 4) ** -0.5, seed)
    for n in (0, 3, 6, 9, 12):
        if ln_affine == "random":
            sd[f"encoder.{n}.weight"] = _uniform(f"enc/ln{n}.g", (channels,), 0.5, seed) + 1.0
            sd[f"encoder.{n}.bias"] = _uniform(f"enc/ln{n}.b", (channels,), 0.1, seed)
        else:
            sd[f"encoder.{n}.weight"] = torch.ones(channels)
            sd[f"encoder.{n}.bias"] = torch.zeros(channels)
    for n in (2, 5, 8, 11):
        sd[f"encoder.{n}.weight"] = _uniform(f"enc/fc{n}.w", (channels, channels), channels ** -0.5, seed)
    sd["encoder.14.weight"] = _uniform("enc/fc14.w", (z_dim, channels), channels ** -0.5, seed)
    sd["encoder.14.bias"] = _uniform("enc/fc14.b", (z_dim,), channels ** -0.5, seed)
    bound = 1.0 / 512 if codebook == "init" else 1.5
    sd["codebook.embedding"] = _uniform("enc/codebook." + codebook, (n_embeddings, z_dim), bound, seed)
    sd["codebook.ema_count"] = torch.zeros(n_embeddings)
    sd["codebook.ema_weight"] = sd["codebook.embedding"].clone()
    k = c_dim ** -0.5
    sd["rnn.weight_ih_l0"] = _uniform("enc/rnn.wih", (4 * c_dim, z_dim), k, seed)
    sd["rnn.weight_hh_l0"] = _uniform("enc/rnn.whh", (4 * c_dim, c_dim), k, seed)
    sd["rnn.bias_ih_l0"] = _uniform("enc/rnn.bih", (4 * c_dim,), k, seed)
    sd["rnn.bias_hh_l0"] = _uniform("enc/rnn.bhh", (4 * c_dim,), k, seed)
    # order keys like the reference's state_dict()
    order = ["conv.weight"]
    for n in (0, 2, 3, 5, 6, 8, 9, 11, 12, 14):
        order.append(f"encoder.{n}.weight")
        if n not in (2, 5, 8, 11):
            order.append(f"encoder.{n}.bias")
    order += ["codebook.embedding", "codebook.ema_count", "codebook.ema_weight",
              "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0"]
    return {k_: sd[k_] for k_ in order}


def vocoder_state_dict(seed: int = SEED, size_i_codebook=512, dim_i_embedding=64, n_speakers=102,
                       dim_speaker_embedding=64, dim_voc_latent=256, size_i_embed_ar=256,
                       size_h_rnn=896, size_h_fc=256, bits_mu_law=8):
    """Random-init ``Vocoder.state_dict()`` of the project spec (``config.py:58-77``)."""
    sd = {}
    sd["code_embedding.weight"] = _normalish("voc/code_emb", (size_i_codebook, dim_i_embedding), seed)
    sd["speaker_embedding.weight"] = _normalish("voc/spk_emb", (n_speakers, dim_speaker_embedding), seed)
    hp = dim_voc_latent // 2
    k = hp ** -0.5
    for layer in range(2):
        din = (dim_i_embedding + dim_speaker_embedding) if layer == 0 else 2 * hp
        for suf in ("", "_reverse"):
            p = f"rnnms.prenet.%s_l{layer}{suf}"
            sd[p % "weight_ih"] = _uniform("voc/" + p % "wih", (3 * hp, din), k, seed)
            sd[p % "weight_hh"] = _uniform("voc/" + p % "whh", (3 * hp, hp), k, seed)
            sd[p % "bias_ih"] = _uniform("voc/" + p % "bih", (3 * hp,), k, seed)
            sd[p % "bias_hh"] = _uniform("voc/" + p % "bhh", (3 * hp,), k, seed)
    n_cls = 2 ** bits_mu_law
    sd["rnnms.ar.embedding.weight"] = _normalish("voc/ar.emb", (n_cls, size_i_embed_ar), seed)
    k = size_h_rnn ** -0.5
    sd["rnnms.ar.rnn.weight_ih_l0"] = _uniform("voc/ar.wih", (3 * size_h_rnn, size_i_embed_ar + dim_voc_latent), k, seed)
    sd["rnnms.ar.rnn.weight_hh_l0"] = _uniform("voc/ar.whh", (3 * size_h_rnn, size_h_rnn), k, seed)
    sd["rnnms.ar.rnn.bias_ih_l0"] = _uniform("voc/ar.bih", (3 * size_h_rnn,), k, seed)
    sd["rnnms.ar.rnn.bias_hh_l0"] = _uniform("voc/ar.bhh", (3 * size_h_rnn,), k, seed)
    sd["rnnms.ar.fc1.weight"] = _uniform("voc/ar.fc1.w", (size_h_fc, size_h_rnn), size_h_rnn ** -0.5, seed)
    sd["rnnms.ar.fc1.bias"] = _uniform("voc/ar.fc1.b", (size_h_fc,), size_h_rnn ** -0.5, seed)
    sd["rnnms.ar.fc2.weight"] = _uniform("voc/ar.fc2.w", (n_cls, size_h_fc), size_h_fc ** -0.5, seed)
    sd["rnnms.ar.fc2.bias"] = _uniform("voc/ar.fc2.b", (n_cls,), size_h_fc ** -0.5, seed)
    return sd
