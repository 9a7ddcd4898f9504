"""Drop-in for the inference-side functions of the reference's ``preprocess.py``.

``wave_to_mel(wave, conf)`` keeps the reference signature (``preprocess.py:53``) but runs on the HIP
device through ``vqcpc_melfront_*`` (and also accepts a padded batch with ``lengths``).  ``mulaw_encode`` /
``mulaw_decode`` are the reference's numpy formulas (``preprocess.py:20-35``), kept for completeness.
librosa is not available offline: the HIP front-end is checked against ``oracle/mel_ref.py`` -- parity unpinned.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


@dataclass
class ConfPreprocessing:
    """``preprocess.py:38-50``; defaults ``config.py:103-112``."""
    sr: int = 16000
    n_fft: int = 2048
    n_mels: int = 80
    fmin: int = 50
    preemph: float = 0.97
    top_db: int = 80
    hop_length: int = 160
    win_length: int = 400
    bits: int = 8


_handles = {}


def _handle(conf: ConfPreprocessing, device):
    key = (conf.sr, conf.n_fft, conf.n_mels, conf.hop_length, conf.win_length, float(conf.fmin), float(conf.preemph),
           float(conf.top_db), device.index)
    if key not in _handles:
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.load().vqcpc_melfront_create(conf.sr, conf.n_fft, conf.n_mels, conf.hop_length, conf.win_length,
                                                         float(conf.fmin), float(conf.preemph), float(conf.top_db), C.byref(h)))
        _handles[key] = h
    return _handles[key]


@torch.no_grad()
def wave_to_mel(wave, conf: ConfPreprocessing = None, lengths=None) -> torch.Tensor:
    """``preprocess.py:53-75``: waveform (L,) -> normalised log-mel (n_mels, 1 + L // hop).

    ``wave`` (B, Lmax) with ``lengths`` -> (B, n_mels, 1 + Lmax // hop), frames past each utterance's end zero.
    A numpy waveform is moved to the current HIP device (the reference's function takes numpy).
    """
    conf = conf or ConfPreprocessing()
    if isinstance(wave, np.ndarray):
        wave = torch.from_numpy(np.ascontiguousarray(wave, dtype=np.float32)).cuda()
    _lib.require_cuda(wave, "wave")
    single = wave.dim() == 1
    w = (wave[None] if single else wave).detach().to(torch.float32).contiguous()
    B, Lmax = w.shape
    lens = [Lmax] * B if lengths is None else [int(v) for v in lengths]
    out = torch.empty(B, conf.n_mels, 1 + Lmax // conf.hop_length, device=w.device)
    arr = (C.c_int * B)(*lens)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().vqcpc_melfront_run(_handle(conf, w.device), w.data_ptr(), arr, B, Lmax, out.data_ptr(),
                                                  _lib.current_stream()))
    return out[0] if single else out


def mulaw_encode(x, mu: int):
    """``preprocess.py:20-27``: linear [-1, 1] -> discrete [0, mu)."""
    mu = mu - 1
    fx = np.sign(x) * np.log1p(mu * np.abs(x)) / np.log1p(mu)
    return np.floor((fx + 1) / 2 * mu + 0.5)


def mulaw_decode(y, mu: int):
    """``preprocess.py:30-35``: mu-law [-1, 1] -> linear [-1, 1]."""
    mu = mu - 1
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


_resamplers = {}


@torch.no_grad()
def resample(wave, sr_in: int, sr_out: int, lengths=None) -> torch.Tensor:
    """The resampling ``librosa.load(path, sr=sr_out)`` applies to a file stored at ``sr_in`` (``convert.py:54-56``):
    librosa ^0.8 ``res_type='kaiser_best'`` (resampy's band-limited sinc interpolation) on the HIP device.

    ``wave`` (L,) -> (ceil(L * sr_out / sr_in),); (B, Lmax) with ``lengths`` -> (B, ceil(Lmax * ratio)), each row zero behind
    its own resampled length.  resampy is not available offline: checked against ``oracle/resample_ref.py`` -- parity unpinned.
    """
    if isinstance(wave, np.ndarray):
        wave = torch.from_numpy(np.ascontiguousarray(wave, dtype=np.float32)).cuda()
    _lib.require_cuda(wave, "wave")
    single = wave.dim() == 1
    w = (wave[None] if single else wave).detach().to(torch.float32).contiguous()
    if sr_in == sr_out:
        return w[0].clone() if single else w.clone()
    B, Lmax = w.shape
    lens = [Lmax] * B if lengths is None else [int(v) for v in lengths]
    key = (int(sr_in), int(sr_out), w.device.index)
    lib = _lib.load()
    if key not in _resamplers:
        h = C.c_void_p()
        with torch.cuda.device(w.device):
            _lib.check(lib.vqcpc_resampler_create(int(sr_in), int(sr_out), C.byref(h)))
        _resamplers[key] = h
    h = _resamplers[key]
    Lout = lib.vqcpc_resampler_out_len(h, Lmax)
    out = torch.empty(B, Lout, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.vqcpc_resampler_run(h, w.data_ptr(), (C.c_int * B)(*lens), B, Lmax, out.data_ptr(), Lout, _lib.current_stream()))
    return out[0] if single else out
