"""Drop-in for the two pyloudnorm calls of ``convert.py`` -- on the HIP device.

``import vectorquantizedcpc_amd.loudness as pyloudnorm`` keeps ``convert.py:50,57,79,80`` as written::

    meter = pyloudnorm.Meter(sr)
    ref_loudness = meter.integrated_loudness(wav)
    output = pyloudnorm.normalize.loudness(output, output_loudness, ref_loudness)

Both also take a padded batch (``lengths=``) and then stay on the device (fp64 tensors of LUFS), which is what
``match_loudness`` and the batched ``convert`` CLI use.  Kernels: ``csrc/loudness.hip`` through
``vqcpc_loudness_*``; pyloudnorm is not available offline, so parity is against ``oracle/loudness_ref.py``
(BS.1770-4 as pyloudnorm 0.1 states it) -- parity unpinned.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_handles = {}


def _handle(rate: int, device):
    key = (int(rate), device.index)
    if key not in _handles:
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.load().vqcpc_loudness_create(int(rate), C.byref(h)))
        _handles[key] = h
    return _handles[key]


def _as_batch(data, lengths):
    if isinstance(data, np.ndarray):
        data = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).cuda()
    _lib.require_cuda(data, "data")
    single = data.dim() == 1
    w = data[None] if single else data
    if w.dim() != 2:
        raise ValueError("mono audio only: (L,) or (B, Lmax) with lengths")
    lens = [w.shape[1]] * w.shape[0] if lengths is None else [int(v) for v in lengths]
    if len(lens) != w.shape[0]:
        raise ValueError("lengths must have one entry per row")
    return single, w, lens


class Meter:
    """``pyloudnorm.Meter(rate)`` with the default K-weighting filter and 400 ms blocks (``convert.py:50``)."""

    def __init__(self, rate: int):
        self.rate = int(rate)

    @torch.no_grad()
    def integrated_loudness(self, data, lengths=None, return_blocks: bool = False):
        """(L,) -> float LUFS (``convert.py:57,79``).  (B, Lmax) + ``lengths`` -> (B,) float64 device tensor.

        Raises ``ValueError`` for audio shorter than one 400 ms block, as pyloudnorm's ``valid_audio`` does.
        """
        single, w, lens = _as_batch(data, lengths)
        w = w.detach().to(torch.float32).contiguous()
        B, Lmax = w.shape
        lib = _lib.load()
        h = _handle(self.rate, w.device)
        nblk = [lib.vqcpc_loudness_blocks(h, n) if 0 < n <= Lmax else 0 for n in lens]
        if min(nblk) == 0:
            raise ValueError("Audio must have length greater than the block size.")
        lufs = torch.empty(B, dtype=torch.float64, device=w.device)
        z = torch.empty(sum(nblk), dtype=torch.float64, device=w.device) if return_blocks else None
        with torch.cuda.device(w.device):
            _lib.check(lib.vqcpc_loudness_integrated(h, w.data_ptr(), (C.c_int * B)(*lens), B, Lmax, lufs.data_ptr(),
                                                     z.data_ptr() if return_blocks else None, _lib.current_stream()))
        out = float(lufs.item()) if single else lufs
        return (out, list(torch.split(z, nblk))) if return_blocks else out


@torch.no_grad()
def loudness(data, input_loudness, target_loudness, lengths=None, rate: int = 16000):
    """``pyloudnorm.normalize.loudness`` (``convert.py:80``): ``data * 10^((target - input) / 20)``, fp32 out.

    Loudness values are floats or (B,) tensors; the result is a new tensor on the device of ``data``.
    """
    single, w, lens = _as_batch(data, lengths)
    out = w.detach().to(torch.float32).clone().contiguous()
    B, Lmax = out.shape
    meas = torch.as_tensor(input_loudness, dtype=torch.float64).to(out.device).reshape(-1).contiguous()
    targ = torch.as_tensor(target_loudness, dtype=torch.float64).to(out.device).reshape(-1).contiguous()
    if meas.numel() != B or targ.numel() != B:
        raise ValueError("one input and one target loudness per row")
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().vqcpc_loudness_normalize(_handle(rate, out.device), out.data_ptr(), (C.c_int * B)(*lens), B, Lmax,
                                                        meas.data_ptr(), targ.data_ptr(), _lib.current_stream()))
    return out[0] if single else out


class normalize:
    """Namespace so that ``pyloudnorm.normalize.loudness(...)`` reads the same."""
    loudness = staticmethod(loudness)


@torch.no_grad()
def match_loudness(wavs, ref_lufs, rate: int = 16000):
    """``convert.py:79-80`` for a list of generated waveforms: measure each, scale it to its reference loudness.

    ``wavs``: list of (L_i,) device tensors; ``ref_lufs``: (B,) floats/tensor.  One batched call each way.
    """
    lens = [int(w.numel()) for w in wavs]
    batch = torch.zeros(len(wavs), max(lens), device=wavs[0].device)
    for i, w in enumerate(wavs):
        batch[i, : lens[i]] = w
    got = Meter(rate).integrated_loudness(batch, lengths=lens)
    out = loudness(batch, got, ref_lufs, lengths=lens, rate=rate)
    return [out[i, : lens[i]] for i in range(len(wavs))]
