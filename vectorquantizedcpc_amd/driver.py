"""Batched, length-bucketed drivers for the reference's per-utterance loops.

``encode.py:42-46`` and ``convert.py:52-77`` process ONE utterance per call.  On a GPU the
throughput comes from batching, so these drivers bucket utterances by length, zero-pad inside a
bucket and call the batched HIP path -- while returning, for every utterance, exactly what a
batch-1 call on that utterance alone returns:

* zero right-padding is exact for the valid frames: ``nn.Conv1d(k=4, s=2, p=1)`` zero-pads there
  anyway (``model.py:43``), everything after it is per-frame or causal (LSTM, AR loop);
* the reference's conv back-end (and with it the fp32 summation order) depends on the CALL SHAPE
  (``vqcpc.h``: conv_mode).  A batch-1 call uses im2col order up to 256 frames and the oneDNN order
  above, so buckets never mix the two classes and pass the batch-1 mode explicitly;
* every utterance keeps its own sampling-stream id (``utt_ids``), so the drawn samples do not
  depend on the bucketing.
"""
from typing import List, Optional, Sequence

import torch

from .model import Encoder
from .network_vocoder import Vocoder


def out_frames(T: int) -> int:
    """Code frames of a T-frame mel: Conv1d(k=4, s=2, p=1) output length (``model.py:43``)."""
    return (T - 2) // 2 + 1


def batch1_conv_mode(in_channels: int, T: int) -> int:
    """Back-end ATen picks for a (1, C, T) call: 1 = im2col, 2 = oneDNN direct (``vqcpc.h``)."""
    return 2 if in_channels * T > 20480 else 1


def make_buckets(lengths: Sequence[int], modes: Sequence[int], max_batch: int, max_pad_frac: float):
    """Greedy buckets over utterances sorted by (mode, length): at most ``max_batch`` members and
    at most ``max_pad_frac`` of a bucket's frames are padding."""
    order = sorted(range(len(lengths)), key=lambda i: (modes[i], lengths[i], i))
    buckets, cur = [], []
    for i in order:
        if cur:
            lo, hi = lengths[cur[0]], lengths[i]
            total = sum(lengths[j] for j in cur) + lengths[i]
            too_padded = hi * (len(cur) + 1) - total > max_pad_frac * hi * (len(cur) + 1)
            if len(cur) >= max_batch or modes[i] != modes[cur[0]] or too_padded or lo <= 0:
                buckets.append(cur)
                cur = []
        cur.append(i)
    if cur:
        buckets.append(cur)
    return buckets


def generate_checked(vocoder: Vocoder, idx, spk, **kw):
    """``Vocoder.generate`` with its default check: one stream synchronisation, the handle's status word read, and the call
    repeated ONCE if an in-kernel hand-off of the resident decoders gave up (``convert.py:75-83`` writes the waveform right
    after ``generate``; nothing incomplete may reach it).  Kept as a name for the callers that must not pass ``async_``."""
    kw.pop("async_", None)
    return vocoder.generate(idx, spk, **kw)


def _pad_stack(mels: Sequence[torch.Tensor], ids: Sequence[int], device) -> torch.Tensor:
    T = max(int(mels[i].shape[-1]) for i in ids)
    out = torch.zeros(len(ids), mels[ids[0]].shape[0], T, device=device)
    for k, i in enumerate(ids):
        out[k, :, : mels[i].shape[-1]] = mels[i].to(device)
    return out


@torch.no_grad()
def encode_utterances(encoder: Encoder, mels: Sequence[torch.Tensor], want_context: bool = False,
                      max_batch: int = 64, max_pad_frac: float = 0.25):
    """``encode.py:42-46`` over a list of (80, T_i) mels -> list of dicts with the per-utterance
    ``z`` (T_i', 64), ``indices`` (T_i') and, if asked, ``c`` (T_i', 256)."""
    dev = next(encoder.parameters()).device
    C = encoder.conf.in_channels
    lengths = [int(m.shape[-1]) for m in mels]
    modes = [batch1_conv_mode(C, t) for t in lengths]
    out: List[Optional[dict]] = [None] * len(mels)
    for ids in make_buckets(lengths, modes, max_batch, max_pad_frac):
        batch = _pad_stack(mels, ids, dev)
        z, c, idx, _ = encoder._encode_native(batch, want_c=want_context, conv_mode=modes[ids[0]])
        if want_context:                         # the resident context scan of a one-utterance call may have given up
            try:
                encoder.check()
            except RuntimeError as e:
                import warnings
                warnings.warn(f"encode repeated on the fallback path: {e}")
                z, c, idx, _ = encoder._encode_native(batch, want_c=True, conv_mode=modes[ids[0]])
                encoder.check()
        for k, i in enumerate(ids):
            n = out_frames(lengths[i])
            out[i] = {"z": z[k, :n], "indices": idx[k, :n], "c": c[k, :n] if want_context else None}
    return out


@torch.no_grad()
def convert_utterances(encoder: Encoder, vocoder: Vocoder, mels: Sequence[torch.Tensor], speakers: Sequence[int],
                       seed: int, utt_ids: Optional[Sequence[int]] = None, max_batch: int = 64,
                       max_pad_frac: float = 0.25, slots: int = 0, clock=None,
                       mem_budget_bytes: int = 8 << 30) -> List[torch.Tensor]:
    """``convert.py:72-77`` over a list of utterances -> list of 1-D waveforms (160 * 2 * T_i' samples).

    ``slots`` > 0: continuous batching -- decode calls over as many utterances as ``mem_budget_bytes`` of device work space
    allow (see ``decode_chunks``; the reference's 9 474-utterance set, ``README.md:125``, is ~35 GB of conditioning rows in one
    call), each with that many decode slots, a slot running utterances back to back (longest first), instead of one call per
    length bucket.  The samples are the same either way (per-utterance sampling streams).
    """
    dev = next(encoder.parameters()).device
    C = encoder.conf.in_channels
    up = vocoder.conf.rnnms.upsampling_t
    lengths = [int(m.shape[-1]) for m in mels]
    modes = [batch1_conv_mode(C, t) for t in lengths]
    utt_ids = list(range(len(mels))) if utt_ids is None else list(utt_ids)
    n_codes_all = [out_frames(t) for t in lengths]
    out: List[Optional[torch.Tensor]] = [None] * len(mels)
    codes: List[Optional[torch.Tensor]] = [None] * len(mels)
    for ids in make_buckets(lengths, modes, max_batch, max_pad_frac):
        batch = _pad_stack(mels, ids, dev)
        idx = encoder._encode_native(batch, want_c=False, conv_mode=modes[ids[0]])[2]
        if clock: clock("encode")
        if slots > 0:
            for k, i in enumerate(ids):
                codes[i] = idx[k, : n_codes_all[i]]
            continue
        n_codes = [n_codes_all[i] for i in ids]
        spk = torch.tensor([int(speakers[i]) for i in ids], device=dev)
        wav = generate_checked(vocoder, idx, spk, n_codes=n_codes, seed=seed, utt_ids=[utt_ids[i] for i in ids])
        if clock: clock("decode")
        for k, i in enumerate(ids):
            out[i] = wav[k, : 2 * up * n_codes[k]]
    if slots > 0:
        vocoder.set_option("slots", slots)
        try:
            for ids in decode_chunks(n_codes_all, mem_budget_bytes, up):
                idx = torch.zeros(len(ids), max(n_codes_all[i] for i in ids), dtype=torch.int64, device=dev)
                for k, i in enumerate(ids):
                    idx[k, : codes[i].numel()] = codes[i]
                spk = torch.tensor([int(speakers[i]) for i in ids], device=dev)
                wav = generate_checked(vocoder, idx, spk, n_codes=[n_codes_all[i] for i in ids], seed=seed, utt_ids=[utt_ids[i] for i in ids])
                for k, i in enumerate(ids):
                    out[i] = wav[k, : 2 * up * n_codes_all[i]].clone() if len(ids) < len(mels) else wav[k, : 2 * up * n_codes_all[i]]
        finally:
            vocoder.set_option("slots", 0)
        if clock: clock("decode")
    return out


# device bytes a decode call needs per conditioning frame of an utterance's OWN (ragged rows: glue series 128 + hoisted gate
# inputs 768 + two prenet layers' outputs 2 x 256 + the Gcond row 3 x 896, fp32), and per code / output sample of the padded
# (B, T_max) grids that remain (int64 code indices, fp32 waveform)
BYTES_PER_OWN_FRAME = 4 * (128 + 768 + 256 + 256 + 3 * 896)
BYTES_PER_PADDED_FRAME = 4                  # half an int64 code index
BYTES_PER_PADDED_SAMPLE = 4


def decode_chunks(n_codes: Sequence[int], mem_budget_bytes: int, upsample: int = 160) -> List[List[int]]:
    """Utterance ids, in order, cut into decode calls whose device work space stays under ``mem_budget_bytes`` (at least one
    utterance per call).  Sizes for the reference's dimensions (``config.py:62-77``)."""
    chunks, cur, own, tmax = [], [], 0, 0
    for i, nc in enumerate(n_codes):
        f = 2 * int(nc)
        t2 = max(tmax, f)
        need = (own + f) * BYTES_PER_OWN_FRAME + (len(cur) + 1) * t2 * (BYTES_PER_PADDED_FRAME + BYTES_PER_PADDED_SAMPLE * upsample)
        if cur and need > mem_budget_bytes:
            chunks.append(cur)
            cur, own, t2 = [], 0, f
        cur.append(i)
        own += f
        tmax = t2
    if cur:
        chunks.append(cur)
    return chunks


@torch.no_grad()
def front_end_utterances(waves, rates, device, sr: int = 16000, max_batch: int = 64, max_pad_frac: float = 0.25, conf=None,
                         clock=None):
    """``convert.py:54-70`` for a list of mono waveforms (numpy / CPU tensors at their files' own rates): resample to ``sr``
    (``librosa.load(sr=...)``), reference loudness (``convert.py:57``, before the peak normalisation), log-mel -- each ONE
    batched call per length bucket instead of three synchronising calls per utterance (the C ABI takes (B, Lmax) + lengths).
    Returns (list of (80, T_i) device mels, list of float reference LUFS).  ``clock(name)``: optional stage timer."""
    from . import loudness, preprocess
    n = len(waves)
    mels, ref = [None] * n, [None] * n
    meter = loudness.Meter(sr)
    lens_in = [int(len(w)) for w in waves]
    hop = (conf or preprocess.ConfPreprocessing()).hop_length
    for ids in make_buckets(lens_in, [int(r) for r in rates], max_batch, max_pad_frac):
        rate = int(rates[ids[0]])
        L = max(lens_in[i] for i in ids)
        batch = torch.zeros(len(ids), L, device=device)               # each utterance straight into its padded row (one copy)
        for k, i in enumerate(ids):
            batch[k, : lens_in[i]].copy_(torch.as_tensor(waves[i], dtype=torch.float32), non_blocking=True)
        lens = [lens_in[i] for i in ids]
        if clock: clock("upload")
        if rate != sr:
            batch = preprocess.resample(batch, rate, sr, lengths=lens)
            lens = [-(-l * sr // rate) for l in lens]                       # ceil(l * sr / rate), as librosa.resample(fix=True)
            if clock: clock("resample")
        lufs = meter.integrated_loudness(batch, lengths=lens)
        if clock: clock("loudness_in")
        mel = preprocess.wave_to_mel(batch, conf, lengths=lens)
        lufs = lufs.tolist()
        for k, i in enumerate(ids):
            mels[i] = mel[k, :, : 1 + lens[k] // hop]
            ref[i] = lufs[k]
        if clock: clock("mel")
    return mels, ref
