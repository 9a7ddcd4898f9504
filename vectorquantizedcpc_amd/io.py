"""On-disk formats of the reference's inference CLIs.

* input mels: ``<path>.mel.npy`` float (80, T) (``encode.py:44``)
* metadata: ``test.json`` = list of ``[.., .., .., path]`` (``encode.py:18-20, :42``);
  ``speakers.json`` = list of speaker names, sorted (``convert.py:19-20``);
  synthesis list = ``[[utterance_path, speaker_id, out_filename], ...]`` (``convert.py:22-24, :52``)
* encoder checkpoint: ``torch.load(path)["encoder"]`` (``train_cpc.py:23-29``, ``encode.py:29-30``)
* code vectors: one text row per frame, ``np.savetxt(fmt="%.16f")`` (``encode.py:50-52``), same for the
  auxiliary context / pre-VQ dumps (``encode.py:54-67``)
* audio out: 16 kHz float32 WAV (``convert.py:83`` wrote ``output.astype(np.float32)``)
"""
import json
from pathlib import Path
from typing import Dict, List, Tuple

import numpy as np
import torch


def load_mel(path) -> torch.Tensor:
    """``np.load(path.with_suffix('.mel.npy'))`` -> float32 (80, T) tensor (``encode.py:44``)."""
    p = Path(path)
    if p.suffix != ".npy":
        p = p.with_suffix(".mel.npy")
    a = np.load(p, allow_pickle=False)
    if a.ndim != 2:
        raise ValueError(f"{p}: expected a (n_mels, T) array, got shape {a.shape}")
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def read_wav_file(path) -> Tuple[int, np.ndarray]:
    """(rate, mono float32 samples in [-1, 1]) of a WAV file as soundfile / librosa decode it: integer PCM is scaled by its
    full range FIRST (int16 / 2^15, int32 / 2^31, uint8 (a - 128) / 128), then the channels are averaged
    (``librosa.load(mono=True)``) -- averaging first would turn the integers into floats that are never scaled."""
    from scipy.io import wavfile
    rate, a = wavfile.read(str(Path(path).with_suffix(".wav")))
    if a.dtype.kind == "i":
        a = a.astype(np.float32) / float(np.iinfo(a.dtype).max + 1)
    elif a.dtype.kind == "u":
        a = (a.astype(np.float32) - 128.0) / 128.0
    else:
        a = a.astype(np.float32)
    if a.ndim > 1:
        a = a.mean(axis=1, dtype=np.float32)
    return int(rate), np.ascontiguousarray(a, dtype=np.float32)


def load_wav(path, sr: int = 16000) -> torch.Tensor:
    """Mono waveform at ``sr`` as float32 in [-1, 1]: ``librosa.load(path, sr=sr)`` (``convert.py:54-56``).  A file stored
    at another rate is resampled like librosa does (``preprocess.resample``: kaiser_best sinc interpolation, on the HIP
    device -- so that case needs a GPU; the result comes back as a CPU tensor like the other case)."""
    rate, a = read_wav_file(path)
    w = torch.from_numpy(a)
    if rate != sr:
        from .preprocess import resample
        w = resample(w.cuda(), rate, sr).cpu()
    return w


def save_frames_text(path, frames) -> None:
    """One row per frame, ``%.16f`` (``encode.py:50-52``)."""
    a = frames.detach().cpu().numpy() if hasattr(frames, "detach") else np.asarray(frames)
    with open(Path(path).with_suffix(".txt"), "w") as f:
        np.savetxt(f, a, fmt="%.16f")


def load_frames_text(path) -> np.ndarray:
    return np.loadtxt(Path(path).with_suffix(".txt"), dtype=np.float64, ndmin=2).astype(np.float32)


def save_wav(path, wav, sr: int = 16000) -> None:
    """float32 WAV (``convert.py:83``)."""
    from scipy.io import wavfile
    a = wav.detach().cpu().numpy() if hasattr(wav, "detach") else np.asarray(wav)
    wavfile.write(str(Path(path).with_suffix(".wav")), sr, a.astype(np.float32))


def load_encoder_checkpoint(path) -> Dict[str, torch.Tensor]:
    """``checkpoint["encoder"]`` with a loader that executes nothing from the file."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    return ck["encoder"] if "encoder" in ck else ck


def load_vocoder_checkpoint(path, expected: Dict[str, torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """``checkpoint["vocoder"]`` (``convert.py:45``) or a Lightning checkpoint rooted at ``VocoderModel.model``
    (``vocoder.py:47``), loaded without executing anything from the file.

    ``expected`` = ``Vocoder(conf).state_dict()``: the key names under ``rnnms.*`` are this project's GUESS at
    the absent third-party module's layout, so a real checkpoint is matched to them by
    ``remap_state_dict`` (exact name, else unique suffix + shape) rather than trusted to ``load_state_dict``.
    """
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if "vocoder" in ck:
        sd = ck["vocoder"]
    elif "state_dict" in ck:
        sd = {k[len("model."):]: v for k, v in ck["state_dict"].items() if k.startswith("model.")}
    else:
        sd = ck
    return remap_state_dict(sd, expected) if expected is not None else sd


def _suffixes(key: str):
    parts = key.split(".")
    for n in range(len(parts), 0, -1):                 # longest suffix first
        yield ".".join(parts[-n:])


def remap_state_dict(found: Dict[str, torch.Tensor], expected: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Rename the tensors of ``found`` to the keys of ``expected``.

    An expected key takes the found key of the same name; otherwise the ONE unused found key that shares its
    longest dotted suffix (``...prenet.net.weight_ih_l0`` for ``rnnms.prenet.weight_ih_l0``, at least the last
    component) and has the same shape.  Anything ambiguous, missing or left over raises ``KeyError`` listing
    expected versus found names and shapes -- never a silent partial load.
    """
    out, used = {}, set()
    for k, t in expected.items():
        if k in found and tuple(found[k].shape) == tuple(t.shape):
            out[k] = found[k]
            used.add(k)
    problems = []
    for k, t in expected.items():
        if k in out:
            continue
        pick = None
        for suf in _suffixes(k):
            cands = [f for f in found if f not in used and (f == suf or f.endswith("." + suf))
                     and tuple(found[f].shape) == tuple(t.shape)]
            if len(cands) == 1:
                pick = cands[0]
                break
            if len(cands) > 1:
                problems.append(f"{k} {tuple(t.shape)}: ambiguous, candidates {sorted(cands)}")
                break
        if pick is not None:
            out[k] = found[pick]
            used.add(pick)
        elif not any(p.startswith(k + " ") for p in problems):
            problems.append(f"{k} {tuple(t.shape)}: no tensor with a matching name suffix and shape")
    extra = [f"{f} {tuple(found[f].shape)}" for f in found if f not in used]
    if problems or extra:
        raise KeyError("checkpoint does not fit the module.\n  unresolved expected keys:\n    " + "\n    ".join(problems or ["-"])
                       + "\n  unused checkpoint keys:\n    " + "\n    ".join(extra or ["-"])
                       + "\n  expected: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in expected.items())
                       + "\n  found: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in found.items()))
    return out


def read_test_metadata(dataset_root) -> List[Path]:
    """``datasets/<name>/test.json`` -> utterance paths relative to ``datasets/`` (``encode.py:18-20, :42-43``)."""
    root = Path(dataset_root)
    with open(root / "test.json") as f:
        meta = json.load(f)
    return [root.parent / row[3] for row in meta]


def read_synthesis_list(list_path, speakers_path) -> Tuple[List[Tuple[str, int, str]], List[str]]:
    """(utterance, speaker index, output name) triples and the sorted speaker list (``convert.py:19-24, :73``)."""
    with open(speakers_path) as f:
        speakers = sorted(json.load(f))
    with open(list_path) as f:
        items = json.load(f)
    return [(str(p), speakers.index(s), str(o)) for p, s, o in items], speakers
