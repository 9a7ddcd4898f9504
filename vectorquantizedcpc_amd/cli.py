"""Command-line equivalents of the reference's ``encode.py`` and ``convert.py`` on MI355X.

    python -m vectorquantizedcpc_amd.cli encode  --dataset datasets/2019/english --out-dir out/z
                                                 [--cpc-checkpoint ckpt.pt | --random-init] [--save-auxiliary]
    python -m vectorquantizedcpc_amd.cli convert --dataset datasets/2019/english --synthesis-list list.json
                                                 --in-dir mels/ --out-dir out/wav
                                                 [--cpc-checkpoint .. --vocoder-checkpoint .. | --random-init] [--seed 13]

``encode`` mirrors ``encode.py:14-67``.  ``convert`` mirrors ``convert.py:17-83`` from the mel onwards: inputs are
``<in_dir>/<utterance>.mel.npy`` or, if absent, ``<utterance>.wav`` (16 kHz) run through the HIP mel
front-end (``preprocess.wave_to_mel`` = ``convert.py:54-70``).  For ``.wav`` inputs the output is re-normalised
to the input's integrated loudness (``convert.py:57,79-80``) by the HIP meter in ``loudness.py``; a ``.mel.npy``
input carries no reference loudness and its output is written as generated.  Utterances are batched by the
length-bucketed drivers; every output equals the batch-1 result.
"""
import argparse
import sys
from pathlib import Path

import torch

from . import ConfEncoder, ConfVocoder, Encoder, Vocoder, driver, io, loudness, synth


def _models(args, need_vocoder):
    dev = torch.device(args.device)
    enc = Encoder(ConfEncoder(80, 512, 512, 64, 256))
    enc.load_state_dict(synth.encoder_state_dict() if args.random_init else io.load_encoder_checkpoint(args.cpc_checkpoint))
    enc = enc.to(dev).eval()
    voc = None
    if need_vocoder:
        voc = Vocoder(ConfVocoder())
        voc.load_state_dict(synth.vocoder_state_dict() if args.random_init else
                            io.load_vocoder_checkpoint(args.vocoder_checkpoint, expected=voc.state_dict()))
        voc = voc.to(dev).eval()
    return enc, voc


def encode_dataset(args) -> int:
    paths = io.read_test_metadata(args.dataset)
    out_dir = Path(args.out_dir)
    out_dir.mkdir(exist_ok=True, parents=True)
    enc, _ = _models(args, need_vocoder=False)
    aux = []
    if args.save_auxiliary:                                    # encode.py:34-40
        enc.encoder[-1].register_forward_hook(lambda m, i, o: aux.append(o.clone()))
    mels = [io.load_mel(p) for p in paths]
    if args.save_auxiliary:
        # the hook delivers one batch at a time: keep the reference's one-utterance-per-call order
        for p, mel in zip(paths, mels):
            z, c, _ = enc.encode(mel[None].to(args.device))
            try:
                enc.check()                                    # nothing incomplete may be written
            except RuntimeError:
                aux.clear()
                z, c, _ = enc.encode(mel[None].to(args.device))
                enc.check()
            io.save_frames_text(out_dir / p.stem, z[0])
            for name, t in (("auxiliary_embedding1", c[0]), ("auxiliary_embedding2", aux.pop()[0])):
                d = out_dir.parent / name
                d.mkdir(exist_ok=True, parents=True)
                io.save_frames_text(d / p.stem, t)
    else:
        for p, r in zip(paths, driver.encode_utterances(enc, mels, max_batch=args.max_batch)):
            io.save_frames_text(out_dir / p.stem, r["z"])
    print(f"encoded {len(paths)} utterances -> {out_dir}")
    return 0


def convert_files(enc, voc, entries, out_dir, seed, max_batch: int = 64, slots: int = 0, timings=None):
    """``convert.py:52-83`` over ``entries`` = [(input path without suffix, speaker id, output name)]: ``.mel.npy`` inputs
    are used as they are; ``.wav`` inputs go through the batched HIP front end (resample at load, reference loudness, log-mel:
    ``driver.front_end_utterances``) and their outputs are re-normalised to the input's loudness (``convert.py:79-80``).
    ``timings``: a dict that receives wall seconds per stage (the device is synchronised at every stage boundary then)."""
    import time
    dev = next(enc.parameters()).device
    t_last = [time.perf_counter()]

    def clock(name):
        if timings is None:
            return
        torch.cuda.synchronize(dev)
        now = time.perf_counter()
        timings[name] = timings.get(name, 0.0) + now - t_last[0]
        t_last[0] = now

    mels = [None] * len(entries)
    wav_ids, waves, rates = [], [], []
    for i, (p, _, _) in enumerate(entries):
        if Path(p).with_suffix(".mel.npy").exists():
            mels[i] = io.load_mel(p).to(dev)
        else:
            rate, a = io.read_wav_file(p)
            wav_ids.append(i); waves.append(a); rates.append(rate)
    clock("read_files")
    ref = {}
    if wav_ids:
        fm, fl = driver.front_end_utterances(waves, rates, dev, max_batch=max_batch, clock=clock)
        for i, m, l in zip(wav_ids, fm, fl):
            mels[i] = m
            ref[i] = l
    wavs = driver.convert_utterances(enc, voc, mels, [s for _, s, _ in entries], seed=seed, max_batch=max_batch, slots=slots,
                                     clock=clock)
    if ref:                                                    # convert.py:79-80, one batched call each way
        ids = sorted(ref)
        for i, w in zip(ids, loudness.match_loudness([wavs[i] for i in ids], [ref[i] for i in ids])):
            wavs[i] = w
    clock("loudness_out")
    host = [w.cpu() for w in wavs]
    clock("download")
    for (_, _, name), w in zip(entries, host):
        io.save_wav(Path(out_dir) / name, w, 16000)
    clock("write_files")
    return wavs


def convert_dataset(args) -> int:
    items, _ = io.read_synthesis_list(args.synthesis_list, Path(args.dataset) / "speakers.json")
    in_dir, out_dir = Path(args.in_dir), Path(args.out_dir)
    out_dir.mkdir(exist_ok=True, parents=True)
    enc, voc = _models(args, need_vocoder=True)
    convert_files(enc, voc, [(in_dir / p, s, name) for p, s, name in items], out_dir, args.seed, max_batch=args.max_batch)
    print(f"converted {len(items)} utterances -> {out_dir}")
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="vectorquantizedcpc_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("encode", "convert"):
        p = sub.add_parser(name)
        p.add_argument("--dataset", required=True, help="datasets/<name> directory (test.json, speakers.json)")
        p.add_argument("--out-dir", required=True)
        p.add_argument("--cpc-checkpoint")
        p.add_argument("--random-init", action="store_true", help="seeded random-init weights (no checkpoint ships with the reference)")
        p.add_argument("--device", default="cuda")
        p.add_argument("--max-batch", type=int, default=64)
        if name == "encode":
            p.add_argument("--save-auxiliary", action="store_true")
        else:
            p.add_argument("--vocoder-checkpoint")
            p.add_argument("--synthesis-list", required=True)
            p.add_argument("--in-dir", required=True)
            p.add_argument("--seed", type=int, default=synth.SEED)
    args = ap.parse_args(argv)
    if not args.random_init and not args.cpc_checkpoint:
        ap.error("give --cpc-checkpoint (and --vocoder-checkpoint for convert) or --random-init")
    return encode_dataset(args) if args.cmd == "encode" else convert_dataset(args)


if __name__ == "__main__":
    sys.exit(main())
