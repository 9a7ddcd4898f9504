#!/usr/bin/env python3
"""bench.py -- VQ-CPC inference hot path on MI355X: audio samples/s of the WaveRNN-style decode
(+ encoder frames/s), one process per GPU.

A "step" = one pass of the convert.py path (convert.py:72-77) over one per-GPU batch of synthetic
utterances that are already resident in HBM: mel (B, 80, 200) -> Encoder.encode indices ->
Vocoder.generate (B x 32 000 samples).  The per-GPU batch is BASELINE.json configs[3]'s shard
(256 utterances / 8 GPUs = 32, 2 s each), the same at every N (weak scaling); for N > 1 the
waveforms are gathered on rank 0 over RCCL inside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torch.distributed.run environment, this process starts the N ranks
itself (fresh children of `python -m torch.distributed.run`, before it makes any GPU call),
relays rank 0's JSON line and exits with the children's status.  Under torch.distributed.run
(RANK / WORLD_SIZE set) it is one of the ranks.

Rank 0 prints ONE JSON line.  Extra keys next to the contract's: `roofline` (decode step vs the
fp32 MFMA peak), `cpu_baseline` (the PyTorch-CPU port of the same path on this box's host cores,
bounded samples, 1 thread and all usable cores), `encoder` (BASELINE configs[1], 64 x 128
frames, + configs[0]'s 1 x 200), `single_utterance` (configs[2], 1 x 32 000 samples),
`one_gpu_256` (configs[3]'s whole batch on one GPU, with its own roofline), `manifest`
(configs[4] stand-in), `teacher_forced` (SURVEY 8f-4 shape); for N > 1 `rccl_ranks` and `gather`.

  python bench.py --workload manifest [--gpus N] [--manifest M] [--force-gather]

BASELINE configs[4] (full test-set synthesis, convert.py:52-83 sharded): every rank builds the same seeded ragged manifest of M
utterances, takes its LPT share (shard.convert_sharded), encodes in length buckets and decodes by continuous batching, and the
waveforms are gathered on rank 0 (all_gather of the block shapes, gather of the int64 side table, gather of the padded blocks:
shard.gather_waveforms).  The line carries samples/s of the whole job, per-rank decode seconds (load imbalance) and the gather's
milliseconds.  --force-gather runs the three collectives at world size 1 as well (RCCL executes the ragged path on a 1-GPU box).
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import vectorquantizedcpc_amd as V                      # noqa: E402
from vectorquantizedcpc_amd import shard, synth         # noqa: E402

FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
HBM_PEAK_GBS = 8000.0
# algorithmic FLOP per decoded sample: W_hh 2 408 448 + embedding half of W_ih 688 128
# + fc1 229 376 + fc2 65 536 MAC, + the conditioning half of W_ih once per 160-sample frame
FLOP_PER_SAMPLE = 2 * (2408448 + 688128 + 229376 + 65536) + 2 * 688128 / 160.0
# the same without the 688 128 MACs of the sample-embedding half of W_ih, which every path replaces by a 256-row table
# lookup (Gemb): the arithmetic a decoded sample actually executes
FLOP_EXECUTED_PER_SAMPLE = 2 * (2408448 + 229376 + 65536) + 2 * 688128 / 160.0
FLOP_PER_FRAME = 2555904        # encoder, per output frame without the LSTM
GRU_MAC = 2408448               # W_hh MACs per sample: the dominant kernel's algorithmic work
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
KERNEL_SOURCES = ("vectorquantizedcpc_amd/csrc/vocoder.hip", "vectorquantizedcpc_amd/csrc/ar_xcd.hip",
                  "vectorquantizedcpc_amd/csrc/ar_xcm.hip", "vectorquantizedcpc_amd/csrc/ar_shared.h",
                  "vectorquantizedcpc_amd/csrc/ar_chain.h", "vectorquantizedcpc_amd/csrc/ar_xcd.h",
                  "vectorquantizedcpc_amd/csrc/common.h")


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def kernel_source_sha():
    """Identity of the decode kernels a PMC measurement belongs to (.git does not travel to the GPU box): a hash of their
    sources without `//` comments and blank lines, so that a comment edit does not orphan a measurement."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "r", encoding="utf-8") as f:
            for line in f:
                code = line.split("//", 1)[0].rstrip()          # no string literal in these files contains "//"
                if code.strip():
                    h.update(code.encode("utf-8") + b"\n")
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks as fresh children (no GPU call in this process)
# ------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv, check_devices=True, nproc=None):
    """Parent of a multi-GPU run.  torch.cuda.device_count() does not initialise the GPU on this image;
    nothing else here touches it.  Returns the exit status for sys.exit."""
    if not args.selftest_spawn and check_devices:
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} asked for but this node exposes {n_dev} GPU(s); "
                  f"nothing was run (use --gpus {max(n_dev, 1)} here)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc or args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    log("starting %d ranks: %s" % (args.gpus, " ".join(cmd[1:])))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:                       # relay rank 0's one JSON line, pass anything else to stderr
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        elif s:
            print(s, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        print(f"bench.py: a rank failed (torch.distributed.run exit status {rc})", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 3
    print(line, flush=True)
    return 0


def selftest_rank(args):
    """CPU rehearsal of the N > 1 plumbing (tests/test_bench_spawn_cpu.py): gloo ranks, a barrier, the
    gather on rank 0 and the JSON relay -- no kernels, nothing measured."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        if args.selftest_spawn == 2 and rank == world - 1:
            raise SystemExit(7)                   # a failing rank must fail the parent
        if args.selftest_spawn == 3:
            # the configs[4] leg (manifest_sharded: LPT shards, ragged gather, per-rank statistics) with a stand-in decode on CPU
            frames, spk = synthetic_manifest(args.manifest)
            mels = [torch.zeros(80, max(2, f // 50)) for f in frames]

            def fake_decode(ids, ms, speakers):
                L = max([m.shape[-1] // 2 * 2 * 160 for m in ms], default=0)
                out = torch.zeros(len(ids), L)
                for k, (i, m, s) in enumerate(zip(ids, ms, speakers)):
                    nn = m.shape[-1] // 2 * 2 * 160
                    out[k, :nn] = float(i) + 0.001 * s
                return out
            r = manifest_sharded(args, rank, world, torch.device("cpu"), mels, spk, fake_decode, None, True)
            if rank == 0:
                r["selftest"] = True
                print(json.dumps(r), flush=True)
            return
        wav = torch.full((2, 8), float(rank))
        out = [torch.empty_like(wav) for _ in range(world)] if rank == 0 else None
        dist.barrier()
        dist.gather(wav, out, dst=0)
        if rank == 0:
            ok = all(float(o[0, 0]) == r for r, o in enumerate(out))
            print("stray line on stdout", flush=True)      # must not reach the parent's stdout
            print(json.dumps({"metric": "selftest", "value": float(ok), "n_gpus": world, "rccl_ranks": dist.get_world_size(),
                              "backend": "gloo", "selftest": True}), flush=True)
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------
def manifest_sharded(args, rank, world, dev, mels, speakers, decode_fn, check_fn, launched):
    """BASELINE configs[4]: the ragged manifest over `world` ranks -- shard.convert_sharded (LPT by sample count, local decode,
    check, ragged gather on rank 0), timed as a whole between barriers; max over ranks.  Returns rank 0's result object."""
    ddp = launched and dist.is_initialized()
    n = len(mels)
    total_samples = sum(int(m.shape[-1]) // 2 * 2 * 160 for m in mels)
    st = {}
    for _ in range(max(args.warmup, 0)):
        shard.convert_sharded(mels, speakers, decode_fn, check_fn=check_fn, force_collective=args.force_gather)
    times = []
    res = None
    for _ in range(max(args.steps, 1)):
        if ddp:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res = shard.convert_sharded(mels, speakers, decode_fn, check_fn=check_fn, force_collective=args.force_gather, stats=st)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        if ddp:
            dist.barrier()
        times.append(time.perf_counter() - t0)
    dt = sum(times)
    per_rank = torch.tensor([dt, st["decode_s"], st["gather_s"], float(st["utterances"]), float(st["samples"])], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(per_rank) for _ in range(world)] if ddp else [per_rank]
    if ddp:
        dist.all_gather(allr, per_rank)
    if rank != 0:
        return {}
    rows = [[float(v) for v in r.tolist()] for r in allr]
    dt = max(r[0] for r in rows)
    dec = [r[1] for r in rows]
    ok = res is not None and len(res) == n and all(int(w.numel()) == int(m.shape[-1]) // 2 * 2 * 160 for w, m in zip(res, mels))
    value = total_samples * len(times) / dt
    return {
        "metric": "audio samples/sec (WaveRNN-style decode, convert.py path: encode + generate)",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": len(times), "warmup": args.warmup,
        "ms_per_step": dt / len(times) * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4] stand-in: ragged manifest of {n} utterances (log-normal 1-10 s, seeded; the reference "
                               f"ships no test-set manifest) -> LPT shards over {world} rank(s) -> encode in length buckets -> decode by "
                               f"continuous batching ({args.manifest_batch} slots asked for) -> ragged gather on rank 0",
                   "utterances": n, "audio_seconds": total_samples / 16000.0, "weights": "random-init (seed 13)",
                   "parallelism": f"utterance-sharded x{world} (LPT), one ragged gather (all_gather of block shapes + two gathers)"},
        "realtime_factor_16k": value / 16000.0,
        "all_utterances_gathered_with_their_lengths": bool(ok),
        "rccl_ranks": dist.get_world_size() if ddp else 1, "backend": dist.get_backend() if ddp else "none",
        "per_rank": {"utterances": [int(r[3]) for r in rows], "samples": [int(r[4]) for r in rows],
                     "decode_s_last_step": dec, "gather_ms_last_step": [r[2] * 1e3 for r in rows],
                     "decode_imbalance_max_over_mean": max(dec) / (sum(dec) / len(dec)) if sum(dec) > 0 else None},
        "gather": {"forced_at_world_1": bool(args.force_gather and world == 1), "ms_rank0_last_step": rows[0][2] * 1e3,
                   "how": "shard.gather_waveforms: all_gather of (count, L) per rank, gather of the (count_max, 2) int64 side table, "
                          "gather of the max-padded (count_max, L_max) fp32 blocks; device-synchronised on both sides"},
    }


def build_models(dev):
    enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
    enc.load_state_dict(synth.encoder_state_dict())
    voc = V.Vocoder(V.ConfVocoder())
    voc.load_state_dict(synth.vocoder_state_dict())
    return enc.to(dev).eval(), voc.to(dev).eval()


def timed(fn, steps, warmup, dev, world):
    ddp = dist.is_available() and dist.is_initialized()
    for _ in range(warmup):
        fn()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    if ddp:
        dist.barrier()
    dt = time.perf_counter() - t0
    if ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def wall(fn, dev, reps=1):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps, out


# ------------------------------------------------------------------------------------------
# CPU baseline: the PyTorch-CPU port on this box's host cores
# ------------------------------------------------------------------------------------------
def host_cpu_info():
    """os.cpu_count(), lscpu's physical core count, and the threads this process may actually use
    (affinity and cgroup quota; no other cap)."""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in txt.splitlines() if ":" in l}
        info["lscpu_model"] = kv.get("Model name")
        info["lscpu_cores"] = int(kv.get("Core(s) per socket", "0")) * int(kv.get("Socket(s)", "0")) or None
        info["lscpu_threads_per_core"] = int(kv.get("Thread(s) per core", "0")) or None
    except (OSError, ValueError, subprocess.SubprocessError):
        info["lscpu_cores"] = None
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    info["affinity"] = n
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            info["cgroup_quota_cpus"] = int(quota) / int(period)
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    info["usable_threads"] = max(1, n)
    return info


def median_time(fn, warmup=3, reps=10):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def cpu_baseline(n_utt):
    """Legs of SURVEY 8(d): encode C1 (1 x 200) and C2 (64 x 128), the AR loop at B = 1 (C3: 2 000 steps in
    10 timed chunks, extrapolated linearly to 32 000) and at B = n_utt (the bench workload), each with
    torch.set_num_threads(1) and (all usable threads), warm-up 3, median of 10.  `value` = the bench
    workload's leg on all usable threads."""
    from oracle import torch_ref
    info = host_cpu_info()
    esd, vsd = synth.encoder_state_dict(), synth.vocoder_state_dict()
    tv = torch_ref.TorchVocoder(vsd)
    mel1, mel2 = synth.mel("bench/c1", 1, 200), synth.mel("bench/c2", 64, 128)
    zB = synth.randint("bench/codes", (n_utt, 100), 512)
    z1 = synth.randint("bench/c3", (1, 100), 512)
    spkB, spk1 = torch.arange(n_utt) % 102, torch.zeros(1, dtype=torch.long)
    chunk1, chunkB = 200, 40                                       # decode steps per timed repetition
    noise1 = torch_ref.make_noise(1, chunk1, synth.SEED)           # the protocol's noise: generated outside the timing
    noiseB = torch_ref.make_noise(n_utt, chunkB, synth.SEED)
    cond1, condB = tv.condition(z1, spk1), tv.condition(zB, spkB)  # prenet once (it is < 0.1 % of a real call)
    legs = {}
    t_all = time.perf_counter()
    for threads in sorted({1, info["usable_threads"]}):
        torch.set_num_threads(threads)
        tag = f"{threads}t"
        d = median_time(lambda: torch_ref.encoder_encode(esd, mel1, want_c=True))
        legs[f"c1_encode_1x200_{tag}"] = {"ms": d * 1e3, "frames_per_s": 100 / d}
        d = median_time(lambda: torch_ref.encoder_encode(esd, mel2, want_c=True))
        legs[f"c2_encode_64x128_{tag}"] = {"ms": d * 1e3, "frames_per_s": 4096 / d}
        d = median_time(lambda: torch_ref.encoder_encode(esd, mel2, want_c=False))
        legs[f"c2_encode_64x128_no_context_{tag}"] = {"ms": d * 1e3, "frames_per_s": 4096 / d}
        d = median_time(lambda: tv.generate(z1, spk1, seed=synth.SEED, n_steps=chunk1, noise=noise1, cond=cond1))
        legs[f"c3_decode_b1_{tag}"] = {"us_per_step": d / chunk1 * 1e6, "samples_per_s": chunk1 / d,
                                       "seconds_for_32000_samples_extrapolated": d / chunk1 * 32000,
                                       "timed": f"median of 10 x {chunk1} steps (2 000 steps), extrapolated linearly"}
        d = median_time(lambda: tv.generate(zB, spkB, seed=synth.SEED, n_steps=chunkB, noise=noiseB, cond=condB))
        legs[f"decode_b{n_utt}_{tag}"] = {"us_per_step": d / chunkB * 1e6, "samples_per_s": n_utt * chunkB / d}
        log(f"cpu baseline, {threads} thread(s): " + ", ".join(f"{k} {list(v.values())[0]:.4g}" for k, v in legs.items() if k.endswith(tag)))
    total = time.perf_counter() - t_all
    nt = info["usable_threads"]
    main = legs[f"decode_b{n_utt}_{nt}t"]
    return {"value": main["samples_per_s"], "unit": "samples/s", "cores": nt, "kind": "port",
            "sample": f"{n_utt} concurrent utterances x {chunkB}-step chunks of the same decode workload, warm-up 3, median of 10 "
                      f"(PyTorch-CPU ops, {nt} threads; all legs together {total:.1f} s of CPU work)",
            "encoder_frames_per_s": legs[f"c2_encode_64x128_no_context_{nt}t"]["frames_per_s"],
            "host": info, "legs": legs}


# ------------------------------------------------------------------------------------------
PATH_NAMES = {0: "the launch-per-step kernels", 2: "the per-XCD resident decoders (ar_xcd_kernel)",
              3: "the matrix-core per-XCD resident decoders (ar_xcm_kernel)"}
# what bounds the dominant kernel of each decode path.  ar_xcd_kernel<4> (17..32 slots: the bench workload) runs its W_hh / fc1 chains on
# the matrix pipe (v_mfma_f32_4x4x1_16B_f32); with one or two slots per XCD (<= 16 utterances) the kernel issues NO MFMA instruction and
# is priced against the fp32 vector peak, which equals the fp32 MFMA peak (157.3 TFLOP/s).
BOUND = {0: "mfma", 2: "mfma", 3: "mfma"}


def xcd_bound(slots):
    return "mfma" if slots > 16 else "valu"


def synthetic_manifest(n, seed=synth.SEED):
    """BASELINE configs[4] stand-in: the reference ships no test-set manifest (datasets are
    git-ignored), so lengths are drawn log-normally between 1 and 10 s (seeded), speakers round-robin."""
    import numpy as np
    u = synth.uniform01("bench/manifest", 2 * n, seed).reshape(n, 2)
    g = np.sqrt(-2.0 * np.log(np.maximum(u[:, 0], 1e-12))) * np.cos(2 * np.pi * u[:, 1])      # host-side only
    secs = np.clip(np.exp(np.log(3.0) + 0.5 * g), 1.0, 10.0)
    frames = (secs * 100).astype(int)                           # 10 ms hop
    return [int(f) for f in frames], [i % 102 for i in range(n)]


def run_manifest(enc, voc, dev, n_utt, max_batch):
    from vectorquantizedcpc_amd import driver
    frames, spk = synthetic_manifest(n_utt)
    mels = [synth.mel(f"bench/man{i % 8}", 1, max(frames))[0][:, :f].contiguous().to(dev) for i, f in enumerate(frames)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    wavs = driver.convert_utterances(enc, voc, mels, spk, seed=synth.SEED, max_batch=64, max_pad_frac=0.15, slots=max_batch)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    samples = sum(int(w.numel()) for w in wavs)
    loop_ms, steps = voc.last_timing()                          # the decode loop alone (HIP events inside generate)
    slots = voc.last_slots()                                    # what the decode loop really ran through (the resident decoders cap the option)
    path = voc.last_path()
    return {"workload": f"synthetic manifest (configs[4] stand-in): {n_utt} utterances, log-normal 1-10 s, "
                        f"encoder in length buckets, decode by continuous batching over {slots} slots of {PATH_NAMES.get(path, path)} "
                        f"(asked for: {max_batch})",
            "slots": slots, "slots_asked_for": max_batch, "decode_path": path,
            "utterances": n_utt, "audio_seconds": samples / 16000.0, "wall_s": dt,
            "samples_per_s": samples / dt, "realtime_factor_16k": samples / 16000.0 / dt,
            "decode_loop_s": loop_ms * 1e-3, "decode_steps": int(steps), "us_per_step": loop_ms * 1e3 / max(steps, 1),
            "slot_occupancy": samples / float(max(steps, 1) * max(slots, 1)),
            "workspace_peak_bytes": voc.workspace_bytes(),
            "roofline": {"bound": BOUND.get(path, "mfma"), "achieved": FLOP_PER_SAMPLE * samples / (loop_ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": FLOP_PER_SAMPLE * samples / (loop_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                         "frac_executed": FLOP_EXECUTED_PER_SAMPLE * samples / (loop_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                         "how": "6 782 976 FLOP per decoded sample x samples of the manifest / decode loop time (HIP events)"}}


def run_convert_e2e(enc, voc, dev, n_utt, slots):
    """`convert.py:52-83` end to end, wav files in -> wav files out, as `python -m vectorquantizedcpc_amd.cli convert` runs it:
    `n_utt` synthetic 22.05 kHz int16 wavs on disk (log-normal 1-10 s, seeded) -> read -> resample to 16 kHz -> reference
    loudness -> log-mel (all three batched per length bucket on the GPU) -> encode -> decode (continuous batching over `slots`
    decode slots) -> loudness re-normalisation -> float32 wav files.  Wall seconds per stage (device synchronised at every
    stage boundary) and their shares; writing the synthetic inputs is not timed."""
    import shutil
    import tempfile
    import numpy as np
    from scipy.io import wavfile
    from vectorquantizedcpc_amd import cli
    frames, spk = synthetic_manifest(n_utt)
    tmp = tempfile.mkdtemp(prefix="vqcpc_e2e_")
    try:
        in_dir, out_dir = os.path.join(tmp, "in"), os.path.join(tmp, "out")
        os.makedirs(in_dir), os.makedirs(out_dir)
        entries = []
        for i, f in enumerate(frames):
            n = int(f * 160 * 22050 / 16000)
            u = synth.uniform01(f"e2e/{i % 16}", n + 64, synth.SEED)[:n]
            env = 0.25 + 0.5 * np.abs(np.sin(np.arange(n) * (2 * np.pi / 22050.0) * (0.7 + 0.1 * (i % 5))))
            wavfile.write(os.path.join(in_dir, f"u{i:04d}.wav"), 22050, ((u * 2.0 - 1.0) * env * 20000).astype(np.int16))
            entries.append((os.path.join(in_dir, f"u{i:04d}"), spk[i], f"o{i:04d}"))
        cli.convert_files(enc, voc, entries[:8], out_dir, synth.SEED, slots=slots)          # warm-up: handles, graphs, tables
        torch.cuda.synchronize(dev)
        tm = {}
        t0 = time.perf_counter()
        wavs = cli.convert_files(enc, voc, entries, out_dir, synth.SEED, slots=slots, timings=tm)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        samples = sum(int(w.numel()) for w in wavs)
        n_out = len([f for f in os.listdir(out_dir) if f.endswith(".wav")])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    front = sum(tm.get(k, 0.0) for k in ("read_files", "upload", "resample", "loudness_in", "mel"))
    back = sum(tm.get(k, 0.0) for k in ("loudness_out", "download", "write_files"))
    return {"workload": f"convert.py:52-83 end to end: {n_utt} wav files at 22.05 kHz (log-normal 1-10 s) -> resample -> loudness -> "
                        f"log-mel -> encode -> decode ({voc.last_slots()} slots; asked for {slots}) -> loudness -> {n_out} wav files at 16 kHz",
            "utterances": n_utt, "audio_seconds": samples / 16000.0, "wall_s": wall, "samples_per_s": samples / wall,
            "realtime_factor_16k": samples / 16000.0 / wall, "decode_slots": voc.last_slots(), "decode_path": voc.last_path(),
            "stage_s": {k: round(v, 4) for k, v in tm.items()},
            "stage_share": {k: round(v / wall, 4) for k, v in tm.items()},
            "front_end_and_io_share": (front + back) / wall, "model_share": (tm.get("encode", 0.0) + tm.get("decode", 0.0)) / wall}


def gru_roofline(voc, n_utt, step_us, n_steps=0, samples_per_utt=0):
    """`roofline` object for the GRU-step kernel of the LAST generate() call: HIP events around 2000
    back-to-back launches on the launch stream (vqcpc_vocoder_kernel_times)."""
    step_tflops = FLOP_PER_SAMPLE * n_utt / (step_us * 1e-6) / 1e12
    exec_frac = FLOP_EXECUTED_PER_SAMPLE / FLOP_PER_SAMPLE
    path = voc.last_path()
    if path in (2, 3):
        # ONE launch for the whole call: eight resident decoders, one per XCD (csrc/ar_xcd.hip; csrc/ar_xcm.hip from 69
        # utterances in flight: 16 slots per XCD on the matrix cores).  The launch IS the decode
        # loop, so its duration comes from the HIP events around it (vqcpc_vocoder_last_timing) and `achieved` prices every
        # sample of the launch with SURVEY 8d's 6 782 976 FLOP.
        steps = max(int(n_steps), 1)                 # steps of the longest XCD (more utterances than slots run back to back)
        per_utt = int(samples_per_utt) or steps
        launch_us = step_us * steps
        flop = FLOP_PER_SAMPLE * n_utt * per_utt
        step_tflops = flop / (launch_us * 1e-6) / 1e12
        slots = voc.last_slots()
        four = path == 2 and slots > 16
        return {"bound": (xcd_bound(slots) if path == 2 else BOUND[path]),
                "bound_detail": (("the serial chain of a sample step (cell update -> h_t exchange -> fc1 -> a_t exchange -> fc2 -> candidate exchange -> "
                                  "x_t: ~2.5 of 3.7 us) with the fp32 matrix pipe busy 1.2 us per SIMD and step (336 v_mfma_f32_4x4x1_16B_f32)")
                                 if four else
                                 ("fp32 vector issue + three in-XCD exchange latencies per sample step; with one or two slots per XCD the kernel "
                                  "issues no MFMA instruction and is priced against the fp32 vector peak (= the fp32 MFMA peak, 157.3 TFLOP/s)"))
                                if path == 2 else "fp32 MFMA issue (4.7 us of a 10.3 us step) + three in-XCD exchange latencies per sample step",
                "kernel": ("ar_xcd_kernel (ONE launch per call: a resident, weight-stationary decoder per XCD -- W_hh in VGPRs, fc1 / fc2 / "
                           "embedding table in LDS, h_t / a_t / candidates exchanged through the XCD's own L2; the fp32 fma chains of the "
                           + ("four slots of an XCD as v_mfma_f32_4x4x1_16B_f32 (one instruction per term for 4 rows x 4 slots), "
                              if four else "slots as v_fmac_f32_dpp, ") +
                           "bit-identical to the MFMA kernels of the other paths)") if path == 2 else
                          ("ar_xcm_kernel (ONE launch per call: a resident, weight-stationary decoder per XCD for 16 decode slots -- "
                           "[W_hh; W_fc1] h_t as six v_mfma_f32_16x16x4_f32 tiles per workgroup with the A fragments pinned in VGPRs, "
                           "fc2 fragments in LDS, h_t / a_t / candidates exchanged through the XCD's own L2)"),
                "achieved": step_tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": step_tflops / FP32_PEAK_TFLOPS,
                "frac_executed": step_tflops * exec_frac / FP32_PEAK_TFLOPS,
                "traffic": None, "flop_per_launch": flop, "avg_launch_us": launch_us, "utterances_per_launch": n_utt,
                "samples_per_launch": n_utt * per_utt,
                "algorithmic_bytes_per_launch": 4.0 * (8 * (GRU_MAC + 688128 + 229376 + 65536) + n_utt * per_utt) + 4.0 * n_utt * (per_utt // 160 + 1) * 2688,
                "how": "HIP events on the launch stream around the one launch of the call (vqcpc_vocoder_last_timing); flop = "
                       "6 782 976 per decoded sample x samples of the launch; algorithmic bytes = one copy of the recurrent weights "
                       "per XCD + the conditioning rows + the waveform",
                "launches_per_sample": 1.0 / (n_utt * per_utt),
                "decode_step": {"us": step_us, "tflops": step_tflops, "frac": step_tflops / FP32_PEAK_TFLOPS,
                                "frac_executed": step_tflops * exec_frac / FP32_PEAK_TFLOPS, "flop": flop / steps,
                                "how": "the launch's duration / steps of the longest XCD; frac_executed leaves out the 688 128 MACs per "
                                       "sample of the embedding half of W_ih, which is a table lookup on every path"}}
    gru_us, fc1_us, fc2_us, per_launch, kind = voc.kernel_times(2000)
    per_launch = min(int(per_launch), n_utt)                 # utterances one launch covers (one tile group)
    names = {0: "ar_gru_kernel<14,1> (one tile)", 1: "ar_gru_kernel<14,2> (two tiles per workgroup)",
             2: "ar_gru_big_kernel<14> (LDS-staged state, full 16-row gate tiles)",
             4: "ar_gru_kernel<14,.,.,fused> (ONE launch: fc2 + draw of sample t-1 in front of the GRU step of sample t)",
             5: "ar_gru_big_kernel<14,fused> (ONE launch: fc2 + draw of sample t-1 in front of the large-batch GRU step of sample t)"}
    fused = int(kind) in (4, 5)
    whole = False
    flop = 2.0 * (GRU_MAC + (65536 if fused else 0) + (229376 if whole else 0)) * per_launch
    achieved = flop / (gru_us * 1e-6) / 1e12
    # algorithmic bytes of one GRU-step launch: W_hh once + state in/out + gate inputs per utterance
    # (+ W_fc2 once and the fc1 outputs per utterance when fc2 rides in the same launch)
    alg_bytes = (4.0 * (GRU_MAC + per_launch * (2 * 896 + 2 * 3 * 896)) + (4.0 * (65536 + per_launch * 256) if fused else 0.0) +
                 (4.0 * 229376 if whole else 0.0))
    return {"bound": "mfma", "kernel": names.get(int(kind), "ar_gru") + ": W_hh h for all utterances + GRU cell update",
            "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_PEAK_TFLOPS,
            "traffic": None, "algorithmic_bytes_per_launch": alg_bytes, "flop_per_launch": flop,
            "avg_launch_us": gru_us, "utterances_per_launch": per_launch,
            "how": "HIP events on the launch stream around 2000 back-to-back launches, each on the next sample step so that "
                   "a fused launch waits for its own fc2 workgroups' candidates as in the decode loop (includes the ~1.5 us "
                   "dependent-launch boundary)",
            "launches_per_sample": 1 if whole else (2 if fused else 3),
            "other_kernels_us": {"ar_fc1_kernel" + (" (as its own launch, not on the fused path)" if whole else ""): fc1_us,
                                 "ar_fc2_kernel" + (" (as its own launch, not on the fused path)" if fused else ""): fc2_us},
            "decode_step": {"us": step_us, "tflops": step_tflops, "frac": step_tflops / FP32_PEAK_TFLOPS,
                            "frac_executed": step_tflops * exec_frac / FP32_PEAK_TFLOPS,
                            "flop": FLOP_PER_SAMPLE * n_utt,
                            "how": "HIP events around the whole decode loop / samples per utterance; frac_executed leaves out the "
                                   "688 128 MACs per sample of the embedding half of W_ih (a table lookup on every path)"}}


def attach_traffic(roof):
    """HBM-side bytes per launch of the dominant kernel come from separate `rocprofv3 --pmc` passes
    (tools/collect_traffic.py writes profiles/r04_pmc_traffic.json with the sha of the kernel sources it
    measured).  A file measured on other kernel sources or another batch is refused, not reported."""
    roof["traffic_source"] = "none"
    try:
        pmc = json.load(open(TRAFFIC_JSON))
    except (OSError, ValueError):
        return
    if pmc.get("kernel_source_sha") != kernel_source_sha():
        roof["traffic_source"] = f"stale: {os.path.relpath(TRAFFIC_JSON, ROOT)} was measured on other kernel sources; re-run tools/collect_traffic.py"
        return
    if roof.get("avg_launch_us") is None or pmc.get("utterances") != roof["utterances_per_launch"]:
        roof["traffic_source"] = "offline file covers another batch size"
        return
    if any((k in pmc.get("kernel", "")) != (k in roof["kernel"]) for k in ("ar_xcd", "ar_xcm")):
        roof["traffic_source"] = "offline file covers another kernel"
        return
    roof["traffic"] = pmc["traffic_bytes_per_launch"]
    roof["traffic_unit"] = "bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE; mostly Infinity-Cache hits)"
    ratio = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
    if "samples_per_launch" in pmc and roof.get("samples_per_launch"):
        # A one-launch-per-call kernel: the PMC pass measured a SHORTER call (8 000 samples per utterance) than the bench's.  Both
        # sides are given for the measured launch and per sample step; the ratio compares like with like (round 3 divided the short
        # launch's bytes by the long launch's algorithmic bytes: 0.70).
        steps_m = pmc["samples_per_launch"] / pmc["utterances"]
        roof["traffic_measured_on_samples_per_launch"] = pmc["samples_per_launch"]
        roof["traffic_bytes_per_sample_step"] = pmc["traffic_bytes_per_launch"] / steps_m
        roof["algorithmic_bytes_of_the_measured_launch"] = pmc["algorithmic_bytes_per_launch"]
        roof["algorithmic_bytes_per_sample_step_of_the_measured_launch"] = pmc["algorithmic_bytes_per_launch"] / steps_m
        ratio = pmc["traffic_bytes_per_launch"] / pmc["algorithmic_bytes_per_launch"]
    roof["traffic_source"] = (f"offline: tools/collect_traffic.py -> {os.path.relpath(TRAFFIC_JSON, ROOT)} "
                              f"(kernel sources {pmc['kernel_source_sha']}, {pmc.get('launch_mode', '?')} launches, "
                              f"kernel {pmc.get('kernel', '?')})")
    roof["traffic_over_algorithmic"] = ratio


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utterances-per-gpu", type=int, default=32)
    ap.add_argument("--frames", type=int, default=200, help="mel frames per utterance (200 = 2 s = 32 000 samples)")
    ap.add_argument("--manifest", type=int, default=512, help="synthetic ragged manifest of this many utterances (0 = skip)")
    ap.add_argument("--manifest-batch", type=int, default=256, help="decode slots of the manifest workload")
    ap.add_argument("--workload", default="shard", choices=("shard", "manifest"),
                    help="shard: configs[3]'s per-GPU batch (the contract's default); manifest: configs[4], the ragged manifest LPT-sharded "
                         "over the ranks with the ragged gather")
    ap.add_argument("--force-gather", action="store_true", help="manifest workload: run the gather's collectives at world size 1 too")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--selftest-spawn", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ       # under torch.distributed.run
    if not launched and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if launched and args.selftest_spawn:
        return selftest_rank(args)
    if args.workload == "manifest" and args.force_gather and not launched and args.gpus == 1:
        sys.exit(spawn_ranks(args, sys.argv[1:], check_devices=True, nproc=1))      # a world of one RCCL rank

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; pass the same N to both "
              "(or run plain `python bench.py --gpus N`, which starts its own ranks)", file=sys.stderr)
        sys.exit(2)
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.device_count() <= local:
            print(f"bench.py: rank {rank} has no GPU {local} on this node", file=sys.stderr)
            sys.exit(2)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if launched:
        # RCCL prints a version banner on stdout when its communicator comes up (at the first collective):
        # send file descriptor 1 to stderr until then, so that stdout carries the one JSON line only
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)     # RCCL
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    enc, voc = build_models(dev)
    if os.environ.get("VQCPC_BENCH_NO_GRAPH"):           # eager launches of the same kernels (debugging)
        voc.set_option("use_graph", 0)
    if args.workload == "manifest":
        from vectorquantizedcpc_amd import driver
        frames, spk_all = synthetic_manifest(args.manifest)
        mels_all = [synth.mel(f"bench/man{i % 8}", 1, max(frames))[0][:, :f].contiguous().to(dev) for i, f in enumerate(frames)]

        def decode_fn(ids, mels, speakers):
            if not ids:
                return torch.zeros(0, 0, device=dev)
            wavs = driver.convert_utterances(enc, voc, mels, speakers, seed=synth.SEED, utt_ids=list(ids), max_batch=64,
                                             max_pad_frac=0.15, slots=args.manifest_batch)
            out = torch.zeros(len(wavs), max(int(w.numel()) for w in wavs), device=dev)
            for k, w in enumerate(wavs):
                out[k, : w.numel()] = w
            return out
        result = manifest_sharded(args, rank, world, dev, mels_all, spk_all, decode_fn, voc.check, launched)
        if rank == 0:
            result["decode_slots_rank0"] = voc.last_slots()
            result["decode_path_rank0"] = voc.last_path()
            result["workspace_peak_bytes_rank0"] = voc.workspace_bytes()
            print(json.dumps(result), flush=True)
        if launched:
            dist.destroy_process_group()
        return
    Bp, T = args.utterances_per_gpu, args.frames
    n_total = Bp * world
    ids = shard.partition_contiguous(n_total, world)[rank]
    mel = torch.cat([synth.mel(f"bench/utt{i}", 1, T) for i in ids]).to(dev)      # resident in HBM
    spk = torch.tensor([i % 102 for i in ids], device=dev)
    L = 160 * (T // 2) * 2
    state = {}

    def step():
        idx = enc.encode_indices(mel)                         # convert.py:76 (context discarded)
        wav = voc.generate(idx, spk, seed=synth.SEED, utt_base=ids[0], async_=True)     # no synchronisation inside the timed step
        if launched:
            out = [torch.empty_like(wav) for _ in range(world)] if rank == 0 else None
            dist.gather(wav, out, dst=0)                      # the one exchange step (RCCL over xGMI)
            state["gathered"] = out
        state["wav"] = wav

    log(f"models built; timing {args.steps} step(s) of {Bp} utterances x {L} samples on {world} GPU(s)")
    dt = timed(step, args.steps, args.warmup, dev, world)
    voc.check()                                               # an in-kernel hand-off that gave up anywhere in the timed steps is an error
    log(f"timed region done: {dt:.3f} s")
    samples = n_total * L * args.steps
    value = samples / dt
    loop_ms, n_loop = voc.last_timing()                       # HIP events around the last decode loop
    step_us = loop_ms * 1e3 / max(n_loop, 1)
    roof = gru_roofline(voc, Bp, step_us, n_loop, L)
    attach_traffic(roof)

    result = {
        "metric": "audio samples/sec (WaveRNN-style decode, convert.py path: encode + generate)",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3] per-GPU shard: {Bp} utterances x {T} mel frames "
                               f"-> Encoder.encode indices -> Vocoder.generate {L} samples each",
                   "utterances_per_gpu": Bp, "samples_per_utterance": L, "weights": "random-init (seed 13)",
                   "parallelism": f"utterance-sharded x{world}, one RCCL gather"},
        "realtime_factor_16k": value / 16000.0,
        "realtime_factor_16k_per_gpu": value / 16000.0 / world,
        "roofline": roof,
    }
    if launched:
        # the exchange step alone: every rank's (Bp, L) fp32 waveforms to rank 0, max over ranks of 5 repetitions
        wav = state["wav"]
        out = [torch.empty_like(wav) for _ in range(world)] if rank == 0 else None
        dist.gather(wav, out, dst=0)
        gt = timed(lambda: dist.gather(wav, out, dst=0), 5, 1, dev, world) / 5
        result["rccl_ranks"] = dist.get_world_size()
        result["backend"] = dist.get_backend()
        result["gather"] = {"bytes_per_rank": wav.numel() * 4, "bytes_total": wav.numel() * 4 * world, "ms": gt * 1e3,
                            "how": "dist.gather of the step's waveforms to rank 0 alone, barrier-bracketed, mean of 5"}

    solo = rank == 0 and world == 1
    if solo and not args.no_extras:
        # BASELINE configs[1]: encoder conv+VQ forward, batch 64 x 128 frames; configs[0]'s 1 x 200 next to it
        m2 = synth.mel("bench/c2", 64, 128).to(dev)
        enc.encode_indices(m2)
        de, _ = wall(lambda: enc.encode_indices(m2), dev, 20)
        fps = 64 * 64 / de
        alg_bytes = 64 * 80 * 128 * 4 + 5132544 + 4096 * (64 * 4 + 8)
        m1 = synth.mel("bench/c1", 1, 200).to(dev)
        enc.encode_indices(m1)
        d1, _ = wall(lambda: enc.encode_indices(m1), dev, 50)
        enc.encode(m1)
        d1c, _ = wall(lambda: enc.encode(m1), dev, 20)
        enc.encode(m2)
        d2c, _ = wall(lambda: enc.encode(m2), dev, 20)
        result["encoder"] = {"workload": "BASELINE configs[1]: 64 x 80 x 128 mel -> 4096 code frames",
                             "frames_per_s": fps, "ms": de * 1e3,
                             "tflops": FLOP_PER_FRAME * fps / 1e12,
                             "frac_fp32_peak": FLOP_PER_FRAME * fps / 1e12 / FP32_PEAK_TFLOPS,
                             "algorithmic_GBps": alg_bytes / de / 1e9,
                             "frac_hbm_peak": alg_bytes / de / 1e9 / HBM_PEAK_GBS,
                             "c1_1x200_ms": d1 * 1e3, "c1_1x200_with_context_ms": d1c * 1e3,
                             "c2_with_context_ms": d2c * 1e3}
        # BASELINE configs[2]: one utterance of 32 000 samples
        z1 = synth.randint("bench/c3", (1, 100), 512).to(dev)
        s1 = torch.zeros(1, dtype=torch.long, device=dev)
        voc.generate(z1, s1, seed=synth.SEED, utt_base=0, max_steps=2000)
        d1, _ = wall(lambda: voc.generate(z1, s1, seed=synth.SEED, utt_base=0), dev)
        result["single_utterance"] = {"workload": "BASELINE configs[2]: 1 utterance x 32 000 samples",
                                      "samples_per_s": 32000 / d1, "realtime_factor_16k": 2.0 / d1,
                                      "us_per_sample": d1 / 32000 * 1e6}
        # BASELINE configs[3] whole (256 utterances) on ONE GPU: it fits, so this is the single-GPU ceiling
        n256 = 256
        mel256 = torch.cat([synth.mel(f"bench/utt{i % 64}", 1, T) for i in range(n256)]).to(dev)
        spk256 = (torch.arange(n256) % 102).to(dev)

        def step256():
            return voc.generate(enc.encode_indices(mel256), spk256, seed=synth.SEED, utt_base=0)
        step256()
        d256, _ = wall(step256, dev)
        ms256, n256_loop = voc.last_timing()
        r256 = gru_roofline(voc, n256, ms256 * 1e3 / max(n256_loop, 1), n256_loop, L)
        r256["traffic_source"] = "none"
        how256 = {3: "(continuous batching through the 128 decode slots of the matrix-core per-XCD decoders)",
                  2: "(continuous batching through the 32 decode slots of the per-XCD decoders)"}.get(voc.last_path(), "(two tile groups of 128 on two streams)")
        result["one_gpu_256"] = {"workload": f"BASELINE configs[3] unsharded: {n256} utterances x {L} samples on one GPU " + how256,
                                 "samples_per_s": n256 * L / d256, "realtime_factor_16k": n256 * L / 16000.0 / d256,
                                 "ms_per_step": d256 * 1e3, "roofline": r256}
        del mel256
        # teacher-forced Vocoder.forward at the reference's training shape (vocoder.py:51-66)
        melf = synth.mel("fwd/mel", 32, 32).to(dev)
        idxf = enc.encode_indices(melf)
        xf = synth.randint("fwd/x", (32, 5119), 256).to(dev)
        spkf = (torch.arange(32) % 102).to(dev)
        voc(xf, idxf, spkf)
        dtf, _ = wall(lambda: voc(xf, idxf, spkf), dev, 3)
        result["teacher_forced"] = {"workload": "Vocoder.forward (32, 5119) -> (32, 5119, 256) energies (vocoder.py:62 shape)",
                                    "ms": dtf * 1e3, "samples_per_s": 32 * 5119 / dtf}
    if solo and args.manifest > 0:
        log(f"manifest workload: {args.manifest} utterances")
        result["manifest"] = run_manifest(enc, voc, dev, args.manifest, args.manifest_batch)
    if solo and args.manifest > 0 and not args.no_extras:
        log(f"convert end to end: {args.manifest} wav files")
        result["convert_e2e"] = run_convert_e2e(enc, voc, dev, args.manifest, args.manifest_batch)
    if solo and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(Bp)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
