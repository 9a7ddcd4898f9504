#!/usr/bin/env python3
"""bench.py -- VQ-CPC inference hot path on MI355X: audio samples/s of the WaveRNN-style decode
(+ encoder frames/s), one process per GPU.

A "step" = one pass of the convert.py path (convert.py:72-77) over one per-GPU batch of synthetic
utterances that are already resident in HBM: mel (B, 80, 200) -> Encoder.encode indices ->
Vocoder.generate (B x 32 000 samples).  The per-GPU batch is BASELINE.json configs[3]'s shard
(256 utterances / 8 GPUs = 32, 2 s each), the same at every N (weak scaling); for N > 1 the
waveforms are gathered on rank 0 over RCCL inside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra keys next to the contract's: `roofline` (decode step vs the
fp32 MFMA peak), `cpu_baseline` (the PyTorch-CPU port of the same path on this box's host
cores, bounded sample), `encoder` (BASELINE configs[1], 64 x 128 frames) and `single_utterance`
(configs[2], 1 x 32 000 samples).
"""
import argparse
import json
import os
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
FLOP_PER_FRAME = 2555904        # encoder, per output frame without the LSTM


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


def host_cores():
    """Cores this process may actually use (cgroup / affinity aware), capped at the GPU box's share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(n_utt, budget_s=12.0):
    """PyTorch-CPU port of the same decode loop on this box's host cores, bounded sample."""
    from oracle import torch_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    tv = torch_ref.TorchVocoder(synth.vocoder_state_dict())
    z = synth.randint("bench/codes", (n_utt, 100), 512)
    spk = torch.arange(n_utt) % 102
    noise = torch_ref.make_noise(n_utt, 40, synth.SEED)
    tv.generate(z, spk, seed=synth.SEED, n_steps=10, noise=noise[:, :10])   # warm-up (thread pool, MKL)
    t0 = time.perf_counter()
    tv.generate(z, spk, seed=synth.SEED, n_steps=40, noise=noise)    # calibration
    per_step = (time.perf_counter() - t0) / 40
    n_steps = int(max(80, min(3200, budget_s / max(per_step, 1e-6))))
    log(f"cpu baseline: {cores} threads, ~{per_step * 1e3:.2f} ms/step, timing {n_steps} steps")
    noise = torch_ref.make_noise(n_utt, n_steps, synth.SEED)          # RNG of the protocol: not timed
    reps, dt = 0, 0.0
    while dt < budget_s and reps < 8:                                 # ~10-15 s of CPU work in all
        t0 = time.perf_counter()
        tv.generate(z, spk, seed=synth.SEED, n_steps=n_steps, noise=noise)
        dt += time.perf_counter() - t0
        reps += 1
    n_steps *= reps
    esd = synth.encoder_state_dict()
    mel = synth.mel("bench/c2", 64, 128)
    torch_ref.encoder_encode(esd, mel, want_c=False)
    t1 = time.perf_counter()
    for _ in range(5):
        torch_ref.encoder_encode(esd, mel, want_c=False)
    de = (time.perf_counter() - t1) / 5
    return {"value": n_utt * n_steps / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n_utt} concurrent utterances x {n_steps} decode steps of the same workload "
                      f"(PyTorch-CPU ops, {cores} threads; prenet included), {dt:.1f} s",
            "encoder_frames_per_s": 64 * 64 / de}


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
    return {"workload": f"synthetic manifest (configs[4] stand-in): {n_utt} utterances, log-normal 1-10 s, "
                        f"encoder in length buckets, decode by continuous batching over {max_batch} slots",
            "utterances": n_utt, "audio_seconds": samples / 16000.0, "wall_s": dt,
            "samples_per_s": samples / dt, "realtime_factor_16k": samples / 16000.0 / dt}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utterances-per-gpu", type=int, default=32)
    ap.add_argument("--frames", type=int, default=200, help="mel frames per utterance (200 = 2 s = 32 000 samples)")
    ap.add_argument("--manifest", type=int, default=0, help="also run a synthetic ragged manifest of this many utterances")
    ap.add_argument("--manifest-batch", type=int, default=128, help="decode slots of the manifest workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ       # under torch.distributed.run
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N"

    enc, voc = build_models(dev)
    if os.environ.get("VQCPC_BENCH_NO_GRAPH"):           # counter collection (rocprofv3 --pmc) needs plain launches
        voc.set_option("use_graph", 0)
    Bp, T = args.utterances_per_gpu, args.frames
    n_total = Bp * world
    ids = shard.partition_contiguous(n_total, world)[rank]
    mel = torch.cat([synth.mel(f"bench/utt{i}", 1, T) for i in ids]).to(dev)      # resident in HBM
    spk = torch.tensor([i % 102 for i in ids], device=dev)
    L = 160 * (T // 2) * 2
    state = {}

    def step():
        idx = enc.encode_indices(mel)                         # convert.py:76 (context discarded)
        wav = voc.generate(idx, spk, seed=synth.SEED, utt_base=ids[0])
        if launched:
            out = [torch.empty_like(wav) for _ in range(world)] if rank == 0 else None
            dist.gather(wav, out, dst=0)                      # the one exchange step (RCCL over xGMI)
            state["gathered"] = out
        state["wav"] = wav

    log(f"models built; timing {args.steps} step(s) of {Bp} utterances x {L} samples on {world} GPU(s)")
    dt = timed(step, args.steps, args.warmup, dev, world)
    log(f"timed region done: {dt:.3f} s")
    samples = n_total * L * args.steps
    value = samples / dt
    loop_ms, n_loop = voc.last_timing()                       # HIP events around the last decode loop
    step_us = loop_ms * 1e3 / max(n_loop, 1)
    gru_us, fc1_us, fc2_us, per_launch = voc.kernel_times(2000)   # HIP events around back-to-back launches
    per_launch = min(int(per_launch), Bp)                     # utterances one launch covers (tile group)
    # dominant kernel = the GRU step: algorithmic FLOP per launch = W_hh mat-vec for every utterance it covers
    gru_flop = 2.0 * 2408448 * per_launch
    achieved = gru_flop / (gru_us * 1e-6) / 1e12
    step_tflops = FLOP_PER_SAMPLE * Bp / (step_us * 1e-6) / 1e12
    # HBM-side bytes per launch of that kernel: collected offline with rocprofv3 --pmc (separate
    # passes, gfx950 FETCH_SIZE correction applied) -- profiles/r01_pmc_traffic.json, same batch only
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if pmc.get("utterances") == per_launch:
            traffic = pmc["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    # algorithmic bytes of one GRU-step launch: W_hh once + state in/out + gate inputs per utterance
    gru_bytes = 4.0 * (2408448 + per_launch * (2 * 896 + 2 * 3 * 896))

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
        "roofline": {"bound": "mfma", "kernel": "ar_gru_kernel<14> (GRU step: W_hh h for all utterances + cell update)",
                     "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_unit": "bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE; mostly Infinity-Cache hits)",
                     "algorithmic_bytes_per_launch": gru_bytes,
                     "flop_per_launch": gru_flop, "avg_launch_us": gru_us, "utterances_per_launch": per_launch,
                     "how": "HIP events on the launch stream around 2000 back-to-back launches (includes the "
                            "~1.5 us dependent-launch boundary)",
                     "other_kernels_us": {"ar_fc1_kernel": fc1_us, "ar_fc2_kernel": fc2_us},
                     "decode_step": {"us": step_us, "tflops": step_tflops, "frac": step_tflops / FP32_PEAK_TFLOPS,
                                     "flop": FLOP_PER_SAMPLE * Bp,
                                     "how": "HIP events around the whole decode loop / samples per utterance"}},
    }

    if rank == 0 and world == 1 and not args.no_extras:
        # BASELINE configs[1]: encoder conv+VQ forward, batch 64 x 128 frames
        m2 = synth.mel("bench/c2", 64, 128).to(dev)
        for _ in range(3):
            enc.encode_indices(m2)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            enc.encode_indices(m2)
        torch.cuda.synchronize(dev)
        de = (time.perf_counter() - t0) / reps
        fps = 64 * 64 / de
        alg_bytes = 64 * 80 * 128 * 4 + 5132544 + 4096 * (64 * 4 + 8)
        result["encoder"] = {"workload": "BASELINE configs[1]: 64 x 80 x 128 mel -> 4096 code frames",
                             "frames_per_s": fps, "ms": de * 1e3,
                             "tflops": FLOP_PER_FRAME * fps / 1e12,
                             "frac_fp32_peak": FLOP_PER_FRAME * fps / 1e12 / FP32_PEAK_TFLOPS,
                             "algorithmic_GBps": alg_bytes / de / 1e9,
                             "frac_hbm_peak": alg_bytes / de / 1e9 / HBM_PEAK_GBS}
        # BASELINE configs[2]: one utterance of 32 000 samples
        z1 = synth.randint("bench/c3", (1, 100), 512).to(dev)
        s1 = torch.zeros(1, dtype=torch.long, device=dev)
        voc.generate(z1, s1, seed=synth.SEED, utt_base=0, max_steps=2000)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        voc.generate(z1, s1, seed=synth.SEED, utt_base=0)
        torch.cuda.synchronize(dev)
        d1 = time.perf_counter() - t0
        result["single_utterance"] = {"workload": "BASELINE configs[2]: 1 utterance x 32 000 samples",
                                      "samples_per_s": 32000 / d1, "realtime_factor_16k": 2.0 / d1}
    if rank == 0 and world == 1 and args.manifest > 0:
        log(f"manifest workload: {args.manifest} utterances")
        result["manifest"] = run_manifest(enc, voc, dev, args.manifest, args.manifest_batch)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(Bp)
    if rank == 0:
        print(json.dumps(result))
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
