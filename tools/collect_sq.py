#!/usr/bin/env python3
"""Shader-sequencer counters of the resident decode kernels, from rocprofv3 PMC passes (GPU box only): what the in-kernel stamps
of ONE worker say about a sample step ("the matrix pipe works 5.1 of 10.3 us", "the step is vector-issue bound on two service
waves"), stated by the hardware counters for ALL workers.

    python3 tools/collect_sq.py [--mode xcd|xcm] [--utterances 32|128] [--out profiles/r04_pmc_sq_xcd32.json]

Separate `rocprofv3 --pmc` passes of at most 8 SQ counters each (MI355X_MICROARCH.md "rocprofv3 PMC slots"), each a fresh child
with the program itself behind `--`; only --pmc (no trace domains).  The decode is one call of --codes codes per utterance
(default 25 = 8 000 samples), i.e. ONE dispatch of the kernel.  Units (same guide, cycle-constants table): SQ_WAVE_CYCLES,
SQ_BUSY_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count QUAD-cycles (4 clocks) summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts
clocks summed over SIMDs... the derived lines below say which division was applied.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"],
    ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM"],
    ["SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU_FMA_F32", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"],
]


def target(mode, n_utt, codes):
    import torch
    import vectorquantizedcpc_amd as V
    from vectorquantizedcpc_amd import synth
    voc = V.Vocoder(V.ConfVocoder())
    voc.load_state_dict(synth.vocoder_state_dict())
    voc = voc.cuda().eval()
    voc.set_option("xcm" if mode == "xcm" else "xcd", 1)
    z = synth.randint("traffic/z", (n_utt, codes), 512).cuda()
    spk = (torch.arange(n_utt) % 102).cuda()
    wav = voc.generate(z, spk, seed=13, utt_base=0)
    ms, n = voc.last_timing()
    print(f"[sq target] {mode}: {n_utt} x {wav.shape[1]} samples, path {voc.last_path()}, decode loop {ms:.3f} ms = {ms * 1e3 / n:.3f} us per step",
          file=sys.stderr, flush=True)


def averages(out_dir, needle):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if needle in row["Kernel_Name"]:
                    a = acc[(row["Kernel_Name"], row["Counter_Name"])]
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=0)
    ap.add_argument("--codes", type=int, default=25)
    ap.add_argument("--mode", default="xcd", choices=("xcd", "xcm"))
    ap.add_argument("--target", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    if args.utterances <= 0:
        args.utterances = 32 if args.mode == "xcd" else 128
    if args.target:
        return target(args.mode, args.utterances, args.codes)
    out_path = args.out or os.path.join(ROOT, "profiles", f"r04_pmc_sq_{args.mode}{args.utterances}.json")

    import bench                                   # kernel_source_sha(): imports torch, makes no GPU call
    scratch = os.path.join("/tmp", "vqcpc_sq")
    os.makedirs(scratch, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    needle = "ar_xcd" if args.mode == "xcd" else "ar_xcm"
    res, us_per_step = {}, None
    for i, counters in enumerate(PASSES):
        out_dir = os.path.join(scratch, f"{args.mode}_{args.utterances}_p{i}")
        cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", out_dir, "--",
               sys.executable, os.path.abspath(__file__), "--target", "--mode", args.mode,
               "--utterances", str(args.utterances), "--codes", str(args.codes)]
        print("[collect_sq]", " ".join(cmd), file=sys.stderr, flush=True)
        logp = os.path.join(scratch, f"{args.mode}_{args.utterances}_p{i}.log")
        with open(logp, "w") as log:
            rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=600).returncode
        txt = open(logp).read()
        for line in txt.splitlines():
            if line.startswith("[sq target]") and "us per step" in line:
                us_per_step = float(line.rsplit("=", 1)[1].split()[0])
        if rc != 0:
            print(f"[collect_sq] pass {i} failed with status {rc}: {txt[-1500:]}", file=sys.stderr)
            continue
        res.update(averages(out_dir, needle))
    kernels = sorted({k for k, _ in res})
    if not kernels:
        print("[collect_sq] no dispatch of the decode kernel in the counter files", file=sys.stderr)
        return 1
    kern = kernels[0]
    c = {name: v for (k, name), (n, v) in res.items() if k == kern}
    steps = 320 * args.codes
    out = {"kernel": kern, "utterances": args.utterances, "mode": args.mode, "sample_steps": steps, "counters_per_launch": c,
           "us_per_step_profiled": us_per_step,
           "workgroups": 256, "waves_per_workgroup": 12, "simds": 1024,
           "kernel_source_sha": bench.kernel_source_sha(),
           "command": "python3 tools/collect_sq.py --mode %s --utterances %d" % (args.mode, args.utterances)}
    d = {}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        # quad-cycles summed over the 3072 waves of the launch; the kernel's own length in clocks = wave cycles x 4 / waves
        d["kernel_clocks_per_wave"] = wc * 4.0 / c.get("SQ_WAVES", 3072.0)
        d["clocks_per_sample_step"] = d["kernel_clocks_per_wave"] / steps
        if us_per_step:
            d["implied_clock_GHz"] = d["clocks_per_sample_step"] / (us_per_step * 1e3)
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                     "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS"):
            if name in c:
                d[name + "_over_WAVE_CYCLES"] = c[name] / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # clocks in which a SIMD's matrix pipe is busy, summed over the launch's SIMD-clocks: per SIMD = / 1024
            per_simd = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
            d["mfma_busy_clocks_per_simd"] = per_simd
            d["mfma_busy_over_kernel_clocks"] = per_simd / d["kernel_clocks_per_wave"]
            d["mfma_busy_clocks_per_simd_per_step"] = per_simd / steps
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_VALU_FMA_F32"):
        if name in c:
            d[name + "_per_wave_per_step"] = c[name] / c.get("SQ_WAVES", 3072.0) / steps
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in c:
        d["mfma_f32_mops_per_step"] = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] / steps
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_over_lds_active"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    out["derived"] = d
    out["units"] = ("SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES is clocks "
                    "summed over SIMDs (MI355X_MICROARCH.md, cycle constants); ratios of the first family are unit-free")
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
