#!/usr/bin/env python3
"""HBM-side traffic per launch of the dominant decode kernel, from rocprofv3 PMC counters (GPU box only).

    python3 tools/collect_traffic.py [--utterances 32] [--mode xcd|xcm|graph16|eager]

Runs separate `rocprofv3 --pmc` passes (FETCH_SIZE; WRITE_SIZE; TCC_HIT_sum TCC_MISS_sum -- they do not fit one pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots") over a short decode of `--utterances` concurrent utterances and writes
profiles/r04_pmc_traffic.json: per-launch averages of the GRU-step kernel, the gfx950 FETCH_SIZE x2 correction
(MI355X_MICROARCH.md "HBM"), and `kernel_source_sha` = the hash bench.py checks before it reports `roofline.traffic`
(a file measured on other kernel sources is refused there).

Launch mode: `xcd` (default) = the per-XCD resident decoders, ONE launch per call (csrc/ar_xcd.hip), measured on a call of
`--codes` codes per utterance (default 25 = 8 000 samples); `xcm` = their matrix-core form (csrc/ar_xcm.hip, 16 slots per XCD), measured the same way; `graph16` replays the per-sample kernels from a hipGraph of 16 steps (the shipped path, shorter replay);
`eager` launches the same kernels one by one.  The shipped 160-step replay cannot be profiled with --pmc on ROCm 7.2:
rocprofiler-sdk faults in its packet interceptor when the HSA intercept queue overflows (DESIGN.md "Measurement").
The parent process never touches the GPU; each pass is a fresh child under rocprofv3.
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


def target(mode, n_utt, codes):
    import torch
    import vectorquantizedcpc_amd as V
    from vectorquantizedcpc_amd import synth
    voc = V.Vocoder(V.ConfVocoder())
    voc.load_state_dict(synth.vocoder_state_dict())
    voc = voc.cuda().eval()
    if mode == "xcd":
        voc.set_option("xcd", 1)
    elif mode == "xcm":
        voc.set_option("xcm", 1)
    elif mode == "eager":
        voc.set_option("xcd", 0)
        voc.set_option("use_graph", 0)
    else:
        voc.set_option("xcd", 0)
        voc.set_option("steps_per_graph", int(mode[len("graph"):] or 160))
    z = synth.randint("traffic/z", (n_utt, codes), 512).cuda()
    spk = (torch.arange(n_utt) % 102).cuda()
    wav = voc.generate(z, spk, seed=13, utt_base=0)
    voc.check()
    print(f"[traffic target] {mode}: {n_utt} x {wav.shape[1]} samples done", file=sys.stderr, flush=True)


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
    ap.add_argument("--utterances", type=int, default=32)
    ap.add_argument("--codes", type=int, default=0, help="codes per utterance (default: 25 = 8 000 samples for xcd, 1 = 320 samples = plenty of launches otherwise)")
    ap.add_argument("--mode", default="xcd")
    ap.add_argument("--target", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_pmc_traffic.json"))
    args = ap.parse_args()
    if args.codes <= 0:
        args.codes = 25 if args.mode in ("xcd", "xcm") else 1
    if args.target:
        return target(args.mode, args.utterances, args.codes)

    import bench                                   # kernel_source_sha(): imports torch, makes no GPU call
    scratch = os.path.join("/tmp", "vqcpc_traffic")
    os.makedirs(scratch, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    for name, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("tcc", ["TCC_HIT_sum", "TCC_MISS_sum"])):
        out_dir = os.path.join(scratch, f"{args.mode}_{args.utterances}_{name}")
        cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", out_dir, "--",
               sys.executable, os.path.abspath(__file__), "--target", "--mode", args.mode,
               "--utterances", str(args.utterances), "--codes", str(args.codes)]
        print("[collect_traffic]", " ".join(cmd), file=sys.stderr, flush=True)
        log = open(os.path.join(scratch, f"{args.mode}_{args.utterances}_{name}.log"), "w")
        rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=600).returncode
        if rc != 0:
            print(f"[collect_traffic] pass {name} failed with status {rc}: see {log.name}", file=sys.stderr)
            return 1
        res.update(averages(out_dir, {"xcd": "ar_xcd", "xcm": "ar_xcm"}.get(args.mode, "ar_gru")))
    kernels = sorted({k for k, _ in res})
    if not kernels:
        print("[collect_traffic] no dispatch of the decode kernel in the counter files", file=sys.stderr)
        return 1
    kern = max(kernels, key=lambda k: res.get((k, "FETCH_SIZE"), (0, 0))[0])
    fetch_kb, write_kb = res[(kern, "FETCH_SIZE")][1], res[(kern, "WRITE_SIZE")][1]
    hit, miss = res[(kern, "TCC_HIT_sum")][1], res[(kern, "TCC_MISS_sum")][1]
    traffic = (2.0 * fetch_kb + write_kb) * 1024.0
    alg = 4.0 * (2408448 + args.utterances * (2 * 896 + 2 * 3 * 896))
    steps = 320 * args.codes
    one_launch = args.mode in ("xcd", "xcm")
    if one_launch:                        # one launch per call: a copy of the recurrent weights per XCD + conditioning rows + waveform
        alg = 4.0 * (8 * (2408448 + 688128 + 229376 + 65536) + args.utterances * steps) + 4.0 * args.utterances * (steps // 160 + 1) * 2688
    elif ", 1>" in kern or ", 2>" in kern:   # fused launch: fc2 + draw of the previous sample rides along (W_fc2 once, fc1 outputs per utterance)
        alg += 4.0 * (65536 + args.utterances * 256)
    if not one_launch and ", 2>" in kern:                    # ... and fc1 (W_fc1 once; the state it reads is the one the GRU reads)
        alg += 4.0 * 229376
    out = {"kernel": kern, "utterances": args.utterances, "launch_mode": args.mode,
           "samples_per_launch": args.utterances * steps if one_launch else args.utterances,
           "dispatches_averaged": res[(kern, "FETCH_SIZE")][0],
           "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
           "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "l2_hit_rate": hit / max(hit + miss, 1.0),
           "tcc_miss_bytes_128B_lines": miss * 128.0,
           "correction": "gfx950: FETCH_SIZE counts a 128-B request as 64 B for wide coalesced reads (MI355X_MICROARCH.md, HBM) "
                         "-> x2; WRITE_SIZE exact",
           "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
           "traffic_over_algorithmic": traffic / alg,
           "kernel_source_sha": bench.kernel_source_sha(),
           "command": "python3 tools/collect_traffic.py --utterances %d --mode %s" % (args.utterances, args.mode)}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
