#!/usr/bin/env python3
"""Where one sample step of the pipelined per-XCD decoder (csrc/ar_xcp.hip) goes: wall-clock stamps (10 ns ticks) of worker 5
of XCD 0 over steps 256..383, per decode slot of the XCD.

Needs the debug build (stamps are compiled out of the shipped library): tools/build_stamps.sh, then
    python3 tools/xcp_timeline.py [utterances ...] [--lag N]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectorquantizedcpc_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "build", "stamps", "libvqcpc_hip.so")
import vectorquantizedcpc_amd as V  # noqa: E402
from vectorquantizedcpc_amd import synth  # noqa: E402

sys.argv0_flags = [a for a in sys.argv[1:] if a in ("--all", "--raw")]
args = [a for a in sys.argv[1:] if a not in ("--all", "--raw")]
lag = None
if "--lag" in args:
    i = args.index("--lag")
    lag = int(args[i + 1])
    del args[i:i + 2]
voc = V.Vocoder(V.ConfVocoder())
voc.load_state_dict(synth.vocoder_state_dict())
voc = voc.cuda().eval()
voc.set_option("xcd", 1)
voc.set_option("xcp", 1)
if lag is not None:
    voc.set_option("xcp_cell_lag", lag)
NAMES = ["wave 10 ready for the slot", "candidates in", "h_t published", "wave 10: its part of h_t swept", "h_t complete in LDS",
         "a_t published", "a_t gathered (fc2 wave)", "candidate published"]
for B in [int(a) for a in args] or [16, 32]:
    z = synth.randint("timeline", (B, 4), 512).cuda()
    spk = torch.zeros(B, dtype=torch.long, device="cuda")
    voc.generate(z, spk, seed=13)
    voc.check()
    assert voc.last_path() == 4, voc.last_path()
    ms, n = voc.last_timing()
    buf = (C.c_ulonglong * (128 * 32))()
    assert _lib.load().vqcpc_debug_xp_stamps(buf) == 0
    s = np.array(buf, dtype=np.int64).reshape(128, 4, 8) * 0.01     # us; [step][slot][stamp]
    bx = (B + 7) // 8
    pl = (C.c_uint * 4)()
    if _lib.load().vqcpc_debug_xp_polls(pl) == 0:
        print("  failed h polls of wave 10 per slot over 128 steps (cumulative over calls):", list(pl))
    print(f"pipelined per-XCD decoders, {B} utterance(s), {bx} slot(s) per XCD: {ms * 1e3 / n:.2f} us per sample step over the call; "
          f"worker 5 of XCD 0, mean / min / max over 112 steps, us")
    a, b = s[8:120], s[9:121]
    for sl in range(bx if "--all" in sys.argv0_flags else 1):
        print(f"  slot {sl}: step period {np.mean(b[:, sl, 2] - a[:, sl, 2]):.2f}; "
              f"offset of its h_t publish behind slot 0's: {np.mean(a[:, sl, 2] - a[:, 0, 2]):.2f}")
        seq = [(2, 0), (0, 3), (3, 4), (4, 5), (5, 6), (6, 7)]
        for i, j in seq:
            d = a[:, sl, j] - a[:, sl, i]
            print(f"    {NAMES[i]:34s} -> {NAMES[j]:34s} {d.mean():6.2f} {d.min():6.2f} {d.max():6.2f}")
        for i, j in [(7, 1)]:
            d = b[:, sl, j] - a[:, sl, i]
            print(f"    {NAMES[i]:34s} -> next step: {NAMES[j]:22s} {d.mean():6.2f} {d.min():6.2f} {d.max():6.2f}")
        d = b[:, sl, 2] - b[:, sl, 1]
        print(f"    {'candidates in':34s} -> {'h_t published':34s} {d.mean():6.2f} {d.min():6.2f} {d.max():6.2f}")
    if "--raw" in sys.argv0_flags:
        ev = []
        short = ["w10_ready", "cand_in", "h_pub", "w10_swept", "h_complete", "a_pub", "a_gath", "cand_pub"]
        for st in range(40, 44):
            for sl in range(bx):
                for k in range(0, 8):
                    ev.append((s[st, sl, k], f"t{st} s{sl} {short[k]}"))
        ev.sort()
        t0 = ev[0][0]
        for tm, name in ev:
            print(f"    {tm - t0:8.2f}  {name}")
    if "--raw" in sys.argv0_flags:
        wb = (C.c_ulonglong * (6 * 32 * 128))()
        if _lib.load().vqcpc_debug_xp_workers(wb) == 0:
            wk = np.array(wb, dtype=np.int64).reshape(6, 32, 128)[:, :, 8:120] * 0.01
            for e, nm in enumerate(["h_t published", "a_t published", "candidate published", "cell update starts", "cell wave 0 starts its pass over slot 1", "cell wave 0 ends its pass over slot 1"]):
                rel = wk[e] - wk[e].min(axis=0, keepdims=True)          # behind the first worker of that step
                print(f"  slot 0, {nm}: worker's mean / max lag behind the step's first worker, us")
                print("    " + " ".join(f"{rel[r].mean():.2f}/{rel[r].max():.2f}" for r in range(32)))
                print(f"    last worker of a step lags {rel.max(axis=0).mean():.2f} on average")
                if e == 0:
                    hw = (C.c_uint * 32)()
                    _lib.load().vqcpc_debug_xp_hwid(hw)
                    slow = int(np.argmax(rel.mean(axis=1)))
                    print(f"    slowest worker {slow}: HW_ID {hw[slow]:#x} (cu {(hw[slow] >> 8) & 15}, sh {(hw[slow] >> 12) & 1}, se {(hw[slow] >> 13) & 7}); all: " + " ".join(f"{hw[r]:#x}" for r in range(32)))
        vb = (C.c_ulonglong * (2 * 12 * 32 * 128))()
        if _lib.load().vqcpc_debug_xp_waves(vb) == 0:
            wv = np.array(vb, dtype=np.int64).reshape(2, 12, 32, 128)[:, :, :, 8:120] * 0.01
            base = wv[1][wv[1] > 0].reshape(-1)  # noqa
            ref = np.where(wv[1] > 0, wv[1], np.inf).min(axis=(0, 1), keepdims=True)       # first start of the pass over slot 1, any wave, any worker
            for e, nm in enumerate(["sweep of slot 1 done (waves 8..11)", "chain pass over slot 1 starts"]):
                print(f"  {nm}: mean lag behind the step's first pass start, per wave (rows) and worker (columns), us")
                for w in range(12):
                    if (wv[e, w] > 0).any():
                        print(f"    wave {w:2d}: " + " ".join(f"{(wv[e, w, r] - ref[0, 0]).mean():5.2f}" for r in range(32)))
