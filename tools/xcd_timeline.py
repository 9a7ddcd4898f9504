#!/usr/bin/env python3
"""Where one sample step of the per-XCD resident decoder (csrc/ar_xcd.hip) goes: wall-clock stamps (10 ns ticks) of
worker 5 of XCD 0 over steps 256..383.

Needs the debug build (stamps are compiled out of the shipped library):
    mkdir -p build/stamps && for f in encoder vocoder ar_xcd melfront loudness resample; do /opt/rocm/bin/hipcc -O3 -std=c++17 \
        -fPIC --offload-arch=gfx950 -ffp-contract=off -DVQCPC_XD_STAMPS -c vectorquantizedcpc_amd/csrc/$f.hip -o build/stamps/$f.o; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libvqcpc_hip.so build/stamps/*.o
    python3 tools/xcd_timeline.py [utterances ...]
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
for a in sys.argv[1:]:
    if a.startswith("--lib="):        # e.g. the stamped build with the exchange waits ablated (-DXD_ABLATE=7)
        _lib.LIB_PATH = os.path.abspath(a[6:])
import vectorquantizedcpc_amd as V  # noqa: E402
from vectorquantizedcpc_amd import synth  # noqa: E402

voc = V.Vocoder(V.ConfVocoder())
voc.load_state_dict(synth.vocoder_state_dict())
voc = voc.cuda().eval()
voc.set_option("xcd", 1)
for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 8, 16, 32]:
    z = synth.randint("timeline", (B, 4), 512).cuda()
    spk = torch.zeros(B, dtype=torch.long, device="cuda")
    voc.generate(z, spk, seed=13)
    voc.check()
    ms, n = voc.last_timing()
    buf = (C.c_ulonglong * (128 * 16))()
    assert _lib.load().vqcpc_debug_xd_stamps(buf) == 0
    s = np.array(buf, dtype=np.int64).reshape(128, 16) * 0.01     # us
    a, b = s[4:120], s[5:121]
    rows = [
        ("service wave 0: past barrier B -> own h_t published (gsum reads, cell update)", a[:, 2] - a[:, 0]),
        ("service wave 0: h published -> h_t of all 32 workers gathered (+ sample out, next state, noise)", a[:, 3] - a[:, 2]),
        ("barrier A", a[:, 4] - a[:, 3]),
        ("service wave 0 (four slots): barrier A -> last term of the fc1 chain issued", a[:, 14] - a[:, 4]),
        ("service wave 0 (four slots): barrier A -> result of the last term there", a[:, 15] - a[:, 4]),
        ("service wave 0: barrier A -> fc1 rows of its slots published", a[:, 5] - a[:, 4]),
        ("service wave 1: barrier A -> fc1 rows of its slots published", a[:, 10] - a[:, 4]),
        ("service wave 0: fc1 published -> W_hh rows 80..83 of the other wave's slots done", a[:, 7] - a[:, 5]),
        ("service wave 1: fc1 published -> W_hh rows 80..83 of the other wave's slots done", a[:, 11] - a[:, 10]),
        ("chain wave 0: barrier A -> a_t of its slot gathered (after its first W_hh chains)", a[:, 8] - a[:, 4]),
        ("chain wave 0: a_t gathered -> candidate published (fc2 + draw)", a[:, 9] - a[:, 8]),
        ("chain wave 0: candidate published -> all its W_hh chains done", a[:, 6] - a[:, 9]),
        ("service wave 0: W_hh rows done -> noise of the next step drawn", a[:, 13] - a[:, 7]),
        ("service wave 0: noise drawn -> x_t known (candidate sweep + argmax)", a[:, 1] - a[:, 13]),
        ("service wave 0: x_t known -> past barrier B", a[:, 12] - a[:, 1]),
        ("whole step", b[:, 0] - a[:, 0]),
    ]
    if B > 16:      # four slots per XCD: the matrix-pipe form -- wave 0 runs fc1 of all four slots, wave 1 W_hh rows 80..83, wave 11 draws the noise
        relabel = {"service wave 0: barrier A -> fc1 rows of its slots published": "service wave 0: barrier A -> a_t of the four slots published (fc1 chain + combine, bias, ReLU, stores)",
                   "service wave 1: barrier A -> fc1 rows of its slots published": "service wave 1: barrier A -> W_hh rows 80..83 of the four slots in LDS",
                   "service wave 0: fc1 published -> W_hh rows 80..83 of the other wave's slots done": "service wave 0: a_t published -> (nothing of its own; two stamps among the released chain waves' matrix instructions)",
                   "service wave 1: fc1 published -> W_hh rows 80..83 of the other wave's slots done": "service wave 1: rows in LDS -> (nothing of its own; two stamps)",
                   "service wave 0: W_hh rows done -> noise of the next step drawn": "service wave 0: -> candidate poll begins (no noise to draw: wave 11 does)",
                   "chain wave 0: barrier A -> a_t of its slot gathered (after its first W_hh chains)": "chain wave 0: barrier A -> a_t of slot 0 gathered (behind 10 of its chain's 14 groups)",
                   "chain wave 0: candidate published -> all its W_hh chains done": "chain wave 0: candidate published -> the rest of its chain done"}
        rows = [(relabel.get(n, n), d) for n, d in rows]
    print(f"per-XCD decoders, {B} utterance(s): {ms * 1e3 / n:.2f} us per sample step over the call; worker 5 of XCD 0, "
          f"mean / min / max over 116 steps, us")
    for name, d in rows:
        if abs(d).max() > 1e4:            # a stamp this configuration does not write (one slot per XCD: wave 1 only runs fc1)
            continue
        print(f"  {name:66s} {d.mean():6.2f} {d.min():6.2f} {d.max():6.2f}")
    if "--workers" in sys.argv:
        wb = (C.c_ulonglong * (6 * 32 * 128))()
        if _lib.load().vqcpc_debug_xd_workers(wb) == 0:
            wk = np.array(wb, dtype=np.int64).reshape(6, 32, 128)[:, :, 8:120] * 0.01
            for e, nm in enumerate(["h_t published (service wave 0)", "past barrier A", "a_t published", "candidate of slot 0 published (chain wave 0)",
                                    "x_t known", "past barrier B"]):
                rel = wk[e] - wk[e].min(axis=0, keepdims=True)
                print(f"  {nm}: every worker's mean lag behind the step's first worker, us; the last one lags {rel.max(axis=0).mean():.2f} on average")
                print("    " + " ".join(f"{rel[r].mean():.2f}" for r in range(32)))
            # the candidate exchange as the workers see it: last candidate published -> x_t known, per worker
            last_c = wk[3].max(axis=0, keepdims=True)
            d = wk[4] - last_c
            print(f"  last worker's candidate published -> x_t known: mean over workers {d.mean():.2f}, slowest worker {d.mean(axis=1).max():.2f}, fastest {d.mean(axis=1).min():.2f}")
            d = wk[1] - wk[0].max(axis=0, keepdims=True)
            print(f"  last worker's h_t published -> past barrier A: mean over workers {d.mean():.2f}, slowest worker {d.mean(axis=1).max():.2f}")
