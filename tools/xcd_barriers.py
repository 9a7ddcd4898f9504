#!/usr/bin/env python3
"""When each of the 12 waves of one workgroup of the per-XCD decoder (csrc/ar_xcd.hip) reaches and leaves the two barriers of a
sample step: worker 5 of XCD 0, steps 256..383 (100 MHz wall clock).  Who the barriers wait for.

Needs a debug build with -DVQCPC_XD_BARS (and only that: the other stamps perturb it), e.g.
    tools/build_stamps.sh "-DVQCPC_XD_BARS" && python3 tools/xcd_barriers.py [--lib=build/stamps/libvqcpc_hip.so] [utterances ...]
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
    if a.startswith("--lib="):
        _lib.LIB_PATH = os.path.abspath(a[6:])
import vectorquantizedcpc_amd as V  # noqa: E402
from vectorquantizedcpc_amd import synth  # noqa: E402

voc = V.Vocoder(V.ConfVocoder())
voc.load_state_dict(synth.vocoder_state_dict())
voc = voc.cuda().eval()
voc.set_option("xcd", 1)
for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 32]:
    z = synth.randint("timeline", (B, 4), 512).cuda()
    spk = torch.zeros(B, dtype=torch.long, device="cuda")
    voc.generate(z, spk, seed=13)
    voc.check()
    ms, n = voc.last_timing()
    buf = (C.c_ulonglong * (128 * 12 * 4))()
    assert _lib.load().vqcpc_debug_xd_bars(buf) == 0
    s = np.array(buf, dtype=np.int64).reshape(128, 12, 4)[4:120] * 0.01          # us
    t0 = s[:, :, 3].max(axis=1)                     # the step's origin: the last wave leaves barrier B of the step BEFORE
    step = t0[1:] - t0[:-1]
    s = s[1:] - t0[:-1, None, None]
    print(f"per-XCD decoders ({os.path.relpath(_lib.LIB_PATH, ROOT)}), {B} utterance(s): {ms * 1e3 / n:.2f} us per sample step over the call; "
          f"step by these stamps {step.mean():.2f}; us after the last wave left the previous step's barrier B, mean over {len(s)} steps")
    print("  wave  role                         arrives at A  leaves A   arrives at B  leaves B")
    for w in range(12):
        if B > 16:          # four slots per XCD: the chains run on the matrix pipe; fc2 on waves 2, 3, 6, 7; waves 4 and 8 wait for a_t to go out
            role = {0: "service + fc1", 1: "service + W_hh rows 80..83", 2: "chain + fc2 of slot 0", 3: "chain + fc2 of slot 1",
                    6: "chain + fc2 of slot 2", 7: "chain + fc2 of slot 3", 4: "chain, held behind fc1", 8: "chain, held behind fc1",
                    11: "chain + next step's noise"}.get(w, "chain")
        else:
            role = "service" if w < 2 else ("chain + fc2 of slot %d" % (w - 2) if w - 2 < max(1, B // 8) else "chain")
            if B <= 8 and w == 1:
                role = "fc1 (one slot)"
        m = s[:, w, :].mean(axis=0)
        print(f"  {w:4d}  {role:28s} {m[0]:10.2f} {m[1]:10.2f} {m[2]:12.2f} {m[3]:10.2f}")
