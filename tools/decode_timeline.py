#!/usr/bin/env python3
"""Where one decode step goes: wall-clock stamps of workgroup (0, 0) of the three per-sample kernels.

Needs the debug build (stamps are compiled out of the shipped library):
    for f in encoder vocoder melfront loudness; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 \
        -ffp-contract=off -DVQCPC_AR_STAMPS -c vectorquantizedcpc_amd/csrc/$f.hip -o build/stamps/$f.o; done
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libvqcpc_hip.so build/stamps/*.o
    python tools/decode_timeline.py [utterances]
Prints the mean, over the 160 steps of the last graph replay, of each segment in microseconds
(the stamp clock ticks every 10 ns).
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
voc = V.Vocoder(V.ConfVocoder())
voc.load_state_dict(synth.vocoder_state_dict())
voc = voc.cuda().eval()
voc.set_option("xcd", 0)                   # the timeline is of the launch-per-step kernels
voc.set_option("fuse_fc2", 0)              # ... one launch per kernel (the stamps are per kernel)
z = synth.randint("timeline", (B, 4), 512).cuda()                     # 4 codes -> 1280 samples = 8 replays of 160
spk = torch.arange(B, device="cuda") % 102
voc.generate(z, spk, seed=13)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (160 * 3 * 6))()
assert _lib.load().vqcpc_debug_ar_stamps(buf) == 0
s = np.array(buf, dtype=np.int64).reshape(160, 3, 6) * 0.01   # us
names = ["gru", "fc1", "fc2"]
rows = []
for k in range(3):
    rows.append((f"{names[k]}: entry -> MFMA results in LDS", s[5:155, k, 1] - s[5:155, k, 0]))
    if k == 0:
        rows.append(("gru: entry -> gate wave knows its slot and step (1st load level)", s[5:155, 0, 4] - s[5:155, 0, 0]))
        rows.append(("gru: entry -> gate wave has its operands (2nd load level)", s[5:155, 0, 2] - s[5:155, 0, 0]))
    rows.append((f"{names[k]}: MFMA done -> last store issued", s[5:155, k, 3] - s[5:155, k, 1]))
    nxt = s[5:155, k + 1, 0] if k < 2 else s[6:156, 0, 0]
    rows.append((f"{names[k]} end -> {names[(k + 1) % 3]} entry (boundary seen by workgroup 0)", nxt - s[5:155, k, 3]))
rows.append(("whole step (gru entry -> next gru entry)", s[6:156, 0, 0] - s[5:155, 0, 0]))
print(f"{B} utterances, workgroup (0, 0), mean / min / max over 150 steps, microseconds")
for name, d in rows:
    print(f"{name:62s} {d.mean():6.2f} {d.min():6.2f} {d.max():6.2f}")
