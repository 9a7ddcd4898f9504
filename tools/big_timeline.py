#!/usr/bin/env python3
"""Where a tile goes inside the large-batch GRU launch: wall-clock stamps (10 ns ticks) of one workgroup at local step 100.

Needs the debug build (stamps are compiled out of the shipped library), see tools/decode_timeline.py:
    python tools/big_timeline.py [utterances] [two_groups]
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
voc = V.Vocoder(V.ConfVocoder())
voc.load_state_dict(synth.vocoder_state_dict())
voc = voc.cuda().eval()
voc.set_option("two_groups", int(sys.argv[2]) if len(sys.argv) > 2 else 0)
z = synth.randint("timeline", (B, 4), 512).cuda()
spk = torch.arange(B, device="cuda") % 102
voc.generate(z, spk, seed=13)
torch.cuda.synchronize()
ms, n = voc.last_timing()
buf = (C.c_ulonglong * (32 * 8))()
assert _lib.load().vqcpc_debug_big_stamps(buf) == 0
s = np.array(buf, dtype=np.int64).reshape(32, 8) * 0.01   # us
nt = int((s[:, 0] > 0).sum())
t0 = s[0, [0, 4]].min()
print(f"{B} utterances: {ms * 1e3 / n:.2f} us per sample step; workgroup 1, local step 100, {nt} tiles; us since its first stamp")
print("tile | MFMA wave: start  mfma_done  next_stored  barrier | cell wave: start  barrier  tile0_resolved  update_stored")
for i in range(nt):
    r = s[i] - t0
    print(f"{i:4d} | {r[0]:8.2f} {r[1]:8.2f} {r[2]:8.2f} {r[3]:8.2f} | {r[4]:8.2f} {r[5]:8.2f} {r[6]:8.2f} {r[7]:8.2f}")
