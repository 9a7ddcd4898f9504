#!/usr/bin/env python3
"""Where one sample step of the persistent single-utterance decoder goes: wall-clock stamps (10 ns ticks) of one
workgroup over steps 256..383.

Needs the debug build (stamps are compiled out of the shipped library):
    mkdir -p build/stamps && for f in encoder vocoder melfront loudness; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC \
        --offload-arch=gfx950 -ffp-contract=off -DVQCPC_PS_STAMPS -c vectorquantizedcpc_amd/csrc/$f.hip -o build/stamps/$f.o; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libvqcpc_hip.so build/stamps/*.o
    python3 tools/persist_timeline.py
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

voc = V.Vocoder(V.ConfVocoder())
voc.load_state_dict(synth.vocoder_state_dict())
voc = voc.cuda().eval()
z = synth.randint("timeline", (1, 4), 512).cuda()
spk = torch.zeros(1, dtype=torch.long, device="cuda")
voc.generate(z, spk, seed=13)
torch.cuda.synchronize()
ms, n = voc.last_timing()
buf = (C.c_ulonglong * (128 * 12))()
assert _lib.load().vqcpc_debug_ps_stamps(buf) == 0
s = np.array(buf, dtype=np.int64).reshape(128, 12) * 0.01     # us
a, b = s[4:120], s[5:121]
rows = [
    ("service wave: step entry -> x_{t-1} known (candidate sweep + argmax)", a[:, 1] - a[:, 0]),
    ("service wave: x known -> own h_t published (cell update)", a[:, 2] - a[:, 1]),
    ("service wave: h published -> h_t of all 64 workgroups gathered", a[:, 3] - a[:, 2]),
    ("barrier A", a[:, 4] - a[:, 3]),
    ("wave 0: barrier A -> fc1 rows published", a[:, 7] - a[:, 4]),
    ("service wave: barrier A -> a_t gathered", a[:, 5] - a[:, 4]),
    ("wave 1: barrier A -> W_hh h_t chains done", a[:, 9] - a[:, 4]),
    ("barrier B (service wave's view)", a[:, 6] - a[:, 5]),
    ("wave 0: barrier B -> candidate published (fc2 + draw)", a[:, 8] - a[:, 6]),
    ("candidate published -> next step's x known", b[:, 1] - a[:, 8]),
    ("whole step", b[:, 0] - a[:, 0]),
]
print(f"persistent decoder, 1 utterance: {ms * 1e3 / n:.2f} us per sample over the call; one workgroup, mean / min / max over 116 steps, us")
for name, d in rows:
    print(f"{name:72s} {d.mean():6.2f} {d.min():6.2f} {d.max():6.2f}")
