#!/usr/bin/env python3
"""Where one sample step of the matrix-core per-XCD decoder (csrc/ar_xcm.hip) goes: wall-clock stamps (10 ns ticks) of
worker 5 of XCD 0 over steps 256..383.  Needs the debug build: sh tools/build_stamps.sh

    python3 tools/xcm_timeline.py [utterances ...]
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
voc.set_option("xcm", 1)
for B in [int(a) for a in sys.argv[1:]] or [128]:
    z = synth.randint("timeline", (B, 4), 512).cuda()
    spk = torch.zeros(B, dtype=torch.long, device="cuda")
    voc.generate(z, spk, seed=13)
    voc.check()
    ms, n = voc.last_timing()
    buf = (C.c_ulonglong * (128 * 24))()
    lib = _lib.load()
    lib.vqcpc_debug_xm_stamps.argtypes = [C.c_void_p]
    assert lib.vqcpc_debug_xm_stamps(buf) == 0
    s = np.array(buf, dtype=np.int64).reshape(128, 24) * 0.01     # us
    a, b = s[4:120], s[5:121]
    rows = [
        ("wave 0: past barrier B -> h_t of the owned units published (cell update)", a[:, 1] - a[:, 0]),
        ("wave 0: h_t of all 32 workers swept into LDS", a[:, 2] - a[:, 1]),
        ("wave 0: barrier A", a[:, 3] - a[:, 2]),
        ("wave 0: 56 MFMAs of its quarter of tile 5 (the oldest wave of its SIMD)", a[:, 4] - a[:, 3]),
        ("wave 1: 56 MFMAs of its quarter of tile 5 (the oldest wave of its SIMD)", a[:, 16] - a[:, 3]),
        ("wave 4: 56 MFMAs of its first quarter (second wave of SIMD 0)", a[:, 15] - a[:, 3]),
        ("wave 0: -> the other three quarters of tile 5 seen", a[:, 18] - a[:, 4]),
        ("wave 0: -> a_t published (fc1 + ReLU)", a[:, 5] - a[:, 18]),
        ("wave 0: 56 MFMAs of its quarter of tile 4", a[:, 22] - a[:, 5]),
        ("wave 0: its K quarter of a_t swept", a[:, 6] - a[:, 22]),
        ("wave 0: fc2 MFMAs of its K quarter", a[:, 20] - a[:, 6]),
        ("wave 0: -> the other three quarters seen", a[:, 21] - a[:, 20]),
        ("wave 0: -> candidates published", a[:, 8] - a[:, 21]),
        ("wave 0: candidates of its four slots swept, x_t, sample out", a[:, 9] - a[:, 8]),
        ("wave 0: embedding rows requested, barrier B", b[:, 0] - a[:, 9]),
        ("whole step", b[:, 0] - a[:, 0]),
    ]
    print(f"matrix-core per-XCD decoders, {B} utterance(s): {ms * 1e3 / n:.2f} us per sample step over the call; worker 5 of XCD 0, "
          f"mean / min / max over 116 steps, us")
    for name, d in rows:
        print(f"  {name:84s} {d.mean():6.2f} {d.min():6.2f} {d.max():6.2f}")
