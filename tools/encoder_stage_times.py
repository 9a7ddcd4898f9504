#!/usr/bin/env python3
"""Where the fused encoder launch goes: the kernel can stop after any front-end stage (`vqcpc_encoder_stage`, the dump of
the 16 x 512 tile costs ~1 us), so the wall time of a call that stops after stage k is the cumulative time up to k.

    python3 tools/encoder_stage_times.py [B T]        (default 64 128 = BASELINE configs[1])
Stages: 0 conv, 1 LN0+ReLU, 2 fc0, 3 LN1, 4 fc1, 5 LN2, 6 fc2, 7 LN3, 8 fc3, 9 LN4, 10 encoder.14 (z_pre); then the whole call.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("VQCPC_LIB"):                       # A/B against another build of the library
    from vectorquantizedcpc_amd import _lib
    _lib.LIB_PATH = os.environ["VQCPC_LIB"]
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
enc.load_state_dict(synth.encoder_state_dict())
enc = enc.cuda().eval()
enc.set_option("fused", 1)
mel = synth.mel("stage/mel", B, T).cuda()


def wall(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


names = ["conv", "LN0", "fc0", "LN1", "fc1", "LN2", "fc2", "LN3", "fc3", "LN4", "encoder.14"]
print(f"{B} x {T} frames -> {B * (T // 2)} rows; us per call stopping after each stage (cumulative), and the increment")
prev = 0.0
for k, nm in enumerate(names):
    us = wall(lambda: enc.stage(mel, k))
    print(f"{k:2d} {nm:11s} {us:8.1f} {us - prev:+8.1f}")
    prev = us
us = wall(lambda: enc.encode_indices(mel))
print(f"   whole call  {us:8.1f} {us - prev:+8.1f}   (+ VQ search, no tile dump)")
