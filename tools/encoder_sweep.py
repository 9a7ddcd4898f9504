#!/usr/bin/env python3
"""Encoder front-end schedules over call sizes: one-launch fused (1), column-split launches (2), layered kernels (0).
Wall time per call, averaged over back-to-back calls with the inputs resident (microseconds).

    python3 tools/encoder_sweep.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth

enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
enc.load_state_dict(synth.encoder_state_dict())
enc = enc.cuda().eval()
print("B,T,rows,fused_us,split_us,layered_us")
for B, T in ((1, 32), (1, 200), (1, 1000), (4, 200), (8, 200), (16, 128), (32, 128), (64, 128), (128, 128), (256, 128)):
    mel = synth.mel("sweep", B, T).cuda()
    row = []
    for mode in (1, 2, 0):
        enc.set_option("fused", mode)
        for _ in range(5):
            enc.encode_indices(mel)
        torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            enc.encode_indices(mel)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / reps * 1e6)
    enc.set_option("fused", -1)
    print(f"{B},{T},{B * (T // 2)},{row[0]:.1f},{row[1]:.1f},{row[2]:.1f}", flush=True)
