#!/usr/bin/env python3
"""A/B timing of alternative builds of libvqcpc_hip.so on one GPU box: the encoder at BASELINE configs[1] (64 x 80 x 128) and
configs[0] (1 x 80 x 200), ms per Encoder.encode_indices call (wall over 200 back-to-back calls), each library in a fresh process.

    python tools/ab_encoder.py build/exp/F/libvqcpc_hip.so build/exp/G/libvqcpc_hip.so
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from vectorquantizedcpc_amd import _lib
_lib.LIB_PATH = sys.argv[1]
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth
enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256)); enc.load_state_dict(synth.encoder_state_dict()); enc = enc.cuda().eval()
out = []
for name, (B, T) in {"c2": (64, 128), "c1": (1, 200), "8k rows": (128, 128)}.items():
    m = synth.mel("bench/" + name, B, T).cuda()
    for _ in range(20): enc.encode_indices(m)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): enc.encode_indices(m)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
    out.append(f"{name}: {best * 1e3:.4f} ms")
print(sys.argv[1], " ".join(out), flush=True)
''' % ROOT
for rnd in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib)], check=False)
