#!/usr/bin/env python3
"""A/B timing of alternative builds of libvqcpc_hip.so on one GPU box: us per sample step of the per-XCD decoders.

    python tools/ab_libs.py build/exp/A/libvqcpc_hip.so build/exp/B/libvqcpc_hip.so ...   (each in a fresh child process)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from vectorquantizedcpc_amd import _lib
_lib.LIB_PATH = sys.argv[1]
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth
voc = V.Vocoder(V.ConfVocoder()); voc.load_state_dict(synth.vocoder_state_dict()); voc = voc.cuda().eval()
out = []
for B in (1, 16, 32, 128):
    z = synth.randint(f"xp/z{B}", (B, 100), 512).cuda(); spk = synth.randint(f"xp/s{B}", (B,), 102).cuda()
    best = 1e9
    for rep in range(3):
        voc.generate(z, spk, seed=13, utt_base=7); voc.check()
        ms, n = voc.last_timing()
        best = min(best, ms * 1e3 / n)
    out.append(f"{B}: {best:.3f}")
print(sys.argv[1], " ".join(out), flush=True)
''' % ROOT
for rnd in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib)], check=False)
