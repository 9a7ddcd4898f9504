#!/usr/bin/env python3
"""Encoder-only loop for rocprofv3: BASELINE configs[1] (64 x 80 x 128 mel -> 4096 code frames), or with `c1` as
first argument configs[0]'s single 2 s utterance (1 x 80 x 200); a trailing `context` also runs the LSTM.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_enc -- python3 tools/profile_encoder.py [c1] [context]
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
args = sys.argv[1:]
mel = (synth.mel("bench/c1", 1, 200) if "c1" in args else synth.mel("bench/c2", 64, 128)).cuda()
want_c = "context" in args
for _ in range(3):
    enc.encode(mel) if want_c else enc.encode_indices(mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 50
for _ in range(reps):
    enc.encode(mel) if want_c else enc.encode_indices(mel)
torch.cuda.synchronize()
print(f"{tuple(mel.shape)} {'encode' if want_c else 'encode_indices'}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call")
