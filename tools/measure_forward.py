#!/usr/bin/env python3
"""teacher-forced Vocoder.forward at the training shape of the reference
(batch 32, mel clip of 32 frames = 16 codes = 5120 samples, vocoder.py:51-66 / config.py:101,117)
and eval-mode Encoder.forward (model.py:72-86) at 64 x 128 frames."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth

enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256)); enc.load_state_dict(synth.encoder_state_dict()); enc = enc.cuda().eval()
voc = V.Vocoder(V.ConfVocoder()); voc.load_state_dict(synth.vocoder_state_dict()); voc = voc.cuda().eval()
mel = synth.mel("fwd/mel", 32, 32).cuda()
idx = enc.encode_indices(mel)
x = synth.randint("fwd/x", (32, 5119), 256).cuda()                  # audio[:, :-1] (vocoder.py:62)
spk = (torch.arange(32) % 102).cuda()
voc(x, idx, spk)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    out = voc(x, idx, spk)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"Vocoder.forward 32 x 5119 -> {tuple(out.shape)}: {dt * 1e3:.1f} ms ({5119 * 32 / dt / 1e6:.2f} M teacher-forced samples/s)")
m2 = synth.mel("fwd/c2", 64, 128).cuda()
enc(m2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    z, c, loss, ppl = enc(m2)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"Encoder.forward 64 x 128 (z, c, loss {float(loss):.5f}, perplexity {float(ppl):.3f}): {dt * 1e3:.3f} ms")
