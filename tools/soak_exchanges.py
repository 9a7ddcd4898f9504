#!/usr/bin/env python3
"""Soak of the three in-kernel exchange paths for ~45 s: the resident context scan (encode of one utterance), the persistent
single-utterance decoder (4000 samples) and the fused fc2 || GRU batch decode (32 x 1920 samples), every result compared bit for
bit with the first round's.  A timed-out exchange would surface as an error of the next call.

    python3 tools/soak_exchanges.py
"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth
enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256)); enc.load_state_dict(synth.encoder_state_dict()); enc = enc.cuda().eval()
voc = V.Vocoder(V.ConfVocoder()); voc.load_state_dict(synth.vocoder_state_dict()); voc = voc.cuda().eval()
mel = synth.mel("soak/mel", 1, 200).cuda()
z0, c0, i0 = enc.encode(mel)
spk = torch.zeros(1, dtype=torch.long, device="cuda")
w0 = voc.generate(i0, spk, seed=3, utt_base=0, max_steps=4000)
z32 = synth.randint("soak/z", (32, 6), 512).cuda(); spk32 = (torch.arange(32) % 102).cuda()
w32 = voc.generate(z32, spk32, seed=3, utt_base=0)
t0 = time.time()
n = 0
while time.time() - t0 < 45:
    z, c, i = enc.encode(mel)
    assert torch.equal(c, c0) and torch.equal(i, i0)
    w = voc.generate(i0, spk, seed=3, utt_base=0, max_steps=4000)
    assert torch.equal(w, w0)
    wb = voc.generate(z32, spk32, seed=3, utt_base=0)
    assert torch.equal(wb, w32)
    n += 1
    if n % 50 == 0:
        print(n, "rounds ok", flush=True)
print("soak ok:", n, "rounds of resident context scan + persistent single-utterance decode (4000 samples) + fused batch decode (32 x 1920)")
