#!/usr/bin/env python3
"""Soak of the in-kernel exchange paths for ~60 s: the resident context scan (encode of one utterance), the per-XCD resident
decoders at 1, 12 and 32 utterances (one, two and four slots per XCD; four slots = the chains on the matrix pipe, fc1's SIMD mates
held behind an LDS flag), 45 ragged utterances through the 32 slots (slot hand-over), their 16-slot matrix-core form at 100 utterances and at 150
utterances through its 128 slots, and the fused fc2 || GRU launches (`xcd` = 0), every result compared bit for bit with the
first round's and every call followed by check().

    python3 tools/soak_exchanges.py [seconds]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectorquantizedcpc_amd as V  # noqa: E402
from vectorquantizedcpc_amd import synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
enc = V.Encoder(V.ConfEncoder(80, 512, 512, 64, 256))
enc.load_state_dict(synth.encoder_state_dict())
enc = enc.cuda().eval()
voc = V.Vocoder(V.ConfVocoder())
voc.load_state_dict(synth.vocoder_state_dict())
voc = voc.cuda().eval()
mel = synth.mel("soak/mel", 1, 200).cuda()
z0, c0, i0 = enc.encode(mel)
enc.check()
cases = []
for name, B, Tc, opts in (("xcd 1", 1, 12, {}), ("xcd 12", 12, 6, {}), ("xcd 32", 32, 6, {}), ("xcd 45 ragged", 45, 5, {}), ("xcm 100", 100, 4, {}),
                          ("xcm 150 ragged", 150, 4, {}), ("launches 32", 32, 6, {"xcd": 0})):
    z = synth.randint("soak/z" + name, (B, Tc), 512).cuda()
    spk = (torch.arange(B) % 102).cuda()
    n_codes = [1 + (3 * b) % Tc for b in range(B)] if "ragged" in name else None
    cases.append((name, z, spk, n_codes, opts, None))
want_path = {"xcd 1": 2, "xcd 12": 2, "xcd 32": 2, "xcd 45 ragged": 2, "xcm 100": 3, "xcm 150 ragged": 3, "launches 32": 0}
ref = {}
t0 = time.time()
n = 0
while time.time() - t0 < seconds:
    z, c, i = enc.encode(mel)
    enc.check()
    assert torch.equal(c, c0) and torch.equal(i, i0)
    for name, zz, spk, n_codes, opts, _ in cases:
        for k, v in opts.items():
            voc.set_option(k, v)
        w, mu = voc.generate(zz, spk, n_codes=n_codes, seed=3, utt_base=0, return_mulaw=True)
        voc.check()
        assert voc.last_path() == want_path[name], (name, voc.last_path())
        for k in opts:
            voc.set_option(k, -1)
        if name not in ref:
            ref[name] = (w.clone(), mu.clone())
        assert torch.equal(w, ref[name][0]) and torch.equal(mu, ref[name][1]), name
    n += 1
    if n % 20 == 0:
        print(n, "rounds ok", flush=True)
print("soak ok:", n, "rounds of resident context scan + " + ", ".join(c[0] for c in cases) + ": every call checked, every result bit-stable")
