#!/usr/bin/env python3
"""Self-oracle fixtures of the vocoder half (lives under tests/: the only place besides
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline that may use ``oracle/``).

The reference's vocoder arithmetic (third-party ``rnnms``) is absent offline, so these vectors come
from this project's own CPU oracle and pin the SPEC against drift -- parity with ``rnnms`` is unpinned.
(``vocoder_glue.npz`` is NOT made here: ``tools/gen_golden.py`` records it from the reference's own
``network_vocoder.py`` with a capture stub for ``rnnms``.)

Usage:  python tests/golden/make_vocoder_fixtures.py
"""
import os
import sys

import numpy as np
import torch

GOLD = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLD))
sys.path.insert(0, ROOT)
from vectorquantizedcpc_amd import synth  # noqa: E402


def vocoder_selforacle():
    import oracle
    sd = synth.vocoder_state_dict()
    out = {}
    for u, (tc, steps) in enumerate(((3, 960), (2, 640))):
        z = synth.randint(f"voc/z{u}", (tc,), 512).numpy()
        spk = int(synth.randint(f"voc/spk{u}", (1,), 102)[0])
        r = oracle.vocoder_generate(sd, z, spk, seed=synth.SEED, utterance=u, n_steps=steps, want_logits=True)
        out[f"z{u}"], out[f"spk{u}"] = z, np.array(spk)
        out[f"samples{u}"] = r["samples"].astype(np.int16)
        out[f"wav{u}"] = r["wav"]
        out[f"logits{u}"] = r["logits"][::64].copy()       # every 64th step
        out[f"cond{u}"] = oracle.vocoder_condition(sd, z, spk)
    np.savez_compressed(os.path.join(GOLD, "vocoder_selforacle.npz"), **out)


if __name__ == "__main__":
    vocoder_selforacle()
    print("wrote vocoder_selforacle.npz")
