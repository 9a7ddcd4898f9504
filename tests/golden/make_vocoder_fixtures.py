#!/usr/bin/env python3
"""Self-oracle fixtures of the vocoder half (lives under tests/: the only place besides
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline that may use ``oracle/``).

The reference's vocoder arithmetic (third-party ``rnnms``) is absent offline, so these vectors come
from this project's own CPU oracle and pin the SPEC against drift -- parity with ``rnnms`` is unpinned.

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


def glue_fixture():
    """network_vocoder.py:69-77 glue layout, restated with torch ops on CPU and frozen as data.

    (The reference module itself needs ``rnnms`` to import; SURVEY 8c verified this layout
    against the reference with a capture stub.  Parity of the recurrence stays unpinned.)
    """
    sd = synth.vocoder_state_dict()
    z = synth.randint("glue/z", (2, 5), 512)
    spk = synth.randint("glue/spk", (2,), 102)
    ze = torch.nn.functional.embedding(z, sd["code_embedding.weight"])
    zu = torch.nn.functional.interpolate(ze.transpose(1, 2), scale_factor=2).transpose(1, 2)
    se = torch.nn.functional.embedding(spk, sd["speaker_embedding.weight"])
    series = torch.cat((zu, se.unsqueeze(1).expand(-1, zu.size(1), -1)), dim=-1)
    np.savez_compressed(os.path.join(GOLD, "vocoder_glue.npz"), z=z.numpy(), speaker=spk.numpy(),
                        series=series.numpy())


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
    glue_fixture()
    vocoder_selforacle()
    print("wrote vocoder_glue.npz, vocoder_selforacle.npz")
