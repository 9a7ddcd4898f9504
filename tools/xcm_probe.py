#!/usr/bin/env python3
"""Probe of the matrix-core per-XCD decoders (csrc/ar_xcm.hip): same bits as the launch-per-step kernels, and the step time.

    python3 tools/xcm_probe.py [quick]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    sd = synth.vocoder_state_dict()
    voc = V.Vocoder(V.ConfVocoder())
    voc.load_state_dict(sd)
    voc = voc.to("cuda").eval()
    cases = [(1, 2, None), (5, 2, None), (16, 2, None), (17, 3, "ragged"), (40, 2, "ragged"), (128, 2, None), (150, 2, "ragged")]
    if not quick:
        cases += [(128, 100, None), (64, 100, None), (256, 100, None)]
    print("case,utterances,samples,same_bits_as_launch_path,xcm_us_per_step,launch_us_per_step,xcm_Msamples_per_s,launch_Msamples_per_s")
    for B, Tc, rag in cases:
        z = synth.randint(f"xcm/z{B}", (B, Tc), 512).cuda()
        spk = synth.randint(f"xcm/s{B}", (B,), 102).cuda()
        n_codes = [max(1, Tc - (b % Tc)) for b in range(B)] if rag else None
        out = {}
        for name, opts in (("xcm", {"xcd": -1, "xcm": 1}), ("launch", {"xcd": 0, "xcm": 0})):
            for k, v in opts.items():
                voc.set_option(k, v)
            torch.cuda.synchronize()
            t0 = time.time()
            wav, mu = voc.generate(z, spk, n_codes=n_codes, seed=13, utt_base=3, return_mulaw=True)
            voc.check()
            torch.cuda.synchronize()
            wall = time.time() - t0
            ms, n = voc.last_timing()
            out[name] = (wav.cpu(), mu.cpu(), ms * 1e3 / max(n, 1), voc.last_path(), wall)
        same = torch.equal(out["xcm"][1], out["launch"][1]) and torch.equal(out["xcm"][0], out["launch"][0])
        total = 320 * sum(n_codes or [Tc] * B)
        nz = int((out["xcm"][1] != 0).sum())
        print(f"B{B}xT{320 * Tc},{B},{320 * Tc},{same},{out['xcm'][2]:.3f},{out['launch'][2]:.3f},"
              f"{total / max(out['xcm'][4], 1e-9) / 1e6:.2f},{total / max(out['launch'][4], 1e-9) / 1e6:.2f} | paths {out['xcm'][3]} {out['launch'][3]} | nonzero {nz} of {total}", flush=True)
        if not same:
            d = (out["xcm"][1] != out["launch"][1])
            print("   first mismatch rows:", d.any(1).nonzero().flatten()[:8].tolist(), "first col:", int(d.any(0).nonzero().flatten()[0]) if d.any() else -1)
    voc.set_option("xcd", -1)
    voc.set_option("xcm", -1)


if __name__ == "__main__":
    main()
