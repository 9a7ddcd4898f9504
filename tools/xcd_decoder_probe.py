#!/usr/bin/env python3
"""Per-XCD resident decoders (csrc/ar_xcd.hip) against the launch-per-step kernels: same samples?  how fast?

    python tools/xcd_decoder_probe.py [--quick]

Prints one line per case: utterances, samples each, bit-equal (mu-law classes and waveform), us per sample step of both
paths (HIP events around the decode loop, vqcpc_vocoder_last_timing).  Writes gpurun_out/xcd_probe.csv.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth


def main():
    quick = "--quick" in sys.argv
    sd = synth.vocoder_state_dict()
    voc = V.Vocoder(V.ConfVocoder())
    voc.load_state_dict(sd)
    voc = voc.to("cuda").eval()
    rows = ["case,utterances,steps,equal,us_per_step_xcd,us_per_step_launch,Msamples_s_xcd,Msamples_s_launch"]
    cases = [(1, 2, 640), (2, 2, 640), (3, 3, 960), (4, 2, 640), (9, 3, 700), (32, 2, 640)]
    if not quick:
        cases += [(1, 100, 32000), (8, 100, 32000), (16, 100, 32000), (32, 100, 32000), (64, 100, 32000)]
    for B, Tc, steps in cases:
        z = synth.randint(f"xp/z{B}", (B, Tc), 512).cuda()
        spk = synth.randint(f"xp/s{B}", (B,), 102).cuda()
        n_codes = None
        if B in (3, 9):                                   # ragged
            n_codes = [max(1, Tc - (b % 3)) for b in range(B)]
        res = {}
        for mode in (1, 0):
            voc.set_option("xcd", mode)
            t0 = time.time()
            wav, mu = voc.generate(z, spk, n_codes=n_codes, seed=13, utt_base=7, return_mulaw=True, max_steps=steps)
            try:
                voc.check()
                err = ""
            except RuntimeError as e:
                err = str(e)
            ms, n = voc.last_timing()
            res[mode] = (wav.cpu(), mu.cpu(), ms, n, err, time.time() - t0)
        eq = bool(torch.equal(res[1][1], res[0][1]) and torch.equal(res[1][0], res[0][0]))
        nz = int((res[0][1] != 0).sum())
        us1 = res[1][2] * 1e3 / max(1, res[1][3])
        us0 = res[0][2] * 1e3 / max(1, res[0][3])
        tot = float(sum(min(steps, 320 * (Tc if n_codes is None else n)) for n in (n_codes or [Tc] * B)))
        line = f"B{B}xT{steps},{B},{steps},{eq},{us1:.3f},{us0:.3f},{tot / res[1][2] / 1e3:.3f},{tot / res[0][2] / 1e3:.3f}"
        print(line, "| nonzero classes", nz, "| wall s", f"{res[1][5]:.2f} {res[0][5]:.2f}", "|", res[1][4][:120], flush=True)
        if not eq:
            d = (res[1][1] != res[0][1]).nonzero()
            print("   first differences (row, t):", d[:5].tolist(), flush=True)
        rows.append(line)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/xcd_probe.csv", "w") as f:
        f.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
