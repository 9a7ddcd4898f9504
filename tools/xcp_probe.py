#!/usr/bin/env python3
"""Pipelined per-XCD decoders (csrc/ar_xcp.hip) against the lockstep ones (ar_xcd.hip) and the launch-per-step kernels:
same samples?  how fast?

    python tools/xcp_probe.py [--quick] [--slots N]

One line per case: utterances, samples each, bit-equal to both other paths (mu-law classes and waveform), us per sample step
of the three paths (HIP events around the decode loop, vqcpc_vocoder_last_timing).  Writes gpurun_out/xcp_probe.csv.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth

MODES = {"xcp": (1, 1), "xcd": (1, 0), "launch": (0, 0)}
WANT_PATH = {"xcp": 4, "xcd": 2, "launch": 0}


def main():
    quick = "--quick" in sys.argv
    sd = synth.vocoder_state_dict()
    voc = V.Vocoder(V.ConfVocoder())
    voc.load_state_dict(sd)
    voc = voc.to("cuda").eval()
    voc.set_option("xcm", 0)
    voc.set_option("xcd_timeout_ms", 100)
    if "--lag" in sys.argv:
        voc.set_option("xcp_cell_lag", int(sys.argv[sys.argv.index("--lag") + 1]))
    rows = ["case,utterances,steps,equal_xcd,equal_launch,us_per_step_xcp,us_per_step_xcd,us_per_step_launch,Msamples_s_xcp"]
    cases = [(9, 2, 640), (16, 2, 640), (17, 3, 700), (24, 2, 640), (32, 2, 640), (40, 2, 640)]
    if not quick:
        cases += [(16, 100, 32000), (24, 100, 32000), (32, 100, 32000), (64, 100, 32000)]
    if "--only" in sys.argv:
        keep = [int(v) for v in sys.argv[sys.argv.index("--only") + 1].split(",")]
        cases = [c for c in cases if c[0] in keep and c[2] > 8000]
    for B, Tc, steps in cases:
        z = synth.randint(f"xp/z{B}", (B, Tc), 512).cuda()
        spk = synth.randint(f"xp/s{B}", (B,), 102).cuda()
        n_codes = None
        if B in (17, 40):                                   # ragged
            n_codes = [max(1, Tc - (b % 3)) for b in range(B)]
        res = {}
        for name, (xcd, xcp) in MODES.items():
            if name == "launch" and steps > 8000 and B != 32:
                continue
            voc.set_option("xcd", xcd)
            voc.set_option("xcp", xcp)
            t0 = time.time()
            wav, mu = voc.generate(z, spk, n_codes=n_codes, seed=13, utt_base=7, return_mulaw=True, max_steps=steps)
            try:
                voc.check()
                err = ""
            except RuntimeError as e:
                err = str(e)
            ms, n = voc.last_timing()
            path = voc.last_path()
            if not err and B > 8 and path != WANT_PATH[name]:
                err = f"path {path}"
            res[name] = (wav.cpu(), mu.cpu(), ms, n, err, time.time() - t0)
        def eq(a, b):
            return bool(torch.equal(res[a][1], res[b][1]) and torch.equal(res[a][0], res[b][0])) if a in res and b in res else None
        us = {k: res[k][2] * 1e3 / max(1, res[k][3]) if k in res else float("nan") for k in MODES}
        tot = float(sum(min(steps, 320 * (Tc if n_codes is None else n)) for n in (n_codes or [Tc] * B)))
        line = (f"B{B}xT{steps},{B},{steps},{eq('xcp', 'xcd')},{eq('xcp', 'launch')},{us['xcp']:.3f},{us['xcd']:.3f},{us['launch']:.3f},"
                f"{tot / res['xcp'][2] / 1e3:.3f}")
        print(line, "| wall s", " ".join(f"{res[k][5]:.2f}" for k in res), "|", " / ".join(res[k][4][:100] for k in res), flush=True)
        if eq("xcp", "xcd") is False:
            d = (res["xcp"][1] != res["xcd"][1]).nonzero()
            print("   first differences (row, t):", d[:8].tolist(), "of", len(d), flush=True)
        rows.append(line)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/xcp_probe.csv", "w") as f:
        f.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
