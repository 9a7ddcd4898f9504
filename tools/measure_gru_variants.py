#!/usr/bin/env python3
"""GRU-step kernel variants at a given batch: the packed 12-row two-tile kernel (224 workgroups)
against the full-16-row-tile LDS-staged kernel (56 workgroups per tile pair), HIP events around 1000 back-to-back
launches each (includes the ~1.5 us dependent-launch boundary), plus the whole decode step per sample.

    python3 tools/measure_gru_variants.py [utterances ...]      (default 16 32 48 64 96)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectorquantizedcpc_amd as V
from vectorquantizedcpc_amd import synth

voc = V.Vocoder(V.ConfVocoder())
voc.load_state_dict(synth.vocoder_state_dict())
voc = voc.cuda().eval()
sizes = [int(a) for a in sys.argv[1:]] or [16, 32, 48, 64, 96]
print("utterances,variant,gru_us,fc1_us,fc2_us,slots_per_launch,kernel_kind,step_us")
for B in sizes:
    z = synth.randint("var/z", (B, 10), 512).cuda()
    spk = (torch.arange(B) % 102).cuda()
    for name, opts in (("default", {}), ("three_launches", {"fuse_fc2": 0}), ("one_group", {"two_groups": 0}),
                       ("full_tile_lds_kernel", {"big_min_tiles": 1 if B > 16 else 0, "two_groups": 0})):
        if name == "full_tile_lds_kernel" and B <= 16:
            continue
        for k, v in opts.items():
            voc.set_option(k, v)
        voc.generate(z, spk, seed=1, utt_base=0)
        voc.generate(z, spk, seed=1, utt_base=0)
        ms, n = voc.last_timing()
        kt = voc.kernel_times(1000)
        print(f"{B},{name},{kt[0]:.3f},{kt[1]:.3f},{kt[2]:.3f},{int(kt[3])},{int(kt[4])},{ms * 1e3 / n:.3f}", flush=True)
        voc.set_option("big_min_tiles", 5)
        voc.set_option("two_groups", 1)
        voc.set_option("fuse_fc2", 1)
