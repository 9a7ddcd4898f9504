#!/usr/bin/env python3
"""The decode table of DESIGN 4: `bench.py --utterances-per-gpu N --no-extras --no-cpu-baseline --manifest 0` for a list of
N (each in a fresh process), one CSV row per N on stdout.

    python3 tools/bench_by_batch.py [N ...]        (default 1 16 32 64 128 256 512)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(a) for a in sys.argv[1:]] or [1, 16, 32, 64, 128, 256, 512]
print("utterances_per_gpu,samples_per_s,step_us,gru_launch_us,gru_frac_fp32_peak,utterances_per_launch,launches_per_sample", flush=True)
for n in sizes:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--utterances-per-gpu", str(n), "--no-extras",
                          "--no-cpu-baseline", "--manifest", "0"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    d = json.loads(out.stdout.decode().strip().splitlines()[-1])
    r = d["roofline"]
    launch = r.get("avg_launch_us")
    print(f"{n},{d['value']:.0f},{r['decode_step']['us']:.2f},{'' if launch is None else f'{launch:.2f}'},{r['frac']:.3f},"
          f"{r.get('utterances_per_launch', n)},{r.get('launches_per_sample', '')}", flush=True)
