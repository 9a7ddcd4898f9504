#!/usr/bin/env python3
"""Per-launch kernel time of the encoder's six column-split launches, from a rocprofv3 kernel trace of
tools/profile_encoder.py c1:   python3 tools/split_layer_times.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if "enc_split_" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
acc = defaultdict(list)
gaps = []
for i, (s, e, g) in enumerate(rows):
    acc[i % 6].append(e - s)
    if i % 6:
        gaps.append(s - rows[i - 1][1])
print("launch,kernel_us_mean,kernel_us_min,n")
for k in range(6):
    v = acc[k][5:]
    print(f"{k},{sum(v) / len(v) / 1e3:.2f},{min(v) / 1e3:.2f},{len(v)}")
print(f"gap between consecutive launches of a call: mean {sum(gaps) / len(gaps) / 1e3:.2f} us")
