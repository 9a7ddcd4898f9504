#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel.

    python tools/pmc_summary.py <dir with *_counter_collection.csv> [kernel-name substring]

Prints ``kernel,counter,dispatches,avg_per_dispatch`` (CSV), the layout of profiles/r01_pmc_*.csv."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if needle in k:
                a = acc[(k, row["Counter_Name"])]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "avg_per_dispatch"])
for (k, c), (n, s) in sorted(acc.items()):
    w.writerow([k, c, n, round(s / n, 1)])
