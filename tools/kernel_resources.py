#!/usr/bin/env python3
"""One line per kernel of a .hip file: VGPRs, SGPRs, spills, occupancy, LDS (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py vectorquantizedcpc_amd/csrc/vocoder.hip [name filter]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (.*?) \[-R", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for name, r in rows.items():
    if flt in name:
        print(f"{name:60s} vgpr {r.get('VGPRs'):>4} agpr {r.get('AGPRs'):>3} sgpr {r.get('TotalSGPRs'):>3} "
              f"spill v{r.get('VGPRs Spill')}/s{r.get('SGPRs Spill')} occ {r.get('Occupancy')} lds {r.get('LDS Size')}")
