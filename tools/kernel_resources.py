#!/usr/bin/env python3
"""One line per kernel of a .hip file: VGPRs, SGPRs, spills, scratch, occupancy (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py vectorquantizedcpc_amd/csrc/ar_xcd.hip [name filter]

Rows are keyed on the MANGLED name (kernels in an anonymous namespace demangle to names that all begin with "(anonymous
namespace)::", and template instantiations share everything in front of their arguments); the demangled name is what is printed.
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
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|"
                  r"LDS Size \[bytes/block\]): (.*?) \[-R", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
names = {}
if rows:
    dem = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
    names = dict(zip(rows, dem))
for mangled, r in rows.items():
    name = names.get(mangled, mangled)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)          # drop the parameter list, keep template arguments
    if flt in name:
        print(f"{name:64s} vgpr {r.get('VGPRs', '?'):>4} agpr {r.get('AGPRs', '?'):>3} sgpr {r.get('TotalSGPRs', '?'):>3} "
              f"spill v{r.get('VGPRs Spill', '?')}/s{r.get('SGPRs Spill', '?')} scratch {r.get('ScratchSize', '?')} occ {r.get('Occupancy', '?')}")
