#!/bin/bash
# Debug build of libvqcpc_hip.so with the in-kernel timeline stamps compiled in (tools/xcd_timeline.py, xcm_timeline.py,
# xcp_timeline.py): build/stamps/libvqcpc_hip.so -- never the shipped library.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/stamps
for f in encoder vocoder ar_xcd ar_xcp ar_xcm melfront loudness resample; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVQCPC_XD_STAMPS \
        -c vectorquantizedcpc_amd/csrc/$f.hip -o build/stamps/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libvqcpc_hip.so build/stamps/*.o
