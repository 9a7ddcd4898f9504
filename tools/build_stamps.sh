#!/bin/sh
# Debug build of libvqcpc_hip.so with the in-kernel timeline stamps compiled in (build/stamps/, never the shipped library).
# usage: tools/build_stamps.sh [-DVQCPC_XD_STAMPS | -DVQCPC_AR_STAMPS]
set -e
cd "$(dirname "$0")/.."
DEF=${1:--DVQCPC_XD_STAMPS}
mkdir -p build/stamps
for f in encoder vocoder ar_xcd ar_xcm melfront loudness resample; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $DEF -c vectorquantizedcpc_amd/csrc/$f.hip -o build/stamps/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libvqcpc_hip.so build/stamps/*.o
