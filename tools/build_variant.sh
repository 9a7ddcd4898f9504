#!/bin/sh
# A/B builds of the per-XCD decoder: build/exp/NAME/libvqcpc_hip.so = the current objects with ar_xcd.hip recompiled with extra defines.
# usage: tools/build_variant.sh NAME "-DXD_GSPLIT=8 ..."   (then: python tools/ab_libs.py build/exp/A/libvqcpc_hip.so build/exp/B/libvqcpc_hip.so)
set -e
cd "$(dirname "$0")/.."
C=vectorquantizedcpc_amd/csrc
mkdir -p build/exp/$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $2 -c $C/ar_xcd.hip -o build/exp/$1/ar_xcd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/$1/libvqcpc_hip.so $C/encoder.o $C/vocoder.o build/exp/$1/ar_xcd.o $C/ar_xcm.o $C/melfront.o $C/loudness.o $C/resample.o
