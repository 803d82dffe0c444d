#!/bin/bash
# Timing-only builds of libqmri (parts of the conv kernels removed; wrong results by design; host code compiled with -DQMRI_TIMING_ONLY so that the
# set-up probe and the range guard do not move the run to the bf16 scheme).  Usage: bash tools/build_timing_only.sh C6_NO_SPLIT C6_NO_MFMA C6_LOADER_IDLE C6P_LOADER_IDLE C6_NO_STORES  (macros of conv6_kernels.hip)
# -> tools/ab/libqmri_<variant>.so (and libqmri_TO.so: production kernels + timing-only host code, the control).  Needs an up-to-date csrc/_build.
set -e
cd "$(dirname "$0")/../qmri_pnp_recon_poc_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -w"
mkdir -p ../../tools/ab
hipcc $F -DQMRI_TIMING_ONLY -c api_net.cpp -o /tmp/api_net_to.o
REST=$(ls _build/*.o | grep -v "conv6_kernels\|api_net")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libqmri_TO.so _build/conv6_kernels.o /tmp/api_net_to.o $REST
for v in "$@"; do
  hipcc $F -D$v -c conv6_kernels.hip -o /tmp/conv6_$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libqmri_$v.so /tmp/conv6_$v.o /tmp/api_net_to.o $REST
done
ls -la ../../tools/ab
