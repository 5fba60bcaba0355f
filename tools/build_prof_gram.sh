#!/bin/bash
# libbgmm_hip_prof.so with the phase clocks of gram_resolve_kernel compiled in (tools/probe.py chain N D K --init rand --prof)
set -e
cd "$(dirname "$0")/../pybgmm_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
hipcc $FLAGS -DBGMM_PROFILE -c kernels_gram.hip -o _obj/kernels_gram_prof.o
OBJS=$(ls _obj/*.hip.o | grep -v kernels_gram.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbgmm_hip_prof.so $OBJS _obj/kernels_gram_prof.o
echo built ../libbgmm_hip_prof.so
