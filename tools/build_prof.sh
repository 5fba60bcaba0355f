#!/bin/bash
# libbgmm_hip_prof.so: the library with the phase clocks of home_kernel compiled in (tools/probe.py chain N D K --prune 3 --prof)
set -e
cd "$(dirname "$0")/../pybgmm_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
mkdir -p _obj
hipcc $FLAGS -DBGMM_HOME_PROF -c kernels_home.hip -o _obj/kernels_home_prof.o
OBJS=$(ls _obj/*.hip.o | grep -v kernels_home.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbgmm_hip_prof.so $OBJS _obj/kernels_home_prof.o
echo built ../libbgmm_hip_prof.so
