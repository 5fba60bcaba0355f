#!/bin/bash
# libbgmm_hip_<name>.so: the library with kernels_home.hip compiled with extra flags (development experiments:
#   tools/build_variant.sh notail -DHX_NOTAIL;  tools/probe.py chain N D K --prune 3 --lib notail)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../pybgmm_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
mkdir -p _obj
hipcc $FLAGS "$@" -c kernels_home.hip -o _obj/kernels_home_$NAME.o
OBJS=$(ls _obj/*.hip.o | grep -v kernels_home.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbgmm_hip_$NAME.so $OBJS _obj/kernels_home_$NAME.o
echo built ../libbgmm_hip_$NAME.so
