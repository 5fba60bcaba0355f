#!/bin/bash
# libbgmm_hip_<name>.so: the library with ONE source file (kernels_home.hip unless FILE=... is set) compiled with extra flags
# (development experiments:  tools/build_variant.sh notail -DHX_NOTAIL;  tools/probe.py chain N D K --prune 3 --lib notail;
#  FILE=kernels_perm.hip tools/build_variant.sh seg256 -DBGMM_PERM_SEG=256;  BGMM_LIB_VARIANT=seg256 python tools/permcheck.py)
set -e
NAME=$1; shift
FILE=${FILE:-kernels_home.hip}
BASE=${FILE%.hip}
cd "$(dirname "$0")/../pybgmm_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
mkdir -p _obj
hipcc $FLAGS "$@" -c $FILE -o _obj/${BASE}_$NAME.o
OBJS=$(ls _obj/*.hip.o | grep -v $FILE.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbgmm_hip_$NAME.so $OBJS _obj/${BASE}_$NAME.o
echo built ../libbgmm_hip_$NAME.so
