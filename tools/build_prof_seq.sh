#!/bin/bash
# libbgmm_hip_prof.so with the phase clocks of sweep_seq_kernel compiled in (tools/probe.py chain 100000 2 20 --prof)
set -e
cd "$(dirname "$0")/../pybgmm_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
hipcc $FLAGS -DBGMM_SEQ_PROF -c kernels_seq.hip -o _obj/kernels_seq_prof.o
OBJS=$(ls _obj/*.hip.o | grep -v kernels_seq.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbgmm_hip_prof.so $OBJS _obj/kernels_seq_prof.o
echo built ../libbgmm_hip_prof.so
