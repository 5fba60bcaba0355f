#!/bin/bash
# libbgmm_hip_tsan.so: the host side of the library (the six api_*.hip files: contexts, the permutations' worker thread,
# the chains' host threads and their rendezvous, the two-stream pipelines) compiled with -fsanitize=thread; the kernels'
# objects as built.  Run with the sanitizer's runtime preloaded (the interpreter is not instrumented):
#   tools/build_tsan.sh && LD_PRELOAD=$(tools/build_tsan.sh --rt) TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" \
#       python tools/tsan_run.py 2> profiles/r06/tsan_host.log
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.tsan-x86_64.so
if [ "$1" = "--rt" ]; then echo $RT; exit 0; fi
set -e
cd "$(dirname "$0")/../pybgmm_amd/csrc"
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fsanitize=thread -Wno-option-ignored"
mkdir -p _obj
for f in api_context api_inputs api_perm api_sweep api_group api_comm; do
  hipcc $FLAGS -c $f.hip -o _obj/${f}_tsan.o &
done
wait
OBJS=$(ls _obj/kernels_*.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -shared-libsan -o ../libbgmm_hip_tsan.so _obj/api_*_tsan.o $OBJS
echo built ../libbgmm_hip_tsan.so
