#!/bin/bash
# A/B of home_kernel variants (tools/build_variant.sh) at rest, certified stays off:
#   tools/home_ab.sh "w1s4 ..."  -> gpurun_out/r06/home_ab.txt   (HIP-event time of the kernel's launches, last sweep)
mkdir -p gpurun_out/r06
L=gpurun_out/r06/home_ab.txt
run() { # shape-name lib args...
  local name=$1 lib=$2; shift 2
  echo "== $name lib=${lib:-default}" >> $L
  timeout 300 python tools/probe.py chain "$@" --init true --prune 3 --timing ${lib:+--lib $lib} 2>&1 | tail -1 >> $L
}
for rep in 1 2; do
for lib in "" $1; do run C4 "$lib" 1000000 64 200 --sweeps 12; run C5 "$lib" 2000000 128 200 --pcrp --sweeps 6; done
done
tail -16 $L
