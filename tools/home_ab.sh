#!/bin/bash
# A/B of home_kernel variants (tools/build_variant.sh) on the three BASELINE shapes at rest, certified stays off:
#   tools/home_ab.sh "nopf w3 w3nopf"   -> gpurun_out/r06/home_ab.txt   (HIP-event time of the kernel's launches, last sweeps)
mkdir -p gpurun_out/r06
L=gpurun_out/r06/home_ab.txt
: > $L
run() { # shape-name lib args...
  local name=$1 lib=$2; shift 2
  echo "== $name lib=${lib:-default}" >> $L
  timeout 300 python tools/probe.py chain "$@" --init true --prune 3 --timing ${lib:+--lib $lib} 2>&1 | tail -2 >> $L
}
for rep in 1 2; do
for lib in "" nopf; do run C4 "$lib" 1000000 64 200 --sweeps 12; done
for lib in "" nopf w3 w3nopf; do run C3 "$lib" 1000000 16 100 --pcrp --sweeps 12; done
for lib in "" nopf; do run C5 "$lib" 2000000 128 200 --pcrp --sweeps 8; done
done
cat $L
