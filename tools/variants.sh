#!/bin/bash
# time home_kernel of each library variant at a BASELINE shape:  tools/variants.sh "base notail ..." N D K [rounds]
VARS=$1; N=${2:-1000000}; D=${3:-64}; K=${4:-200}; R=${5:-2}
for r in $(seq $R); do for v in $VARS; do
  printf "%-10s " $v; python tools/probe.py chain $N $D $K --init true --prune 3 --lib $v --sweeps 40 --timing 2>&1 | grep "timed launches"
done; done
