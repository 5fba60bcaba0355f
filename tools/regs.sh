#!/bin/bash
# register / scratch usage of every kernel of one source file:  tools/regs.sh kernels_home.hip [extra flags]
F=$1; shift
cd "$(dirname "$0")/../pybgmm_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -Rpass-analysis=kernel-resource-usage -c $F -o /tmp/regs_$$.o 2>&1 | python3 -c '
import sys,re
cur=None; rows={}
for l in sys.stdin:
    if "error" in l: print(l.rstrip())
    m=re.search(r"remark:\s+(.*?): (\S+) \[-Rpass", l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=="Function Name": cur=v; rows[cur]={}
    elif cur: rows[cur][k]=v
for n,r in rows.items():
    print("%-60s VGPR %3s AGPR %3s scratch %4s occ %s sgpr-spill %3s vgpr-spill %3s LDS %s" % (n[:60], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
'
rm -f /tmp/regs_$$.o
