#!/bin/bash
# SQ counters of home_kernel in one steady-state C4 sweep (certified stays off):  tools/pmc_home.sh [workload]
WL=${1:-C4}
export TMPDIR=/tmp
cd /tmp
run() {
  rm -rf /tmp/p1
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/p1 -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --inner-pmc --workload $WL --mode evaluated > /dev/null 2>&1
  python - <<EOF
import csv,glob,collections
f=glob.glob("/tmp/p1/**/*counter_collection.csv",recursive=True)
if not f: print("no counters"); raise SystemExit
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "home_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in sorted(acc.items()):
    print("%-28s %14.0f   (%d launches, %.1f us)" % (k, v[-1][0], len(v), v[-1][1]*1e-3))
EOF
}
run SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
run SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run TCC_HIT_sum TCC_MISS_sum
run TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
