#!/bin/bash
# Regenerates the evidence under profiles/<tag>/ on a GPU box:
#   tools/profile_round.sh r01            (run through gpurun; outputs land in gpurun_out/<tag>/)
# 1. the plain bench line, 2. rocprofv3 --kernel-trace --stats of the same command,
# 3. separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ busy counters) as MI355X_MICROARCH.md
#    prescribes, 4. tools/summarize_pmc.py -> pmc_summary.json + traffic_<workload>.json
set -u
TAG=${1:-r01}
WL=${2:-C4}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py --workload $WL > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err"
tail -c 600 "$OUT/bench_$WL.json"
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- \
    python $REPO/bench.py --workload $WL --cpu-visits 0 > "$OUT/bench_${WL}_under_rocprof.json" 2> "$OUT/kt.err"
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$C" -o pmc --output-format csv -- \
        python $REPO/bench.py --workload $WL --steps 2 --warmup 1 --cpu-visits 0 --no-kernel-timing \
        > "$OUT/pmc_$C.log" 2>&1
    # pruning without certified stays: every visit streamed by the pruning kernel
    rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmcprune_$C" -o pmc --output-format csv -- \
        python $REPO/bench.py --workload $WL --steps 2 --warmup 1 --cpu-visits 0 --no-kernel-timing --prune 3 \
        > "$OUT/pmcprune_$C.log" 2>&1
    # the same with pruning off: every (visit, component) pair through the full-evaluation kernel
    rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmcfull_$C" -o pmc --output-format csv -- \
        python $REPO/bench.py --workload $WL --steps 2 --warmup 1 --cpu-visits 0 --no-kernel-timing --prune 1 \
        > "$OUT/pmcfull_$C.log" 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace \
    -d "$OUT/pmc_SQ" -o pmc --output-format csv -- \
    python $REPO/bench.py --workload $WL --steps 2 --warmup 1 --cpu-visits 0 --no-kernel-timing \
    > "$OUT/pmc_SQ.log" 2>&1
cd $REPO
python tools/summarize_pmc.py "$OUT" $WL
