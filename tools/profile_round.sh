#!/bin/bash
# Regenerates the evidence under profiles/<tag>/ on a GPU box:
#   tools/profile_round.sh r04            (run through gpurun; outputs land in gpurun_out/<tag>/)
# 1. the plain bench line (it runs its own rocprofv3 --pmc passes for roofline.traffic),
# 2. rocprofv3 --kernel-trace --stats of the same command without the burn-in / steady_moving legs and the PMC children
#    (the at-rest chain alone: what roofline.avg_launch_ms has to agree with),
# 3. kernel trace of the steady_moving regime alone (C4 shape, mu_scale 0.55: 0.5 % of the visits move at equilibrium),
#    and one --pmc SQ pass over its resolver / proof-pass kernels,
# 4. kernel trace of the burn-in regime alone (C4 from a random start, two sweeps),
# 5. the other BASELINE shapes' bench lines.
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
python bench.py --keep-pmc "$OUT" > "$OUT/bench_C4.json" 2> "$OUT/bench_C4.err"
tail -c 400 "$OUT/bench_C4.json"
cd /tmp
stats_of() { cp "$(find "$1" -name "*kernel_stats.csv" | head -1)" "$2"; }
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- \
    python $REPO/bench.py --cpu-visits 0 --numpy-visits 0 --no-burnin --no-moving --no-pmc > "$OUT/bench_C4_under_rocprof.json" 2> "$OUT/kt.err"
stats_of "$OUT/kt" "$OUT/kernel_stats_C4.csv"
rocprofv3 --kernel-trace --stats -d "$OUT/ktm" -o kt --output-format csv -- \
    python $REPO/tools/probe.py chain 1000000 64 200 --init true --sep 0.55 --sweeps 4 > "$OUT/moving_C4_under_rocprof.log" 2> "$OUT/ktm.err"
stats_of "$OUT/ktm" "$OUT/kernel_stats_C4_moving.csv"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
    --kernel-trace -d "$OUT/pmcm" -o p --output-format csv -- \
    python $REPO/tools/probe.py chain 1000000 64 200 --init true --sep 0.55 --sweeps 3 > /dev/null 2> "$OUT/pmcm.err"
python $REPO/tools/summarize_pmc.py "$OUT/pmcm" gram_resolve_kernel gram_kernel gram_finish_kernel safe_choice_kernel safe_ftab_kernel home_kernel \
    > "$OUT/pmc_sq_C4_moving.txt" 2>> "$OUT/pmcm.err"
rocprofv3 --kernel-trace --stats -d "$OUT/ktb" -o kt --output-format csv -- \
    python $REPO/tools/probe.py chain 1000000 64 200 --init rand --sweeps 2 > "$OUT/burnin_C4_under_rocprof.log" 2> "$OUT/ktb.err"
stats_of "$OUT/ktb" "$OUT/kernel_stats_C4_burnin.csv"
rocprofv3 --kernel-trace --stats -d "$OUT/kt3" -o kt --output-format csv -- \
    python $REPO/bench.py --workload C3 --steps 300 --cpu-visits 0 --numpy-visits 0 --no-burnin --no-moving --no-pmc --no-class-api > "$OUT/bench_C3_under_rocprof.json" 2> "$OUT/kt3.err"
stats_of "$OUT/kt3" "$OUT/kernel_stats_C3.csv"
rocprofv3 --kernel-trace --stats -d "$OUT/ktb5" -o kt --output-format csv -- \
    python $REPO/tools/probe.py chain 2000000 128 200 --init rand --sweeps 1 --pcrp > "$OUT/burnin_C5_under_rocprof.log" 2> "$OUT/ktb5.err"
stats_of "$OUT/ktb5" "$OUT/kernel_stats_C5_burnin.csv"
cd $REPO
for WL in C3 C5 C2; do
    python bench.py --workload $WL --steps 300 --keep-pmc "$OUT" > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err"
    tail -c 300 "$OUT/bench_$WL.json"
done
# 6. SQ counters of home_kernel at rest (C4, C3) and the FP64 matrix pipe's sustained rate
for WL in C4 C3; do bash tools/pmc_home.sh $WL > "$OUT/pmc_sq_home_$WL.txt" 2>&1; done
[ -x tools/mfma_f64_peak ] && tools/mfma_f64_peak > "$OUT/mfma_f64_peak.txt" 2>&1
rm -rf "$OUT/kt" "$OUT/ktb" "$OUT/ktm" "$OUT/pmcm" "$OUT/kt3" "$OUT/ktb5"
