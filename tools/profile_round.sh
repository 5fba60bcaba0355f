#!/bin/bash
# Regenerates the evidence under profiles/<tag>/ on a GPU box:
#   tools/profile_round.sh r02            (run through gpurun; outputs land in gpurun_out/<tag>/)
# 1. the plain bench line (it runs its own rocprofv3 --pmc passes for roofline.traffic),
# 2. rocprofv3 --kernel-trace --stats of the same command (without the burn-in leg and the PMC children),
# 3. rocprofv3 --kernel-trace --stats of the burn-in regime alone (C4 from a random start, two sweeps),
# 4. the other BASELINE shapes' bench lines.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
python bench.py > "$OUT/bench_C4.json" 2> "$OUT/bench_C4.err"
tail -c 400 "$OUT/bench_C4.json"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- \
    python $REPO/bench.py --cpu-visits 0 --no-burnin --no-pmc > "$OUT/bench_C4_under_rocprof.json" 2> "$OUT/kt.err"
cp "$OUT"/kt/kt_kernel_stats.csv "$OUT/kernel_stats_C4.csv" 2>/dev/null || cp $(find "$OUT/kt" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats_C4.csv"
rocprofv3 --kernel-trace --stats -d "$OUT/ktb" -o kt --output-format csv -- \
    python $REPO/tools/gram_probe.py 1000000 64 200 2 rand 0 0 > "$OUT/burnin_C4_under_rocprof.log" 2> "$OUT/ktb.err"
cp $(find "$OUT/ktb" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats_C4_burnin.csv"
cd $REPO
for WL in C3 C5 C2; do
    python bench.py --workload $WL --no-pmc > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err"
    tail -c 300 "$OUT/bench_$WL.json"
done
rm -rf "$OUT/kt" "$OUT/ktb"
