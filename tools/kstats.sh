#!/bin/bash
# per-kernel durations of a command under rocprofv3 --kernel-trace --stats:   tools/kstats.sh <tag> <command ...>
TAG=$1; shift
export TMPDIR=/tmp
D=/tmp/kstats_$TAG
rm -rf $D
( cd /tmp && rocprofv3 --kernel-trace --stats -d $D -o kt --output-format csv -- "$@" > /dev/null 2>&1 )
python - <<EOF
import csv, glob
f = glob.glob("$D/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no stats"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print("%-58s calls %4s  avg %10.1f us  min %10.1f  max %10.1f" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
EOF
