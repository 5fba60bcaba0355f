"""The idle stretches of a rocprofv3 --kernel-trace CSV: where no kernel at all runs for more than `min_us`, with the kernels on
either side.   python tools/trace_gaps.py trace.csv 500"""
import csv, sys
f, min_us = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
busy_until, last = 0, None
tot = 0.0
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if busy_until and s - busy_until > min_us * 1e3:
        out.append(((s - busy_until) / 1e3, (busy_until - t0) / 1e6, last["Kernel_Name"][:40], r["Kernel_Name"][:40]))
        tot += (s - busy_until) / 1e3
    if e > busy_until:
        busy_until, last = e, r
print("%d idle stretches > %.0f us, %.1f ms in all, of %.1f ms traced" % (len(out), min_us, tot / 1e3, (busy_until - t0) / 1e6))
for g in sorted(out, reverse=True)[:12]:
    print("%9.1f us idle at %8.3f ms: after %-40s before %s" % g)
