"""Timeline of a rocprofv3 --kernel-trace CSV: for kernels whose name contains one of the given substrings, the start / end
(us) of a stretch in the middle of the run, so that what overlaps with what can be read off.
    python tools/trace_overlap.py trace.csv 60 resolve_pgroup finish_pgroup cross_pgroup carry_pgroup"""
import csv, sys
f, n = sys.argv[1], int(sys.argv[2])
keys = sys.argv[3:]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in keys)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seg = rows[len(rows) // 2: len(rows) // 2 + n]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = next(k for k in keys if k in r["Kernel_Name"])
    print("%9.1f -> %9.1f  (%6.1f)  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
