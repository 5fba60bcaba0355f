"""How much of a kernel trace ran side by side:   python tools/overlap.py <dir with *kernel_trace.csv> [name substring ...]
Per kernel name: launches, summed duration, and -- over the whole trace -- the wall-clock span, the time at least one kernel
ran, and the sum of durations (sum / busy = average number of kernels in flight)."""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].split(" ")[-1]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in iv)
print("span %.1f ms, busy %.1f ms (%.0f %%), sum of kernel durations %.1f ms -> %.2f kernels in flight while busy" % (
    (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), tot / 1e6, tot / busy))
by = defaultdict(list)
for s, e, n in iv:
    by[n].append(e - s)
for n, ds in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    ds.sort()
    print("  %-40s %7d launches  %9.1f ms  avg %8.1f  median %8.1f  p90 %8.1f us" % (
        n[:40], len(ds), sum(ds) / 1e6, sum(ds) / len(ds) / 1e3, ds[len(ds) // 2] / 1e3, ds[9 * len(ds) // 10] / 1e3))
