"""GPU timeline of the last sweeps of a kernel trace:   python tools/timeline.py <rocprofv3 output dir> [n_rows]
Prints the last n_rows kernel dispatches (start offset, duration, gap to the previous end on the same queue, name) of a
`rocprofv3 --kernel-trace --output-format csv` run -- where the time of a sweep goes between its kernels."""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel_trace.csv under", root)
        return 1
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n:]
    t0 = int(rows[0]["Start_Timestamp"])
    last_end = {}
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get("Queue_Id", "0")
        gap = (s - last_end[q]) * 1e-3 if q in last_end else 0.0
        last_end[q] = e
        print("%10.1f us  +%8.1f us  gap %8.1f  q%-3s %s" % ((s - t0) * 1e-3, (e - s) * 1e-3, gap, q, r["Kernel_Name"].split("(")[0][:60]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
