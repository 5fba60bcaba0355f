"""rocprofv3 kernel trace (csv) -> start_us,dur_us,stream,name of the last `n` dispatches:  trace_compact.py in.csv out.csv [n]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for d in csv.DictReader(f):
        rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Stream_Id"], d["Kernel_Name"].split("(")[0].replace("void ", "")[:40]))
rows.sort()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
rows = rows[-n:]
t0 = rows[0][0]
with open(sys.argv[2], "w") as f:
    for s, e, st, name in rows:
        f.write("%.2f,%.2f,%s,%s\n" % ((s - t0) / 1e3, (e - s) / 1e3, st, name))
