"""Print a window of a rocprofv3 kernel trace (csv) as a timeline: start/end in us relative to the window start."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:36], r.get("Stream_Id", "?")))
rows.sort()
n = len(rows)
lo = int(n * float(sys.argv[2])) if len(sys.argv) > 2 else n // 2
win = rows[lo:lo + (int(sys.argv[3]) if len(sys.argv) > 3 else 24)]
t0 = win[0][0]
for s, e, k, st in win:
    print("%9.1f -> %9.1f  (%8.1f us)  stream %s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, st, k))
