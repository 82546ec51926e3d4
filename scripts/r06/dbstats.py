"""Per-kernel totals of a rocprofv3 run: reads the *_kernel_stats.csv / kernel trace csv or the sqlite .db it wrote.  usage: dbstats.py <dir or file> [steps]"""
import csv
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = []
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
traces = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(path) else []
if traces:
    agg = {}
    for f in traces:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if os.environ.get("BY_GRID"):
                n = "%s  grid(%s,%s)" % (n[:60], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"))
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a = agg.setdefault(n, [0, 0])
            a[0] += 1
            a[1] += d
    rows = [(n, c, t, t / c) for n, (c, t) in agg.items()]
elif dbs:
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name").fetchall()
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print("total %.3f ms  (%.3f ms per step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for n, c, t, a in rows[:int(os.environ.get('TOP', '45'))]:
    print("%-84s %7.1f/step %9.3f ms/step %8.1f us %5.1f%%" % (n[:84], c / steps, t / 1e6 / steps, a / 1e3, 100 * t / tot))
