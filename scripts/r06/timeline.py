"""One training step as a timeline: every launch between two sgd_nesterov launches of a rocprofv3 --kernel-trace csv, with its start offset, duration and queue, plus the
time during which only one / both queues hold a running kernel.  usage: timeline.py <dir> [step index] [marker kernel prefix = sgd_nesterov] [steps to show = 1]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?")) for r in csv.DictReader(open(f))]
rows.sort()
marker = sys.argv[3] if len(sys.argv) > 3 else "sgd_nesterov"
span = int(sys.argv[4]) if len(sys.argv) > 4 else 1
marks = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
a, b = marks[which] + 1, marks[which + span] + 1
step = rows[a:b]
t0 = rows[marks[which]][1]
qs = sorted(set(r[3] for r in step))
print("step %d: %d launches, %.3f ms from the end of the previous update to the end of this one; queues %s" % (which, len(step), (step[-1][1] - t0) / 1e6, qs))
for s, e, n, q, gx, gy in step:
    short = n.replace("void ", "").replace("tx::", "")[:58]
    print("%9.1f %8.1f  q%-2s %s%s  grid(%s,%s)" % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q), "        " * qs.index(q), short, gx, gy))
# occupancy of the time line by queue
ev = []
for s, e, n, q, gx, gy in step:
    ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort()
run = {q: 0 for q in qs}
last = t0
acc = {}
for t, d, q in ev:
    key = tuple(sorted(k for k, v in run.items() if v > 0))
    acc[key] = acc.get(key, 0) + (t - last)
    last = t
    run[q] += d
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-30s %8.3f ms" % ("+".join("q%d" % qs.index(q) for q in k) or "idle", v / 1e6))
