"""Kernel concurrency of a rocprofv3 --kernel-trace run: share of the wall time with 0 / 1 / 2 / 3+ kernels in flight."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
n = len(ev)
seg = ev[int(n * 0.30):int(n * 0.45)]
pts = sorted([(s, 1) for s, e in seg] + [(e, -1) for s, e in seg])
t0, t1 = seg[0][0], max(e for s, e in seg)
cur, last, conc = 0, t0, {}
for t, d in pts:
    conc[cur] = conc.get(cur, 0) + t - last
    cur += d
    last = t
print("window %.1f ms, %d kernels: " % ((t1 - t0) / 1e6, len(seg)) + "  ".join("%d in flight %.1f%%" % (k, 100 * v / (t1 - t0)) for k, v in sorted(conc.items())))
