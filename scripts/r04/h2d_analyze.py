"""Timeline of the PCIe hand-over: reads rocprofv3's kernel_trace / memory_copy_trace CSVs of a bench run and prints, for the LAST
timed region (the h2d one), the H2D copies (size, duration, GB/s, gap to the previous copy) and how the kernels of the steps sit
against them."""
import csv, glob, sys, collections
root = sys.argv[1]
def load(pat):
    rows = []
    for p in glob.glob(root + "/**/*" + pat, recursive=True):
        rows += list(csv.DictReader(open(p)))
    return rows
cp = load("memory_copy_trace.csv")
kn = load("kernel_trace.csv")
print("copies:", len(cp), "kernels:", len(kn))
if cp:
    print("copy columns:", list(cp[0].keys()))
big = [r for r in cp if int(r.get("Size", r.get("size", 0)) or 0) > 10_000_000 and "HOST_TO_DEVICE" in (r.get("Direction", "") + r.get("Name", "")).upper()]
print("H2D copies > 10 MB:", len(big))
big.sort(key=lambda r: int(r["Start_Timestamp"]))
last = big[-40:]
prev_end = None
durs, gaps = [], []
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    durs.append(d)
    if prev_end is not None:
        gaps.append((s - prev_end) / 1e3)
    prev_end = e
if durs:
    import statistics as st
    sz = int(last[0].get("Size", last[0].get("size")))
    print("last %d H2D copies: %.1f MB each, duration median %.1f us (%.1f GB/s), min %.1f max %.1f; gap between copies median %.1f us, max %.1f" % (
        len(last), sz / 1e6, st.median(durs), sz / st.median(durs) / 1e3, min(durs), max(durs), st.median(gaps) if gaps else 0, max(gaps) if gaps else 0))
    t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
    print("span of those copies: %.3f ms -> %.1f us per batch" % ((t1 - t0) / 1e6, (t1 - t0) / 1e3 / len(last)))
    # kernels inside the span, by queue/stream
    ks = [r for r in kn if t0 <= int(r["Start_Timestamp"]) <= t1]
    byq = collections.defaultdict(list)
    for r in ks:
        byq[r.get("Queue_Id", r.get("Stream_Id", "?"))].append(r)
    for q, rows in sorted(byq.items()):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
        print("  queue %s: %d kernels, busy %.1f us of %.1f (%.0f %%)" % (q, len(rows), busy, (t1 - t0) / 1e3, 100 * busy / ((t1 - t0) / 1e3)))
    names = collections.Counter(r["Kernel_Name"][:60] for r in ks)
    for n, c in names.most_common(12):
        tot = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in ks if r["Kernel_Name"][:60] == n) / 1e3
        print("   %-60s x%4d  total %.0f us" % (n, c, tot))
    # other copies in the span
    oth = [r for r in cp if t0 <= int(r["Start_Timestamp"]) <= t1 and r not in last]
    c2 = collections.Counter((r.get("Direction", r.get("Name", "?")), int(r.get("Size", r.get("size", 0)) or 0)) for r in oth)
    for (d, s), c in c2.most_common(8):
        tot = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in oth if (r.get("Direction", r.get("Name", "?")), int(r.get("Size", r.get("size", 0)) or 0)) == (d, s)) / 1e3
        print("   copy %s %d B x%d total %.0f us" % (d, s, c, tot))
