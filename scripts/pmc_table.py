"""Aggregates rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch per kernel name
(only the LAST forward's dispatches of each kernel are kept: skips warm-up / fp32 calibration)."""
import csv, glob, sys, collections
root = sys.argv[1]
table = collections.defaultdict(dict)
for path in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in agg.items():
        table[k][c] = sum(v) / len(v)
        table[k]['_n'] = len(v)
keys = [k for k in table if any(s in k for s in sys.argv[2:])] if len(sys.argv) > 2 else list(table)
for k in sorted(keys):
    print(k[:100], "dispatches:", table[k].get('_n'))
    for c, v in sorted(table[k].items()):
        if c != '_n':
            print("    %-28s %.4g" % (c, v))
