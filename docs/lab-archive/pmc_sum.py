"""Sums rocprofv3 --pmc counters over ALL dispatches of a run (whole-GPU view of a pipelined bench)."""
import csv, glob, sys, collections
tot = collections.Counter(); n = collections.Counter()
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot):
    print("%-32s sum %.6g over %d dispatches" % (k, tot[k], n[k]))
