"""Prints a compact per-kernel table from a rocprofv3 kernel_stats CSV (bench.py run)."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = list(csv.DictReader(open(path)))
tot = 0.0
for r in rows:
    name = r['Name']
    if any(k in name for k in ('conv_mfma', 'stem7x7', 'pool_kernel', 'peaks_refine', 'limb_match', 'group_readout', 'preprocess_kernel')):
        if 'conv_mfma_kernel<0' in name or 'IfEvPK' in name or '<float' in name:
            continue
        us = float(r['TotalDurationNs']) / 1e3 / steps
        tot += us
        print("%-70s calls/step=%5.1f us/step=%8.1f avg_us=%7.2f" % (name[:70], int(r['Calls']) / steps, us, float(r['AverageNs']) / 1e3))
print("total GPU us/step (bf16 path kernels): %.1f" % tot)
