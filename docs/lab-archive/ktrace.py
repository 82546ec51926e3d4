"""Prints the kernel sequence of the last complete bench step from a rocprofv3 kernel_trace CSV."""
import csv, glob, sys
p = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'preprocess_kernel' in r['Kernel_Name']]
i0, i1 = idx[-3], idx[-2]
tot = 0
for r in rows[i0:i1]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print("%-62s grid=%7sx%s wg=%4s lds=%6s vgpr=%4s dur=%6.1f" % (r['Kernel_Name'][:62], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'], r['LDS_Block_Size'], r.get('VGPR_Count'), d))
print("sum of kernel durations: %.1f us" % tot)
