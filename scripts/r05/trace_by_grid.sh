#!/bin/bash
# per-(kernel, grid) mean durations of one train_bench run: trace_by_grid.sh <tag> <fp32|bf16x3> [kernel substring]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o runc -- python3 scripts/train_bench.py 32 4 $2 > $O/tr.log 2>&1
python3 - "$O" "${3:-tconv3}" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/tr/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        key = (r["Kernel_Name"].split("(")[0][-28:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["LDS_Block_Size"] if "LDS_Block_Size" in r else "")
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-30s grid %6s %4s %4s lds %6s  n %4d  mean %8.1f us  total %9.1f us" % (k[0], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v), sum(v)))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
