# rocprofv3 kernel stats of the YoloPoseNet leg, eager on one stream (alone-times per kernel)
mkdir -p gpurun_out/yolo_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/yolo_prof -o runc -- python3 bench.py --net yolo --no-cpu-baseline --no-extras --no-h2d --reps 1 --no-graph --pipeline 1 --steps 20 --warmup 5 > gpurun_out/yolo_prof/log.txt 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/yolo_prof/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print('%-100s %5s avg %8.1f us  %5.1f%%'%(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
tail -2 gpurun_out/yolo_prof/log.txt | cut -c1-600
