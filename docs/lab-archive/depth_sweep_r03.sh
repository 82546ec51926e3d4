# frames/s against batches in flight (bench.py --pipeline) and against GPU_MAX_HW_QUEUES, round-3 kernels
run() { python3 bench.py --no-cpu-baseline --no-extras --no-h2d --reps 3 --pipeline $1 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.readlines()[-1]); print('pipeline', sys.argv[1], 'hwq', os.environ.get('GPU_MAX_HW_QUEUES'), d['value'], d['value_stat']['runs'])" $1; }
for p in 1 2 3 4 6; do run $p; done
for q in 2 8; do export GPU_MAX_HW_QUEUES=$q; for p in 3 4; do run $p; done; done
