run() { python3 bench.py --no-cpu-baseline --pipeline $1 --steps 400 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.readline()); print('pipeline', sys.argv[1], 'hwq', os.environ.get('GPU_MAX_HW_QUEUES'), d['value'], d['roofline']['conv_stack']['tflops_inside_timed_region'])" $1; }
for p in 5 7; do run $p; done
export GPU_MAX_HW_QUEUES=8
for p in 3 4 6 8; do run $p; done
export GPU_MAX_HW_QUEUES=2
for p in 3 4; do run $p; done
