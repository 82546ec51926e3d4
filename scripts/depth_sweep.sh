# frames/s against engines in flight (bench.py --pipeline) and the HW queue limit
run() { python3 bench.py --no-cpu-baseline --pipeline $1 --steps 400 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.readline()); print('pipeline', sys.argv[1], 'hwq', os.environ.get('GPU_MAX_HW_QUEUES'), d['value'], d['roofline']['conv_stack']['tflops_inside_timed_region'])" $1; }
for p in 1 2 3 4 5 6 7; do run $p; done
