# frames/s against batch size per engine x engines in flight (experiment; the bench line is always B = 32)
run() { POPNET_BENCH_BATCH=$1 python3 bench.py --no-cpu-baseline --pipeline $2 --steps $3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('batch', sys.argv[1], 'pipeline', sys.argv[2], d['value'], d['roofline']['conv_stack']['achieved'], d['roofline']['conv_stack']['tflops_inside_timed_region'])" $1 $2; }
run 32 3 400
run 16 3 800
run 16 6 800
run 8 6 1600
run 64 2 200
run 64 3 200
run 96 3 150
