# POPNET_CONV3_PT14=2 in the network: bit-exactness against the generic kernel, then frames/s
run() { python3 bench.py --no-cpu-baseline --steps 400 $* 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('PT14=%s' % os.environ.get('POPNET_CONV3_PT14'), sys.argv[1:], d['value'], r['kernel'], r['achieved'], r['conv_stack']['achieved'], r['conv_stack']['tflops_inside_timed_region'])" $*; }
POPNET_CONV3_PT14=2 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv3 or hipgraph or permutation or ragged" 2>&1 | tail -3
run
POPNET_CONV3_PT14=2 run
run --pipeline 1
POPNET_CONV3_PT14=2 run --pipeline 1
POPNET_CONV3_PT14=2 run --pipeline 4
POPNET_CONV3_PT14=2 run --pipeline 6
