python3 -m pytest tests -q -m gpu -x 2>&1 | tail -2
one() { python3 bench.py --no-cpu-baseline --steps 400 $* 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(sys.argv[1:], d['value'], r['achieved'], r['frac'], r['conv_stack']['achieved'], r['conv_stack']['tflops_inside_timed_region'])" $*; }
one; one
one --net yolo
python3 bench.py --no-cpu-baseline --steps 60 --precision fp32 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('fp32', d['value'], r['achieved'], r['conv_stack']['achieved'])"
