for f in "-DPN_GENERIC_FLAT_WEIGHTS" ""; do
  POPNET_EXTRA_HIPCC_FLAGS="$f" python3 -c "
import sys; sys.path.insert(0,'.')
import importlib; b=importlib.import_module('popnet_amd.build'); b.build(force=True, verbose=False)" > /dev/null 2>&1
  for i in 1 2; do python3 bench.py --no-cpu-baseline --steps 400 --net yolo 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('yolo [%s]' % sys.argv[1], d['value'], r['conv_stack']['achieved'], r['conv_stack']['tflops_inside_timed_region'])" "$f"; done
  python3 bench.py --no-cpu-baseline --steps 400 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('rtpose [%s]' % sys.argv[1], d['value'], r['conv_stack']['achieved'])" "$f"
  python3 bench.py --no-cpu-baseline --steps 60 --precision fp32 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('fp32 [%s]' % sys.argv[1], d['value'], r['conv_stack']['achieved'])" "$f"
done
