# builds the library with each epilogue mode ON THE GPU BOX and benches it (hipcc is in the image)
for m in 0 1 2; do
  POPNET_EXTRA_HIPCC_FLAGS="-DPN_CONV3_FAST_EPILOGUE=$m" python3 -c "
import sys; sys.path.insert(0,'.')
import importlib; b=importlib.import_module('popnet_amd.build'); b.build(force=True, verbose=False)" > /dev/null 2>&1
  for i in 1 2; do python3 bench.py --no-cpu-baseline --steps 400 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('mode', sys.argv[1], d['value'], r['achieved'], r['frac'], r['conv_stack']['achieved'], r['conv_stack']['tflops_inside_timed_region'])" $m; done
done
