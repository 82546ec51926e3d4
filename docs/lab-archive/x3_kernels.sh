# per-kernel alone-times of the bf16x3 (parity mode) step, eager on one stream
python3 bench.py --precision bf16x3 --no-cpu-baseline --no-extras --no-h2d --reps 1 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
r=d['roofline']; print('conv stack ms', r['conv_stack']['ms_per_step'], 'stem+pool ms', r['conv_stack']['stem_pool_ms_per_step'], 'launches', r['conv_stack']['launches_per_step'])
for k in r['conv_stack']['by_kernel']: print(k)
print(d.get('postproc'))
"
