#!/bin/bash
# does a fused-BasicBlock launch that leaves whole CUs to the other streams raise the pipelined throughput?
run() { env "$@" python bench.py --no-extras --no-cpu-baseline --no-h2d --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value', d['value'], 'ms/step', d['ms_per_step'], 'bb64 us/step', [k['us_per_step'] for k in d['roofline']['conv_stack']['by_kernel'] if k['kernel']=='bb64_kernel'])"; }
echo "== default (256 workgroups), pipeline 3"; run X=1
for c in 224 192 128; do echo "== POPNET_BB64_CUS=$c"; run POPNET_BB64_CUS=$c; done
echo "== POPNET_NO_BBLOCK=1"; run POPNET_NO_BBLOCK=1
