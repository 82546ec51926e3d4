#!/bin/bash
# timing-only: what is the pipelined throughput if a class of small launches costs nothing? (POPNET_ABLATE_SKIP, net.hip)
run() { env "$@" python bench.py --no-extras --no-cpu-baseline --no-h2d --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value', d['value'], 'ms/step', d['ms_per_step'])"; }
echo "== baseline"; run X=1
for c in pool head c1x1 c3 stem "pool,head,c1x1" "pool,head,c1x1,c3,stem" bb64 conv4; do echo "== skip $c"; run POPNET_ABLATE_SKIP=$c; done
