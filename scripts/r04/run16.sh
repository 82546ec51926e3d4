#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_16; mkdir -p $O
for sw in "" "POPNET_CONV4=0" "POPNET_CONV4=1" ""; do
  env $sw timeout 600 python bench.py --no-extras --no-cpu-baseline --no-h2d --reps 3 --steps 100 > "$O/bench_$sw.json" 2> "$O/bench_$sw.err"
  python - "$O/bench_$sw.json" "$sw" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2] or 'default', 'value', d['value'], d['value_stat']['runs'], 'stack', d['roofline']['conv_stack']['frac'], d['roofline']['conv_stack']['launches_per_step'])
for k in d['roofline']['conv_stack']['by_kernel'][:4]: print('   ', k['kernel'], k['launches_per_step'], k['avg_launch_us'])
PY
done
