#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_11; mkdir -p $O
for sw in "" "POPNET_CONV3_PT14=1" "POPNET_CONV3_NBUF2=1" "POPNET_CONV4=0" ""; do
  env $sw timeout 600 python bench.py --precision bf16x3 --no-extras --no-cpu-baseline --no-h2d --reps 3 --steps 100 > "$O/bench_x3_$sw.json" 2> "$O/bench_x3_$sw.err"
  python - "$O/bench_x3_$sw.json" "$sw" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2] or 'default', 'value', d['value'], d['value_stat']['runs'])
for k in d['roofline']['conv_stack']['by_kernel'][:4]: print('   ', k['kernel'], k['launches_per_step'], k['avg_launch_us'])
PY
done
