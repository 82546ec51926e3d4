#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_12; mkdir -p $O
for pd in 3 2 4 3; do
  timeout 600 python bench.py --precision bf16x3 --no-extras --no-cpu-baseline --no-h2d --reps 3 --steps 100 --pipeline $pd > "$O/bench_x3_p$pd.json" 2> "$O/bench_x3_p$pd.err"
  python - "$O/bench_x3_p$pd.json" "$pd" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('x3 pipeline', sys.argv[2], 'value', d['value'], d['value_stat']['runs'])
PY
done
for pd in 3 4; do
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-h2d --reps 3 --steps 100 --pipeline $pd > "$O/bench_bf16_p$pd.json" 2> "$O/bench_bf16_p$pd.err"
  python - "$O/bench_bf16_p$pd.json" "$pd" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('bf16 pipeline', sys.argv[2], 'value', d['value'], d['value_stat']['runs'])
PY
done
