#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_17; mkdir -p $O
for sw in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=16" "HSA_ENABLE_INTERRUPT=0" "X=1"; do
  env $sw timeout 600 python bench.py --no-extras --no-cpu-baseline --reps 3 --steps 100 > "$O/bench_$sw.json" 2> "$O/bench_$sw.err"
  python - "$O/bench_$sw.json" "$sw" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], 'value', d['value'], d['value_stat']['runs'], 'h2d', d['h2d_inclusive']['value'])
except Exception as e: print(sys.argv[2], 'failed', e)
PY
done
