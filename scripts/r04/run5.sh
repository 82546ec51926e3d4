#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_5; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_precision.py tests/test_gpu_parity.py -x -q -k "x3 or precision or frames_in or engines or deviation" > $O/pytest_x3.log 2>&1; echo "rc $?" >> $O/pytest_x3.log
tail -n 5 $O/pytest_x3.log
for sw in "" "POPNET_NO_BBLOCK=1"; do
  echo "== bench x3 [$sw]"
  env $sw timeout 600 python bench.py --precision bf16x3 --no-extras --no-cpu-baseline --no-h2d --reps 3 --steps 100 > "$O/bench_x3_$sw.json" 2> "$O/bench_x3_$sw.err"
  python - "$O/bench_x3_$sw.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print('x3 value', d['value'], d['value_stat']['runs'])
    for k in d['roofline']['conv_stack']['by_kernel'][:8]: print(k['kernel'], k['launches_per_step'], k['avg_launch_us'])
    print('stem/pool ms', d['roofline']['conv_stack']['stem_pool_ms_per_step'])
except Exception as e: print('bench parse failed', e)
PY
done
