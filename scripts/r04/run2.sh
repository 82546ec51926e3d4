#!/bin/bash
# round 4, GPU call 2: bb64x3_kernel parity + ablations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_2; mkdir -p $O
for b in bbx3lab_stamp bbx3lab_NOWAIT_stamp bbx3lab_NODMA_A_stamp bbx3lab_NODMA_IN_stamp; do echo "== $b"; timeout 120 ./popnet_amd/build/$b 32 112 112 20; done > $O/bbx3lab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16x3_fused" > $O/pytest_bbx3.log 2>&1; echo "rc $?" >> $O/pytest_bbx3.log
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -k "locked_engine" > $O/pytest_lock.log 2>&1; echo "rc $?" >> $O/pytest_lock.log
timeout 900 python -m pytest tests/test_gpu_precision.py -x -q > $O/pytest_precision.log 2>&1; echo "rc $?" >> $O/pytest_precision.log
timeout 600 python bench.py --precision bf16x3 --no-extras --no-cpu-baseline --no-h2d --reps 3 > $O/bench_x3.json 2> $O/bench_x3.err
for f in $O/pytest_bbx3.log $O/pytest_lock.log $O/pytest_precision.log; do echo "-- $f"; tail -n 4 $f; done; cat $O/bbx3lab.log; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04_2/bench_x3.json') if l.startswith('{')][-1])
    print('x3 value', d['value'], d['value_stat']['runs']); 
    for k in d['roofline']['conv_stack']['by_kernel']: print(k)
except Exception as e: print('bench parse failed', e)
PY
