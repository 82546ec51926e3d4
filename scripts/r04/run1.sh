#!/bin/bash
# round 4, GPU call 1: bb64x3_kernel parity + timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_1
O=gpurun_out/r04_1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16x3_fused or optional_kernel_variants" > $O/pytest_bbx3.log 2>&1; echo "rc $?" >> $O/pytest_bbx3.log
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -k "locked_engine" > $O/pytest_lock.log 2>&1; echo "rc $?" >> $O/pytest_lock.log
timeout 120 ./popnet_amd/build/bbx3lab 32 112 112 100 > $O/bbx3lab.log 2>&1
timeout 120 ./popnet_amd/build/bbx3lab_stamp 32 112 112 20 >> $O/bbx3lab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_precision.py -x -q > $O/pytest_precision.log 2>&1; echo "rc $?" >> $O/pytest_precision.log
timeout 600 python bench.py --precision bf16x3 --no-extras --no-cpu-baseline --no-h2d --reps 3 > $O/bench_x3.json 2> $O/bench_x3.err
tail -3 $O/pytest_bbx3.log $O/pytest_lock.log $O/pytest_precision.log; cat $O/bbx3lab.log; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04_1/bench_x3.json') if l.startswith('{')][-1])
    print('x3 value', d['value'], d['value_stat']['runs']); 
    for k in d['roofline']['conv_stack']['by_kernel']: print(k)
except Exception as e: print('bench parse failed', e)
PY
