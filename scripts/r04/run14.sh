#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_14; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_precision.py -x -q -k "x3" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -n 4 $O/pytest.log
for i in 1 2; do
timeout 600 python bench.py --precision bf16x3 --no-extras --no-cpu-baseline --no-h2d --reps 3 --steps 100 > "$O/bench_x3_$i.json" 2> "$O/bench_x3_$i.err"
python - "$O/bench_x3_$i.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('x3 value', d['value'], d['value_stat']['runs'])
for k in d['roofline']['conv_stack']['by_kernel'][:4]: print('   ', k['kernel'], k['launches_per_step'], k['avg_launch_us'])
PY
done
cd popnet_amd/build; for a in "32 112 112 192 64 3 1 20 v3 1"; do ./convlab_stamp $a | grep "stamps"; done
