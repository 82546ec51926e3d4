#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_10; mkdir -p $O
S=$(date +%s); timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench wall $(( $(date +%s) - S )) s"
python - "$O/bench_driver_cmd.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value', d['value'], 'h2d', d['h2d_inclusive']['value'], d['h2d_inclusive'].get('fraction_of_value'), 'frac', d['roofline']['frac'], 'parity', d['parity_mode'].get('value'), 'yolo', d['yolo'].get('value'))
print('legs', d.get('leg_seconds'))
PY
S=$(date +%s); timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log; echo "pytest wall $(( $(date +%s) - S )) s"; tail -n 4 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 2 $O/smoke.log
