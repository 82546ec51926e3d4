#!/bin/bash
# round-4 final evidence: the driver's command, the default line, the whole GPU suite, smoke
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
S=$(date +%s); timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "driver-cmd bench wall $(( $(date +%s) - S )) s"
S=$(date +%s); timeout 1500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench wall $(( $(date +%s) - S )) s"
for f in bench_driver_cmd bench_default; do python - "$O/$f.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1].split('/')[-1], 'value', d['value'], 'h2d', d['h2d_inclusive']['value'], d['h2d_inclusive'].get('fraction_of_value'), d['h2d_inclusive'].get('host_link',{}).get('GBps'), 'frac', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'stack', d['roofline']['conv_stack']['frac'])
print('   parity', d['parity_mode'].get('value'), d['parity_mode'].get('roofline',{}).get('frac'), 'yolo', d['yolo'].get('value'), d['yolo'].get('conv_stack',{}).get('frac'), 'train', d['train_step'].get('ms_per_step'), d['train_step'].get('bf16x3',{}).get('ms_per_step'), 'cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('b15',{}).get('value'))
print('   legs', d.get('leg_seconds'))
PY
done
S=$(date +%s); timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log; echo "pytest wall $(( $(date +%s) - S )) s"; tail -n 3 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 1 $O/smoke.log
