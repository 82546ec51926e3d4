#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_9; mkdir -p $O
S=$(date +%s); timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
echo "wall $(( $(date +%s) - S )) s"
python - "$O/bench_driver_cmd.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value', d['value'], d['value_stat']['runs'], 'h2d', d['h2d_inclusive']['value'], d['h2d_inclusive'].get('fraction_of_value'), d['h2d_inclusive'].get('host_link'))
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'stack', d['roofline']['conv_stack']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
pm=d.get('parity_mode',{}); print('parity_mode', pm.get('value'), pm.get('roofline',{}).get('frac'))
y=d.get('yolo',{}); print('yolo', y.get('value'), y.get('conv_stack',{}).get('frac'))
print('rccl', d.get('rccl_check',{}).get('value_with_process_group'), d.get('rccl_check',{}).get('gather_records_ok'), d.get('rccl_check',{}).get('flat_gradient_allreduce_ok'))
t=d.get('train_step',{}); print('train', t.get('ms_per_step'), t.get('bf16x3',{}).get('ms_per_step'))
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('b15',{}).get('value'))
print('fidelity', d.get('fidelity'))
print('env', d.get('env')); print('pg', d.get('process_group'))
PY
