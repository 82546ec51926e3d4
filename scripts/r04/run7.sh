#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_7; mkdir -p $O
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - "$O/bench_default.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value', d['value'], d['value_stat']['runs'], 'h2d', d.get('h2d_inclusive',{}).get('value'))
print('roofline', d['roofline']['kernel'][:40], d['roofline']['frac'], d['roofline']['avg_launch_us'], 'stack', d['roofline']['conv_stack']['frac'])
for k in d['roofline']['conv_stack']['by_kernel'][:10]: print('  ', k['kernel'], k['launches_per_step'], k['avg_launch_us'], k['tflops'])
print('postproc', d['postproc']['us_per_step'])
pm=d.get('parity_mode',{}); print('parity_mode', pm.get('value'), pm.get('roofline',{}).get('frac'), pm.get('h2d_inclusive',{}).get('value') if pm.get('h2d_inclusive') else None)
y=d.get('yolo',{}); print('yolo', y.get('value'), y.get('conv_stack',{}).get('frac'))
print('rccl', d.get('rccl_check',{}).get('value_with_process_group'), d.get('rccl_check',{}).get('gather_records_ok'))
t=d.get('train_step',{}); print('train', t.get('ms_per_step'), t.get('bf16x3',{}).get('ms_per_step'))
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('b15'))
print('env', d.get('env'))
PY
echo "== bf16 A/B: fused bb64 vs two launches"
for sw in "" "POPNET_NO_BBLOCK=1"; do
  env $sw timeout 600 python bench.py --no-extras --no-cpu-baseline --no-h2d --reps 3 --steps 100 > "$O/bench_bf16_$sw.json" 2> "$O/bench_bf16_$sw.err"
  python - "$O/bench_bf16_$sw.json" "$sw" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2] or 'default', 'value', d['value'], d['value_stat']['runs'])
PY
done
