set -x
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "strip_kernel_equals or optional_kernel_variants or hipgraph or forward or golden" 2>&1 | tail -5
for i in 1 2; do
POPNET_NO_EMBED1X1=1 timeout 300 python3 bench.py --no-extras --no-cpu-baseline --no-h2d --reps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('NOEMBED', d['value'], d['roofline']['conv_stack']['ms_per_step'], d['roofline']['conv_stack']['launches_per_step'])"
timeout 300 python3 bench.py --no-extras --no-cpu-baseline --no-h2d --reps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('EMBED', d['value'], d['roofline']['conv_stack']['ms_per_step'], d['roofline']['conv_stack']['launches_per_step'])"
done
