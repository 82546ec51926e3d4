#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_4; mkdir -p $O
for b in bbx3lab_stamp bbx3lab_A_G0_stamp bbx3lab_LAGA_stamp bbx3lab_LAGA_A_G0_stamp; do echo "== $b"; timeout 120 ./popnet_amd/build/$b 32 112 112 20; done > $O/bbx3lab.log 2>&1
cat $O/bbx3lab.log
for sw in "" "POPNET_NO_BBLOCK=1"; do
  echo "== bench x3 [$sw]"
  env $sw timeout 600 python bench.py --precision bf16x3 --no-extras --no-cpu-baseline --no-h2d --reps 3 --steps 100 > $O/bench_x3_$sw.json 2> $O/bench_x3_$sw.err
  python - "$O/bench_x3_$sw.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print('x3 value', d['value'], d['value_stat']['runs'])
    for k in d['roofline']['conv_stack']['by_kernel'][:3]: print(k['kernel'], k['launches_per_step'], k['avg_launch_us'])
except Exception as e: print('bench parse failed', e)
PY
done
