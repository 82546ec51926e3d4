#!/bin/bash
# h2d_inclusive fraction under runtime copy-path settings (same box, back to back).  usage: h2d_env.sh <tag>
O=gpurun_out/$1; mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-power > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
h=d['h2d_inclusive']
print("%-28s value %8.0f  h2d %8.0f  frac %.4f  link %s GB/s" % (sys.argv[2], d['value'], h['value'], h['fraction_of_value'], h['host_link']['GBps']))
PY
}
run default A=1
run sdma_off HSA_ENABLE_SDMA=0
run queues4 GPU_MAX_HW_QUEUES=4
run queues16 GPU_MAX_HW_QUEUES=16
run default2 A=1
