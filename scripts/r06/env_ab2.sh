#!/bin/bash
# same-box round-robin of the training step under several environment settings.  usage: env_ab2.sh <tag> <precision> "<VAR=v ...>" "<VAR=v ...>" ...   ("-" = default)
tag=$1; prec=$2; shift 2; O=gpurun_out/$tag; mkdir -p $O
[ -n "$NOTEST" ] || python -m pytest tests/test_gpu_train.py -q -m gpu -k "exact_restatements or ragged" 2>&1 | tail -3
for r in 1 2 3; do
  for sw in "$@"; do
    if [ "$sw" = "-" ]; then e=""; else e="$sw"; fi
    env $e python scripts/train_bench.py 32 50 $prec 2>&1 | grep -v "host enqueue" | tail -1 | cut -c1-40 | sed "s/^/[$sw]  /"
  done
done | tee $O/env_ab2_$prec.txt
