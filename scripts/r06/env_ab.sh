#!/bin/bash
# same-box ABAB of the training step under one environment switch.  usage: env_ab.sh <tag> <VAR=value> [precisions]
tag=$1; sw=$2; O=gpurun_out/$tag; mkdir -p $O
[ -n "$NOTEST" ] || python -m pytest tests/test_gpu_train.py -q -m gpu -k "exact_restatements or ragged" 2>&1 | tail -3
for prec in ${3:-bf16x3 fp32}; do
  for r in 1 2 3; do
    env $sw python scripts/train_bench.py 32 50 $prec 2>&1 | grep -v "host enqueue" | tail -1 | cut -c1-70 | sed "s/^/$sw  /"
    python scripts/train_bench.py 32 50 $prec 2>&1 | grep -v "host enqueue" | tail -1 | cut -c1-70 | sed "s/^/default  /"
  done
done | tee $O/env_ab.txt
