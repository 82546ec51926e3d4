#!/bin/bash
# planes stem (direct planes write / read) against the NCHW f32 hand-over: bit-identity test + same-box ABAB of the training step.  usage: stem_ab.sh <tag>
tag=$1; O=gpurun_out/$tag; mkdir -p $O
[ -n "$2" ] || python -m pytest tests/test_gpu_train.py -q -m gpu -k "exact_restatements or ragged" 2>&1 | tail -3
for prec in bf16x3 fp32; do
  for r in 1 2 3; do
    POPNET_TRAINX_STEM_HANDOVER=1 python scripts/train_bench.py 32 50 $prec 2>&1 | grep -v "host enqueue" | tail -1 | cut -c1-160 | sed "s/^/handover $prec: /"
    python scripts/train_bench.py 32 50 $prec 2>&1 | grep -v "host enqueue" | tail -1 | cut -c1-160 | sed "s/^/direct   $prec: /"
  done
done | tee $O/stem_ab.txt
