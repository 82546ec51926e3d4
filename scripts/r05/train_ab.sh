#!/bin/bash
# training-step A/B on one box: train_ab.sh "<ENV=VAL ...>" ...   (prints ms/step of scripts/train_bench.py, both precisions)
cd "$GRAFT_REPO_ROOT" || exit 1
for envs in "$@"; do
  echo "=== [$envs]"
  for p in fp32 bf16x3; do env $envs timeout 600 python3 scripts/train_bench.py 32 10 $p 2>&1 | grep "ms/step"; done
done
