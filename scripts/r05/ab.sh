#!/bin/bash
# A/B on one box: ab.sh <tag> <mode: x3|quick|yolo> "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]   (an empty string = the default build)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; mode=$2; shift 2
i=0
for envs in "$@"; do
  i=$((i+1)); echo "=== variant $i: [$envs]"
  env $envs bash scripts/r05/check.sh ${tag}_v$i $mode 2>&1 | grep -v "amdgpu.ids\|^quick\|^bench" | head -${AB_LINES:-9}
done
