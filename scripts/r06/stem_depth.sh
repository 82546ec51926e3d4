#!/bin/bash
# duration of the stem weight-gradient launch by prefetch depth (kernel trace of the two-stream step).  usage: stem_depth.sh <tag>
for d in 1 2 4; do
  export POPNET_TRAINX_STEM_DEPTH=$d
  bash scripts/r06/timeline.sh $1_d$d > /dev/null 2>&1
  grep -h "tstem_wgrad" gpurun_out/$1_d$d/timeline_bf16x3.txt | head -1 | cut -c1-100; head -1 gpurun_out/$1_d$d/timeline_bf16x3.txt | cut -c1-90
done
