#!/bin/bash
# stem forward launch duration by blocks per CU (one-stream kernel trace).  usage: stemf_blocks.sh <tag>
export POPNET_TRAINX_STREAMS=1
for b in 1 2 3; do
  export POPNET_STEM_FWD_BLOCKS=$b
  bash scripts/r06/timeline.sh $1_b$b > /dev/null 2>&1
  echo "blocks/CU $b: $(grep -h tstem_fwd gpurun_out/$1_b$b/timeline_bf16x3.txt | head -1 | cut -c1-70)"
done
