#!/bin/bash
# kernel trace of the pipelined inference region -> three consecutive batches as a timeline (marker: group_readout_kernel, the last launch of a batch).  usage: timeline_infer.sh <tag> [bench args]
tag=$1; shift; O=gpurun_out/$tag; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o tl -- python3 $R/bench.py --steps 20 --warmup 5 --reps 1 --no-extras --no-cpu-baseline --no-power --no-h2d "$@" > $R/$O/prof.log 2>&1
cd $R; python scripts/r06/timeline.py $O/prof 30 group_readout 3 > $O/timeline_infer.txt; tail -12 $O/timeline_infer.txt; find $O -name "*kernel_trace.csv" -delete
