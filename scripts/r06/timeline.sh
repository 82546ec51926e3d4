#!/bin/bash
# kernel trace of the two-stream training step -> one step's timeline.  usage: timeline.sh <tag> [precision]
tag=$1; prec=${2:-bf16x3}; O=gpurun_out/$tag; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o tl -- python3 $R/scripts/train_bench.py 32 16 $prec nograph > $R/$O/prof.log 2>&1
cd $R; python scripts/r06/timeline.py $O/prof 12 > $O/timeline_$prec.txt; tail -8 $O/timeline_$prec.txt; find $O -name "*kernel_trace.csv" -delete
