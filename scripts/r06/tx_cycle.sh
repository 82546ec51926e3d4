#!/bin/bash
# one dev cycle of the planes training engine on the GPU box: correctness vs the oracle at two sizes, timing, per-kernel profile.  usage: tx_cycle.sh <tag>
tag=$1; O=gpurun_out/$tag; mkdir -p $O
timeout 300 python scripts/r06/trainx_check.py 2 48 64 > $O/check_small.log 2>&1; grep -n "engine terms\|whole" $O/check_small.log
timeout 300 python scripts/r06/trainx_check.py 2 224 224 > $O/check_224.log 2>&1; grep -n "engine terms\|whole" $O/check_224.log
python scripts/train_bench.py 32 50 bf16x3 > $O/tb_x3.log 2>&1; tail -2 $O/tb_x3.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o tx -- python3 $R/scripts/train_bench.py 32 16 bf16x3 nograph > $R/$O/prof.log 2>&1
cd $R; python scripts/r06/dbstats.py $O/prof 34 > $O/kstats.txt; head -${2:-32} $O/kstats.txt; find $O -name "*kernel_trace.csv" -delete
