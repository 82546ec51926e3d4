#!/bin/bash
# round-6 evidence: bench lines (driver command, 200-step regions), rocprofv3 kernel stats of the bench command and of the eager one-stream runs,
# the planes training step (kernel stats on two streams, PMC on one), the stall breakdown of the inference kernels.
# usage: profiles.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$tag; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py > $O/bench_200.json 2> $O/bench_200.err
E="--no-cpu-baseline --no-extras --no-h2d --no-power --reps 1 --no-graph --pipeline 1 --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o runc -- python3 bench.py --no-cpu-baseline --no-extras --no-power > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_eager1 -o runc -- python3 bench.py $E > $O/stats_eager1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_x3_eager1 -o runc -- python3 bench.py $E --precision bf16x3 > $O/stats_x3_eager1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_bf16x3 -o runc -- python3 scripts/train_bench.py 32 16 bf16x3 nograph > $O/stats_train_bf16x3.log 2>&1
POPNET_TRAINX_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_bf16x3_one_stream -o runc -- python3 scripts/train_bench.py 32 16 bf16x3 nograph > $O/stats_train_bf16x3_one_stream.log 2>&1
for p in bf16x3 bf16x3-nchw fp32; do python scripts/train_bench.py 32 50 $p 2>/dev/null | tail -2 | head -1; done > $O/train_bench_lines.txt
for d in stats stats_eager1 stats_x3_eager1 stats_train_bf16x3 stats_train_bf16x3_one_stream; do f=$(ls $O/$d/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$d.csv; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
bash scripts/r06/pmc_stall.sh $tag train > $O/pmc_train.log 2>&1
bash scripts/r06/pmc_stall.sh $tag infer > $O/pmc_infer.log 2>&1
ls $O; cat $O/train_bench_lines.txt
