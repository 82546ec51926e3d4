#!/bin/bash
# round-5 evidence: rocprofv3 kernel stats of the bench command (pipelined), of the eager one-stream run (bf16, bf16x3, yolo), of the training step
# (fp32, bf16x3), HBM traffic and TA / MFMA / LDS PMC passes (each counter set in its own run, with --kernel-trace only).
# usage: profiles.sh <tag> <commit>
tag=$1; export POPNET_COMMIT=$2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$tag; mkdir -p $O
E="--no-cpu-baseline --no-extras --no-h2d --reps 1 --no-graph --pipeline 1 --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o runc -- python3 bench.py --no-cpu-baseline --no-extras > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_eager1 -o runc -- python3 bench.py $E > $O/stats_eager1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_x3_eager1 -o runc -- python3 bench.py $E --precision bf16x3 > $O/stats_x3_eager1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_yolo_eager1 -o runc -- python3 bench.py $E --net yolo > $O/stats_yolo_eager1.log 2>&1
for p in fp32 bf16x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_$p -o runc -- python3 scripts/train_bench.py 32 6 $p > $O/stats_train_$p.log 2>&1
done
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$set -o runc -- python3 scripts/fwd_only.py 3 > $O/pmc_$set.log 2>&1
done
python3 scripts/make_traffic_json.py $O $O/conv_hbm_traffic.json | tail -30
for mode in bf16 bf16x3; do
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_TA_BUSY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    mkdir -p $O/pmc_$mode; rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$mode/pass$i -o runc -- python3 bench.py --steps 20 --warmup 4 --reps 1 --no-h2d --no-extras --no-cpu-baseline --no-graph --pipeline 1 --precision $mode > $O/pmc_$mode/pass$i.log 2>&1
  done
  python3 scripts/pmc_kernel_table.py $O/pmc_$mode | tee $O/pmc_table_$mode.txt
done
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_TA_BUSY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  mkdir -p $O/pmc_train; rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_train/pass$i -o runc -- python3 scripts/train_bench.py 32 4 bf16x3 nograph > $O/pmc_train/pass$i.log 2>&1
done
python3 scripts/pmc_kernel_table.py $O/pmc_train | tee $O/pmc_table_train_bf16x3.txt
for d in stats stats_eager1 stats_x3_eager1 stats_yolo_eager1 stats_train_fp32 stats_train_bf16x3; do f=$(ls $O/$d/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$d.csv; done
# keep the merge small: drop the raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -la $O
