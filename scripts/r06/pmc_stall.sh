#!/bin/bash
# Round 6: what the sub-0.25 inference kernels wait for, and the MFMA utilisation of the planes training step.
# Each counter set is its own rocprofv3 run with --kernel-trace only (the pool refuses --pmc next to other trace domains); the program sits directly after `--`.
# usage: pmc_stall.sh <tag> infer|train
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; what=$2; O=gpurun_out/$tag; mkdir -p $O
if [ "$what" = train ]; then CMD="python3 scripts/train_bench.py 32 4 bf16x3 nograph"; export POPNET_TRAINX_STREAMS=1
else CMD="python3 bench.py --steps 10 --warmup 3 --reps 1 --no-h2d --no-extras --no-cpu-baseline --no-power --no-graph --pipeline 1 ${3:-}"; fi
mkdir -p $O/pmc_$what; i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$what/pass$i -o runc -- $CMD > $O/pmc_$what/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 scripts/r06/pmc_stall_table.py $O/pmc_$what | tee $O/stall_breakdown_$what.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
