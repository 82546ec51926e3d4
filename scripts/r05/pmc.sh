#!/bin/bash
# PMC table (MFMA busy, TA busy, LDS conflicts) of one eager, un-pipelined bench run: pmc.sh <tag> [bench args, e.g. --precision bf16x3]
# each counter set in its own rocprofv3 run with --kernel-trace only (the pool refuses --pmc next to other trace domains)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift; O=gpurun_out/$tag; mkdir -p $O/pmc
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_TA_BUSY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/pass$i -o runc -- python3 bench.py --steps 10 --warmup 3 --reps 1 --no-h2d --no-extras --no-cpu-baseline --no-graph --pipeline 1 "$@" > $O/pmc/pass$i.log 2>&1
done
python3 scripts/pmc_kernel_table.py $O/pmc | tee $O/pmc_table.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
