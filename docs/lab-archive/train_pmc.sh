# PMC passes over the training step (B from $2, default 32): MFMA busy, TA busy, LDS conflicts, wait buckets per kernel
out=gpurun_out/$1
B=${2:-32}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_TA_BUSY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pass$i -o runc -- python3 scripts/train_bench.py $B 1 ${3:-fp32} nograph > $out/pass$i.log 2>&1
done
python3 scripts/pmc_kernel_table.py $out | tee $out/table.txt
