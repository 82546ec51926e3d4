# separate rocprofv3 --pmc passes over scripts/fwd_only.py (bf16, B=32); results under gpurun_out/$1
out=gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o runc -- python3 scripts/fwd_only.py 3 > $out/$tag.log 2>&1
done
python3 scripts/pmc_table.py $out conv3_kernel conv_mfma
