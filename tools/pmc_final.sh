#!/bin/bash
# SQ counters of the fused forward kernel (two passes), rows of cheb_fused_kernel only
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_sq_$i -- python3 tools/run_forward.py c3 bf16x3 fused 2 > /dev/null 2>&1
  head -1 /tmp/pmc_sq_$i/*/*_counter_collection.csv > $O/pmc_sq_$i.csv
  grep cheb_fused_kernel /tmp/pmc_sq_$i/*/*_counter_collection.csv | cut -c1-400 >> $O/pmc_sq_$i.csv
done
