#!/bin/bash
# needs a diagnostic library: make -C deepsphere-cosmo-tf2_amd/csrc clean && make -C deepsphere-cosmo-tf2_amd/csrc -j8 ABLATE=1
# instruction counts of the fused kernel per ablation build (outputs are wrong when a bit is set)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in ${1:-0 1 2 3}; do
  export DSPH_FUSED_DEBUG=$d
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_abl_$d -- python3 tools/run_forward.py c3 bf16x3 fused 1 > /dev/null 2>&1
  head -1 /tmp/pmc_abl_$d/*/*_counter_collection.csv > gpurun_out/pmc_abl_$d.csv
  grep cheb_fused_kernel /tmp/pmc_abl_$d/*/*_counter_collection.csv >> gpurun_out/pmc_abl_$d.csv
done
