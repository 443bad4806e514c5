#!/bin/bash
# PMC passes of the bench command with the matrix-core scaler variant selected
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FFHIP_SWS_MFMA=1
rm -rf $OUT/pmc_*
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$tag.log 2>&1
  tail -1 $OUT/pmc_$tag.log | cut -c1-200
done
