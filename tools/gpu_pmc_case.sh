#!/bin/bash
# tools/gpu_pmc_case.sh <args of tools/run_case.py> — PMC passes over one scaler configuration
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_*
python $R/tools/run_case.py "$@" | tail -1
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "FETCH_SIZE WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -o case -- python $R/tools/run_case.py "$@" > $OUT/pmc_$tag.log 2>&1
  tail -1 $OUT/pmc_$tag.log | cut -c1-200
done
python $R/tools/pmc_summary.py $OUT ${PMC_TAG:-r01_case} | grep -v "^void at"
