#!/bin/bash
# tools/gpu_pmc_multi.sh <tag> <python script + args> — the three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ set), each its own
# rocprofv3 run with --kernel-trace only, folded into gpurun_out/<tag>_pmc.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_*
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$i -o case -- python $R/"$@" > $OUT/pmc_$i.log 2>&1
  tail -1 $OUT/pmc_$i.log | cut -c1-200
done
python $R/tools/pmc_summary.py $OUT $tag | grep -v "^void at" | tail -40
rm -rf $OUT/pmc_*
