#!/bin/bash
# tools/gpu_pmc_cmd.sh "<counters>" <python script + args> — ONE PMC pass over an arbitrary python command
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_*
pass="$1"; shift
timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_one -o case -- python "$@" > $OUT/pmc_one.log 2>&1
tail -2 $OUT/pmc_one.log | cut -c1-300
python $R/tools/pmc_summary.py $OUT ${PMC_TAG:-r01_one} | grep -v "^void at"
