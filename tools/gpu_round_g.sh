#!/bin/bash
# tools/gpu_round_g.sh — packed-RGB column walker: parity, then the secondary-path bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest rgb" | tee $OUT/pytest_g.log
timeout 900 python -m pytest tests/test_gpu_sws_fast.py tests/test_gpu_sws.py -m gpu -q --maxfail=10 -k "rgb" 2>&1 | tail -40 | tee -a $OUT/pytest_g.log
echo "== bench_more" | tee $OUT/bench_more.log
timeout 600 python tools/bench_more.py 2>&1 | tail -40 | tee -a $OUT/bench_more.log
