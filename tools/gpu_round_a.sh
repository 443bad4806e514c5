#!/bin/bash
# tools/gpu_round_a.sh — first GPU pass of a session: swscale parity (old + fast paths), variant sweep.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest sws" | tee $OUT/pytest_a.log
timeout 1500 python -m pytest tests/test_gpu_sws_fast.py tests/test_gpu_sws.py -m gpu -q --maxfail=12 2>&1 | tail -60 | tee -a $OUT/pytest_a.log
echo "== sweep" | tee $OUT/sweep.log
timeout 600 python tools/sweep_sws.py 2>&1 | tail -40 | tee -a $OUT/sweep.log
