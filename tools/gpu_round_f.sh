#!/bin/bash
# tools/gpu_round_f.sh — matrix-core scaler variant: parity, then the variant sweep
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest mfma" | tee $OUT/pytest_f.log
timeout 900 python -m pytest tests/test_gpu_sws_fast.py -m gpu -q --maxfail=10 -k "mfma" 2>&1 | tail -40 | tee -a $OUT/pytest_f.log
echo "== sweep" | tee $OUT/sweep.log
timeout 600 python tools/sweep_sws.py 2>&1 | tail -40 | tee -a $OUT/sweep.log
