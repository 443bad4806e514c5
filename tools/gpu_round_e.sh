#!/bin/bash
# tools/gpu_round_e.sh — full GPU tier + sweep after the scalar-pointer / LDS-table MDCT changes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu" | tee $OUT/pytest_e.log
timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -30 | tee -a $OUT/pytest_e.log
echo "== sweep" | tee $OUT/sweep.log
timeout 600 python tools/sweep_sws.py 2>&1 | tail -40 | tee -a $OUT/sweep.log
echo "== bench" | tee $OUT/bench.log
timeout 900 python bench.py 2>&1 | tail -2 | tee -a $OUT/bench.log
