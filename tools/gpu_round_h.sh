#!/bin/bash
# tools/gpu_round_h.sh — scaler parity (all sws GPU tests), then headline bench + secondary-path bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest sws" | tee $OUT/pytest_h.log
timeout 1200 python -m pytest tests/test_gpu_sws_fast.py tests/test_gpu_sws.py tests/test_gpu_golden.py -m gpu -q --maxfail=10 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | tee -a $OUT/pytest_h.log
echo "== bench" | tee $OUT/bench_h.log
timeout 600 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -2 | tee -a $OUT/bench_h.log
echo "== bench_more" | tee $OUT/bench_more.log
timeout 600 python tools/bench_more.py 2>&1 | tail -40 | tee -a $OUT/bench_more.log
