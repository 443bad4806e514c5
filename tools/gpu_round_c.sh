#!/bin/bash
# tools/gpu_round_c.sh — VALU issue-rate microbench, swscale parity across variants, variant sweep.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== ubench" | tee $OUT/ubench.log
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate valu_rate.hip 2>&1 | grep -i error; timeout 120 /tmp/valu_rate) 2>&1 | tee -a $OUT/ubench.log
echo "== pytest sws fast + shims" | tee $OUT/pytest_c.log
timeout 1500 python -m pytest tests/test_gpu_sws_fast.py tests/test_gpu_shims.py -m gpu -q --maxfail=12 2>&1 | tail -40 | tee -a $OUT/pytest_c.log
echo "== sweep" | tee $OUT/sweep.log
timeout 600 python tools/sweep_sws.py 2>&1 | tail -40 | tee -a $OUT/sweep.log
