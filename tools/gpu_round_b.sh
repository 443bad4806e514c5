#!/bin/bash
# tools/gpu_round_b.sh — parity of the h264 / me / tx kernels + shims, VALU issue-rate microbench, bench line,
# rocprofv3 kernel trace and PMC passes of the same bench command.  Everything lands under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest h264/me/tx/shims" | tee $OUT/pytest_b.log
timeout 1500 python -m pytest tests/test_gpu_h264.py tests/test_gpu_me.py tests/test_gpu_tx.py tests/test_gpu_shims.py -m gpu -q --maxfail=30 2>&1 | tail -80 | tee -a $OUT/pytest_b.log
echo "== ubench" | tee $OUT/ubench.log
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate valu_rate.hip 2>/dev/null && timeout 120 /tmp/valu_rate) 2>&1 | tee -a $OUT/ubench.log
echo "== bench" | tee $OUT/bench.log
timeout 900 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee -a $OUT/bench.log
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
tail -2 $OUT/prof.log
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  echo "== rocprofv3 pmc $pass"
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$tag.log 2>&1
  tail -1 $OUT/pmc_$tag.log
done
find $OUT -name "*.csv" | head -20
