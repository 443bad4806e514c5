#!/bin/bash
# tools/gpu_round_d.sh — the full GPU tier the driver runs (pytest -m gpu, smoke, bench) + rocprofv3 kernel trace and
# PMC passes of the same bench command.  Everything lands under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu" | tee $OUT/pytest_d.log
timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -30 | tee -a $OUT/pytest_d.log
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT/smoke.log
echo "== bench" | tee $OUT/bench.log
timeout 900 python bench.py 2>&1 | tail -2 | tee -a $OUT/bench.log
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
tail -2 $OUT/prof.log
rm -rf $OUT/pmc_*
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  echo "== rocprofv3 pmc $pass"
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$tag.log 2>&1
  tail -1 $OUT/pmc_$tag.log
done
