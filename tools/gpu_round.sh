#!/bin/bash
# tools/gpu_round.sh — what one gpurun call executes on the MI355X box: parity tests, smoke, bench,
# and a rocprofv3 kernel trace of the same bench command.  Everything lands under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu" | tee $OUT/pytest.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee -a $OUT/pytest.log
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee -a $OUT/smoke.log
echo "== bench" | tee $OUT/bench.log
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -5 | tee -a $OUT/bench.log
if [ "${1:-}" != "noprof" ]; then
  echo "== rocprofv3 kernel trace"
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
  tail -3 $OUT/prof.log
  find $OUT/prof -name "*stats*" | head
fi
