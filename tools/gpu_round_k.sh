#!/bin/bash
# round k: full GPU tier (pytest -m gpu, smoke, bench with extras) — no profiler passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu" | tee $OUT/pytest_k.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 2>&1 | tail -15 | tee -a $OUT/pytest_k.log
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT/smoke.log
echo "== bench" | tee $OUT/bench.log
timeout 900 python bench.py 2>&1 | tail -2 | tee -a $OUT/bench.log
timeout 300 python tools/bench_pfa.py > $OUT/bench_pfa.log 2>&1
tail -4 $OUT/bench_pfa.log
