#!/bin/bash
# tools/gpu.sh — the one runner for work on the MI355X box: what a `gpurun -- 'bash tools/gpu.sh ...'` call executes.
# Every argument is one step, run in order; the first word names the step, the rest are its arguments:
#   "test [pytest args]"          python -m pytest tests -m gpu -q <args>
#   "smoke"                       __graft_entry__.smoke()
#   "bench [bench.py args]"       python bench.py <args>             (the JSON line lands in gpurun_out/bench.json)
#   "py <script> [args]"          python <script> <args>
#   "sh <command> [args]"         any command (e.g. a tools/ubench binary)
#   "prof <tag> <script> [args]"  rocprofv3 --kernel-trace --stats of `python <script> <args>`; the kernel stats CSV is
#                                 copied to gpurun_out/<tag>_kernel_stats.csv
#   "pmc <tag> <script> [args]"   the three PMC passes (FETCH_SIZE / WRITE_SIZE / SQ set), each its own rocprofv3 run with
#                                 --kernel-trace only, folded by tools/pmc_summary.py into gpurun_out/<tag>_pmc.{txt,json}
#   "pmc1 <tag> <counters,comma separated> <script> [args]"   one PMC pass with the given counters
# Logs: gpurun_out/<n>_<step>.log; gpurun merges gpurun_out/ back into the repo, the summaries worth keeping are then
# copied to profiles/ by hand.  STEP_TIMEOUT (seconds, default 900) bounds every step.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
T=${STEP_TIMEOUT:-900}
mkdir -p $OUT
export TMPDIR=/tmp
n=0
for step in "$@"; do
  n=$((n+1))
  set -- $step
  kind=$1; shift
  log=$OUT/${n}_${kind}.log
  echo "== [$n] $step"
  case $kind in
    test)  (cd $R && timeout $T python -m pytest tests -m gpu -q -x "$@" 2>&1 | tail -25) | tee $log ;;
    smoke) (cd $R && timeout $T python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5) | tee $log ;;
    bench) (cd $R && timeout $T python bench.py "$@" 2>&1 | tail -3) | tee $log; grep '^{' $log | tail -1 > $OUT/bench.json ;;
    sh)    (cd $R && timeout $T "$@" 2>&1 | tail -${PY_TAIL:-60}) | tee $log ;;
    py)    (cd $R && timeout $T python "$@" 2>&1 | tail -${PY_TAIL:-60}) | tee $log ;;
    prof)  tag=$1; shift
           (cd /tmp && rm -rf $OUT/prof_$tag && timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o run -- python $R/"$@" > $log 2>&1)
           tail -2 $log | cut -c1-300
           f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1)
           [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv && head -8 $f | cut -c1-220
           rm -rf $OUT/prof_$tag ;;
    pmc)   tag=$1; shift
           rm -rf $OUT/pmc_*; i=0
           for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU"; do
             i=$((i+1))
             (cd /tmp && timeout $T rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$i -o case -- python $R/"$@" > $OUT/pmc_$i.log 2>&1)
             tail -1 $OUT/pmc_$i.log | cut -c1-200
           done
           python $R/tools/pmc_summary.py $OUT $tag | grep -v "^void at" | tail -40 | tee $log
           rm -rf $OUT/pmc_* ;;
    pmc1)  tag=$1; ctr=$(echo $2 | tr ',' ' '); shift; shift
           rm -rf $OUT/pmc_*
           (cd /tmp && timeout $T rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_1 -o case -- python $R/"$@" > $OUT/pmc_1.log 2>&1)
           tail -1 $OUT/pmc_1.log | cut -c1-200
           python $R/tools/pmc_summary.py $OUT $tag | grep -v "^void at" | tail -40 | tee $log
           rm -rf $OUT/pmc_* ;;
    *) echo "unknown step: $kind" ;;
  esac
done
