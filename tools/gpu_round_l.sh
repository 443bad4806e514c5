#!/bin/bash
# round l: full GPU tier (pytest -m gpu, smoke, bench with extras) + kernel-trace summaries of the h264pred and DCT kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu" | tee $OUT/pytest_l.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 2>&1 | tail -15 | tee -a $OUT/pytest_l.log
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT/smoke.log
echo "== bench" | tee $OUT/bench.log
timeout 900 python bench.py 2>&1 | tail -2 | tee -a $OUT/bench.log
timeout 300 python tools/bench_pfa.py > $OUT/bench_pfa.log 2>&1
tail -3 $OUT/bench_pfa.log
timeout 200 python tools/bench_h264_pred.py > $OUT/bench_h264_pred.log 2>&1
tail -4 $OUT/bench_h264_pred.log
timeout 200 python tools/bench_aac.py > $OUT/bench_aac.log 2>&1
tail -3 $OUT/bench_aac.log
cd /tmp && export TMPDIR=/tmp
for c in "1024 0" "1024 1"; do
    set -- $c
    rm -rf /tmp/prof_dct
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dct -o dct -- python $R/tools/run_tx.py $1 $2 9 > /dev/null 2>&1
    f=$(find /tmp/prof_dct -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp $f $OUT/dct$1_$2_kernel_stats.csv
done
rm -rf /tmp/prof_pred
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pred -o pred -- python $R/tools/bench_h264_pred.py > /dev/null 2>&1
f=$(find /tmp/prof_pred -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/h264_pred_kernel_stats.csv
ls $OUT | tail -20
