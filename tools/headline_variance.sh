#!/bin/bash
# tools/headline_variance.sh — the headline line of bench.py several times on ONE box: cold / settled, short / long, bare / under
# rocprofv3 --kernel-trace, to tell box state (clocks, placement) from the tool.  Output: gpurun_out/r05_headline_variance.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r05_headline_variance.txt; : > $O
B="python bench.py --no-extras --no-cpu-baseline --no-pmc"
run() { echo "== $1" >> $O; shift; "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   ms_per_step %.4f  kernel_ms %.4f  frac %.4f  steps %d warmup %d settle %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['steps'], d['warmup'], d['config'].get('settle_ms')))
" >> $O; }
run "cold box: 20/5, no settle"      $B --steps 20 --warmup 5 --settle-ms 0
run "20/5, no settle (2nd process)"  $B --steps 20 --warmup 5 --settle-ms 0
run "20/5, settle 300 ms"            $B --steps 20 --warmup 5 --settle-ms 300
run "20/5, settle 1000 ms"           $B --steps 20 --warmup 5 --settle-ms 1000
run "200/50, no settle"              $B --steps 200 --warmup 50 --settle-ms 0
run "200/50, settle 300"             $B --steps 200 --warmup 50
run "2000/50 (1.7 s of launches)"    $B --steps 2000 --warmup 50
export TMPDIR=/tmp
run "200/50 under rocprofv3 --kernel-trace" rocprofv3 --kernel-trace --stats -d /tmp/hv_prof -- $B --steps 200 --warmup 50
run "200/50 bare again"              $B --steps 200 --warmup 50
run "20/5 default settle, as the driver calls it" $B --steps 20 --warmup 5
cat $O
