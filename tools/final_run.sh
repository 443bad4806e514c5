#!/bin/bash
# tools/final_run.sh — the round's closing pass on the GPU box: the whole -m gpu suite (log kept), smoke, bench.py (JSON line kept), the
# reference's checkasm for the hip flag in one pass, and the rocprofv3 kernel stats of the bench command.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
STEP_TIMEOUT=${SUITE_TIMEOUT:-1500} PY_TAIL=30 bash tools/gpu_pytest.sh gpu_suite tests -q
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_final.json
CK_TAIL=6 timeout 1200 bash tools/run_checkasm.sh > gpurun_out/checkasm.log 2>&1; echo "checkasm rc=$?"
STEP_TIMEOUT=600 bash tools/gpu.sh "prof final_bench bench.py --no-pmc --steps 20 --warmup 5" | tail -12
# the anchor alone: only the 256-frame launches of the headline kernel (no extras, no CPU leg, no PMC children), so that
# algorithmic bytes / AverageNs of the one k_sws_up2 row == roofline.achieved of the same run; tools/headline_check.py says by how much
STEP_TIMEOUT=600 bash tools/gpu.sh "prof headline bench.py --no-extras --no-cpu-baseline --no-pmc --steps 20 --warmup 5" | tail -6
cp gpurun_out/1_prof.log gpurun_out/headline_bench.log 2>/dev/null
python tools/headline_check.py gpurun_out/headline_kernel_stats.csv gpurun_out/headline_bench.log | tee gpurun_out/headline_check.txt
# the headline kernel's instruction mix and traffic (three PMC passes over six 256-frame launches)
STEP_TIMEOUT=600 bash tools/gpu.sh "pmc up2 bench.py --pmc-child" | tail -16
# MBAFF frames in the picture layer: a 1920 x 1088 stream through the whole decoder, the two chains timed and traced
STEP_TIMEOUT=600 bash tools/gpu.sh "py tools/bench_h264_mbaff.py 4" "prof h264_mbaff tools/bench_h264_mbaff.py 4" | tail -8
cp gpurun_out/1_py.log gpurun_out/h264_mbaff_bench.log 2>/dev/null
