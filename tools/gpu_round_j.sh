#!/bin/bash
# round j: 15xM prime-factor MDCT: parity (tx + golden), then timing of the Opus / AAC sizes
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tx.py tests/test_gpu_golden.py -m gpu -x -q --timeout 300 -k "pfa or golden" > gpurun_out/j_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/j_tests.log
tail -15 gpurun_out/j_tests.log
timeout 300 python tools/bench_pfa.py > gpurun_out/j_bench_pfa.log 2>&1
tail -12 gpurun_out/j_bench_pfa.log
