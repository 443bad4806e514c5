#!/bin/bash
# round i: HEVC MC (tuned kernel + weighted / bi modes): parity both kernels, then the microbench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hevc.py tests/test_gpu_golden.py -m gpu -x -q --timeout 300 > gpurun_out/i_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/i_tests.log
tail -5 gpurun_out/i_tests.log
timeout 300 python tools/bench_hevc.py > gpurun_out/i_bench_hevc.log 2>&1
tail -12 gpurun_out/i_bench_hevc.log
