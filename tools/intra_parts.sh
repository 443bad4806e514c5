#!/bin/bash
# tools/intra_parts.sh — the intra reconstruction wavefront alone (no deblocking) on pictures that isolate its parts:
# one row (no hand-off at all), four rows (one workgroup: LDS hand-offs only), 8 rows (one memory boundary), the 4K picture.
for mb in 240x1 240x4 240x8 240x16 240x135; do
  echo "== $mb"
  BENCH_MB=$mb BENCH_NODEBLOCK=1 BENCH_DEPTH=${BENCH_DEPTH:-8} timeout 300 python /root/repo/tools/bench_h264_picture.py --intra 2>&1 | tail -2 | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l)
        print(d['intra_macroblocks'], 'intra MBs', d['ms_per_picture_gpu'], 'ms')
    except Exception as e:
        print(l.strip()[:200])
"
done
