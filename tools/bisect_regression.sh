#!/bin/bash
# tools/bisect_regression.sh — VERDICT r04 weak #9: HEAD's library, HEAD built without -mllvm -amdgpu-mfma-vgpr-form, round 3's library
# (590d041), three rounds, alternating, one box.  The variant libraries are built where /root/reference is not needed (any container):
#   bisect_libs/libffhip_noflag.so, bisect_libs/libffhip_r03.so  (git-ignored; they travel with the gpurun snapshot)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
out=gpurun_out/r05_regression_bisect.txt
mkdir -p gpurun_out
: > $out
for round in 1 2 3; do
  for lib in ffmpeg_amd/libffhip.so bisect_libs/libffhip_noflag.so bisect_libs/libffhip_r03.so; do
    [ -f $lib ] || continue
    timeout 600 python tools/bisect_bench.py $lib 40 2>&1 | grep '^{' | sed "s/^/round $round /" >> $out
  done
done
cat $out
