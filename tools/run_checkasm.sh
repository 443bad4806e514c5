#!/bin/bash
# tools/run_checkasm.sh [test ...] — runs the reference's checkasm (oracle/_ref/checkasm_hip, built by oracle/refbuild `make checkasm`)
# for the `hip` cpu flag, one test at a time, and prints the harness's own per-test verdict.  On the GPU box:
#   gpurun -- 'bash tools/run_checkasm.sh > gpurun_out/checkasm.log'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
EXE=$R/oracle/_ref/checkasm_hip
[ -x "$EXE" ] || { echo "no $EXE (build it where /root/reference exists: make -C oracle/refbuild checkasm)"; exit 2; }
T=${@:-h264dsp h264qpel h264chroma h264pred motion hevc_add_res hevc_idct hevc_deblock hevc_dequant hevc_pel hevc_sao hevc_pred vp9dsp float_dsp av_tx sw_scale sw_ops sw_yuv2rgb sw_yuv2yuv sw_rgb videodsp}
rc=0
cd /tmp
for t in $T; do
  echo "== checkasm --test=$t"
  timeout ${CK_TIMEOUT:-900} "$EXE" --test=$t ${CK_SEED:-1} 2>&1 | tail -n ${CK_TAIL:-12}
  [ ${PIPESTATUS[0]} -eq 0 ] || rc=1
done
exit $rc
