#!/bin/bash
# sws_scale_frame on frames of the hip AVHWDeviceType, through the reference's graph (oracle/_ref/hwcontext_hip_test graph ...): which
# conversions run as one pass in HBM, which the graph splits in two (intermediate frame in device memory), which it does not offer at all
T=oracle/_ref/hwcontext_hip_test
for c in "rgb24 bgr24 640 360" "rgba rgb24 641 359" "yuv444p rgb24 640 360" "rgb24 yuv444p 640 360" "yuv444p10le yuv444p 640 360" "yuv444p yuv444p16le 640 360" \
         "bgr0 rgb24 1920 1080" "yuv444p yuv444p 640 360 960 540" "rgb24 rgb24 640 360 320 180" "yuv444p yuv444p 640 360 1280 360" "yuv444p yuv444p 640 360 640 720" \
         "yuv420p rgb24 640 360" "nv12 yuv420p 640 360" "yuv420p yuv420p 640 360 1280 720"; do
  echo "== $c"; $T graph $c 2>&1 | tail -2
done
