#!/usr/bin/env python3
"""tools/bench_up2_strips.py — the headline launch (nv12 1080p -> 4K bicubic, 256 frames) at strips of 24 .. 120 source rows
(measure build, FFHIP_UP2_STRIP), and with the ragged column blocks shared by 1 / 2 / 4 frames (FFHIP_UP2_FSHIFT): two alternating passes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffmpeg_amd import _lib
_lib.select("measure")
from ffmpeg_amd import swscale as S

dev = torch.device("cuda:0")
n = 256
ctx = S.SwsContext(1920, 1080, 23, 3840, 2160, 23, 4)
src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 1920, 1080)]
dst = [torch.empty((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 3840, 2160)]
for _ in range(300):   # settle the clocks
    ctx.scale_batch(src, dst)
torch.cuda.synchronize()
cfgs = [("product", {})] + [("strip %s" % s, {"FFHIP_UP2_STRIP": s}) for s in ("24", "36", "48", "60", "90", "120")] + \
       [("fshift %s" % s, {"FFHIP_UP2_FSHIFT": s}) for s in ("0", "1", "2")]
for p in range(2):
    for name, env in cfgs:
        for k in ("FFHIP_UP2_STRIP", "FFHIP_UP2_FSHIFT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(20):
            ctx.scale_batch(src, dst)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            ctx.scale_batch(src, dst)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 100
        print(json.dumps({"pass": p, "config": name, "ms": round(ms, 4), "hbm_frac": round(n * 15552000 / (ms * 1e-3) / 8e12, 4)}), flush=True)
