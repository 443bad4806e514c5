#!/usr/bin/env python3
"""tools/bench_rgb24_variants.py — yuv420p -> rgb24 4K, 64 frames: the product kernel and the measured variants of the measure build
(FFHIP_YUV2RGB_VARIANT: st = plain stores (the kernel up to round 4), xcd = XCD-contiguous workgroup numbering, ntl = non-temporal loads too),
three alternating passes, 20 warm-up + 100 timed launches each."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffmpeg_amd import _lib
_lib.select("measure")
from ffmpeg_amd import swscale as S

dev = torch.device("cuda:0")
n, w, h = 64, 3840, 2160
ctx = S.SwsContext(w, h, 0, w, h, 2, 4)
src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(0, w, h)]
dst = [torch.empty((n, h, 3 * w), dtype=torch.uint8, device=dev)]
ref = None
for p in range(3):
    for var in ("", "st", "xcd", "ntl", "st+xcd"):
        if var:
            os.environ["FFHIP_YUV2RGB_VARIANT"] = var
        else:
            os.environ.pop("FFHIP_YUV2RGB_VARIANT", None)
        for _ in range(20):
            ctx.scale_batch(src, dst)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            ctx.scale_batch(src, dst)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 100
        cs = int(dst[0][:2].to(torch.int64).sum().item())
        if ref is None:
            ref = cs
        print(json.dumps({"pass": p, "variant": var or "product", "ms": round(ms, 4), "hbm_frac": round(n * w * h * 4.5 / (ms * 1e-3) / 8e12, 4),
                          "same_pixels": cs == ref}))
