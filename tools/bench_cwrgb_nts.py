#!/usr/bin/env python3
"""tools/bench_cwrgb_nts.py — the scaled packed-RGB column walker with and without non-temporal stores (measure build, FFHIP_CWRGB_NTS=0):
yuv420p 1080p -> rgb24 / bgra 4K bicubic, 32 frames, three alternating passes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffmpeg_amd import _lib
_lib.select("measure")
from ffmpeg_amd import swscale as S

dev = torch.device("cuda:0")
n = 32
for df, bpp in ((2, 3), (28, 4)):
    ctx = S.SwsContext(1920, 1080, 0, 3840, 2160, df, 4)
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(0, 1920, 1080)]
    dst = [torch.empty((n, 2160, bpp * 3840), dtype=torch.uint8, device=dev)]
    ref = None
    for p in range(3):
        for var in ("1", "0"):
            os.environ["FFHIP_CWRGB_NTS"] = var
            for _ in range(10):
                ctx.scale_batch(src, dst)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                ctx.scale_batch(src, dst)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 50
            cs = int(dst[0][:2].to(torch.int64).sum().item())
            ref = cs if ref is None else ref
            byt = n * (1920 * 1080 * 1.5 + 3840 * 2160 * bpp)
            print(json.dumps({"dst": df, "pass": p, "nts": var, "ms": round(ms, 4), "hbm_frac": round(byt / (ms * 1e-3) / 8e12, 4), "same_pixels": cs == ref}), flush=True)
    ctx.close()
