#!/usr/bin/env python3
"""tools/bench_up2rgb.py — yuv420p 1080p -> rgb24 / bgra 4K bicubic, 32 frames: the column walker with RGB output (FFHIP_SWS_UP2RGB=0)
against the exact-2x kernel with the writer fused (sws_up2rgb.hip) and its measured variants (v1 plain stores, v2 direct 8-byte stores
without the LDS transposer), strips of 30 .. 540 luma rows.  Measure build; alternating passes on one box."""
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
configs = [("walker", {"FFHIP_SWS_UP2RGB": "0"})] + \
          [("up2rgb steps %s" % s, {"FFHIP_UP2RGB_STEPS": s}) for s in ("24", "36", "48")] + \
          [("up2rgb plain stores", {"FFHIP_SWS_UP2RGB": "v1"}), ("up2rgb direct stores", {"FFHIP_SWS_UP2RGB": "v2"}),
           ("up2rgb transposer 8-byte pieces", {"FFHIP_SWS_UP2RGB": "v3"}), ("up2rgb transposer read back at once", {"FFHIP_SWS_UP2RGB": "v4"}),
           ("up2rgb direct, steps 30", {"FFHIP_SWS_UP2RGB": "v2", "FFHIP_UP2RGB_STEPS": "30"}),
           ("up2rgb default", {}), ("up2rgb one frame per pack", {"FFHIP_UP2RGB_FPP": "1"}),
           ("up2rgb one frame per pack, steps 24", {"FFHIP_UP2RGB_FPP": "1", "FFHIP_UP2RGB_STEPS": "24"}),
           ("up2rgb two frames per pack, steps 24", {"FFHIP_UP2RGB_FPP": "2", "FFHIP_UP2RGB_STEPS": "24"})]
pad = int(os.environ.get("UP2RGB_PAD_ROWS", "0"))   # extra rows between the frames of the destination batch (another frame pitch)
for df, bpp in ((2, 3), (28, 4)):
    ctx = S.SwsContext(1920, 1080, 0, 3840, 2160, df, 4)
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(0, 1920, 1080)]
    dst = [torch.empty((n, 2160 + pad, bpp * 3840), dtype=torch.uint8, device=dev)[:, :2160]]
    ref = None
    for p in range(2):
        for name, env in configs:
            for k in ("FFHIP_SWS_UP2RGB", "FFHIP_UP2RGB_STEPS", "FFHIP_UP2RGB_FPP"):
                os.environ.pop(k, None)
            os.environ.update(env)
            for _ in range(40):
                ctx.scale_batch(src, dst)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(200):
                ctx.scale_batch(src, dst)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 200
            cs = int(dst[0][:2].to(torch.int64).sum().item())
            ref = cs if ref is None else ref
            byt = n * (1920 * 1080 * 1.5 + 3840 * 2160 * bpp)
            print(json.dumps({"dst": df, "pass": p, "kernel": name, "ms": round(ms, 4), "hbm_frac": round(byt / (ms * 1e-3) / 8e12, 4),
                              "same_pixels": cs == ref}), flush=True)
    ctx.close()
