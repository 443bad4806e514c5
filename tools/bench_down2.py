#!/usr/bin/env python3
"""tools/bench_down2.py — exact 2:1 down-scaling, 4K -> 1080p bicubic, 64 resident frames: the static-schedule kernel
(sws_down2.hip, strip heights, unit order) beside the wide walker it replaces (FFHIP_SWS_DOWN2=0); HIP events."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import swscale as S  # noqa: E402

dev = torch.device("cuda", 0)
n = 64
QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"   # the product configuration only (PMC runs)
for fmt, name in (((23, "nv12"),) if QUICK else ((23, "nv12"), (0, "yuv420p"))):
    s_ = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(fmt, 3840, 2160)]
    d_ = [torch.empty((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(fmt, 1920, 1080)]
    byt = n * (S.frame_bytes(fmt, 3840, 2160) + S.frame_bytes(fmt, 1920, 1080))
    envs = [{"FFHIP_SWS_DOWN2": "0"}, {}]
    envs += [{"FFHIP_DN2_STRIP": str(st), "FFHIP_DN2_XCD": x} for st in (16, 20, 28, 36, 44, 60, 120) for x in ("0", "1")]
    for env in ([{}] if QUICK else envs):
        for k in ("FFHIP_SWS_DOWN2", "FFHIP_DN2_STRIP", "FFHIP_DN2_XCD"):
            os.environ.pop(k, None)
        os.environ.update(env)
        c = S.SwsContext(3840, 2160, fmt, 1920, 1080, fmt, 4)
        for _ in range(3):
            c.scale_batch(s_, d_)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c.scale_batch(s_, d_)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(json.dumps({"case": "%s 4K -> 1080p bicubic x%d" % (name, n), "env": env or "default", "down2": c.down2_path, "ms": round(ms, 4),
                          "Mpixels/s": round(n * 1920 * 1080 / ms / 1e3, 1), "GB/s": round(byt / ms / 1e6, 1),
                          "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
        c.close()
