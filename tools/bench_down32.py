"""Exact 3:2 down-scaling (sws_down32.hip): nv12 / yuv420p 1080p -> 720p and 4K -> 1440p.  python tools/bench_down32.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffmpeg_amd import swscale as S  # noqa: E402

dev = torch.device("cuda:0")
for name, sf, sw, sh, df, dw, dh, n in (("nv12 1080p->720p", 23, 1920, 1080, 23, 1280, 720, 64), ("yuv420p 1080p->720p", 0, 1920, 1080, 0, 1280, 720, 64),
                                        ("nv12 4K->1440p", 23, 3840, 2160, 23, 2560, 1440, 16), ("p010 4K->1440p", 158, 3840, 2160, 158, 2560, 1440, 16),
                                        ("yuv420p10 1080p->720p", 62, 1920, 1080, 62, 1280, 720, 64), ("p010 1440p->1080p", 158, 2560, 1440, 158, 1920, 1080, 32), ("p010 1080p->nv12 720p", 158, 1920, 1080, 23, 1280, 720, 64),
                                        ("yuv420p10 1440p->1080p", 62, 2560, 1440, 62, 1920, 1080, 32),
                                        # exact 2:1 (k_sws_down2) beside them
                                        ("nv12 4K->1080p", 23, 3840, 2160, 23, 1920, 1080, 64), ("p010 4K->1080p", 158, 3840, 2160, 158, 1920, 1080, 16),
                                        ("p010 4K->nv12 1080p", 158, 3840, 2160, 23, 1920, 1080, 16), ("nv12 4K->rgb24 1080p", 23, 3840, 2160, 2, 1920, 1080, 32)):
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(sf, sw, sh)]
    if sf in (158, 62):
        for t_ in src:
            t_.view(torch.int16).bitwise_and_(0x03FF if sf == 62 else -64)
    dst = [torch.zeros((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(df, dw, dh)]
    byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
    c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
    for _ in range(3):
        c.scale_batch(src, dst)
    for p in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            c.scale_batch(src, dst)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(json.dumps({"case": name, "frames": n, "pass": p, "paths": c.paths, "ms": round(ms, 4), "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
    c.close()
