"""8-bit scaling at the common ratios that are not 2: which kernel serves them (ffhip_sws_fast_path bits) and at what fraction of HBM.
python tools/bench_ratios8.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffmpeg_amd import swscale as S  # noqa: E402

dev = torch.device("cuda:0")
for name, sf, sw, sh, df, dw, dh, n in (("nv12 720p->1080p", 23, 1280, 720, 23, 1920, 1080, 64), ("yuv420p 720p->1080p", 0, 1280, 720, 0, 1920, 1080, 64),
                                        ("nv12 1080p->1440p", 23, 1920, 1080, 23, 2560, 1440, 32), ("nv12 1440p->4K", 23, 2560, 1440, 23, 3840, 2160, 16),
                                        ("nv12 1080p->720p", 23, 1920, 1080, 23, 1280, 720, 64), ("nv12 4K->1440p", 23, 3840, 2160, 23, 2560, 1440, 16),
                                        ("nv12 1440p->1080p", 23, 2560, 1440, 23, 1920, 1080, 32), ("nv12 1080p->540p", 23, 1920, 1080, 23, 960, 540, 64)):
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(sf, sw, sh)]
    dst = [torch.zeros((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(df, dw, dh)]
    byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
    c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
    for _ in range(3):
        c.scale_batch(src, dst)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        c.scale_batch(src, dst)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(json.dumps({"case": name, "frames": n, "paths": c.paths, "ms": round(ms, 4), "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
    c.close()
