"""tools/bench_rgbin.py — the round-6 libswscale sources on resident frames: bgra / rgb24 1080p -> nv12 / yuv420p at the source's size and
scaled, p010 -> bgra; per-kernel times come from `tools/gpu.sh "prof rgbin tools/bench_rgbin.py"`."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffmpeg_amd import swscale as S

n = 32
for key, sf, sw, sh, df, dw, dh in (("bgra_1080p_nv12_1080p", 28, 1920, 1080, 23, 1920, 1080), ("rgb24_1080p_yuv420p_1080p", 2, 1920, 1080, 0, 1920, 1080),
                                    ("bgra_1080p_nv12_720p", 28, 1920, 1080, 23, 1280, 720), ("p010_1080p_bgra_1080p", 158, 1920, 1080, 28, 1920, 1080)):
    c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
    s_ = S.alloc_batch(sf, sw, sh, n, "cuda:0")
    d_ = S.alloc_batch(df, dw, dh, n, "cuda:0")
    for t_ in s_:
        t_.random_(0, 256)
    if sf == 158:
        for t_ in s_:
            t_.view(torch.int16).bitwise_and_(-64)
    for _ in range(3):
        c.scale_batch(s_, d_)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        c.scale_batch(s_, d_)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 10
    byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
    print(json.dumps({"case": key, "frames": n, "ms": round(t, 4), "Gpixel/s": round(n * dw * dh / t / 1e6, 1), "hbm_frac": round(byt / t / 1e6 / 8000, 4)}), flush=True)
    c.close()
