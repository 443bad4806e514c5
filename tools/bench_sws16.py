#!/usr/bin/env python3
"""tools/bench_sws16.py — the scaler above 8 bits (k_sws_scale16): p010 / yuv420p10 1080p <-> 4K, frames resident in HBM."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import swscale as S  # noqa: E402

dev = torch.device("cuda", 0)
for key, sf, sw, sh, df, dw, dh, n in (("p010 1080p->4K", 158, 1920, 1080, 158, 3840, 2160, 16), ("yuv420p10 1080p->4K", 62, 1920, 1080, 62, 3840, 2160, 16),
                                       ("yuv420p10 4K->1080p", 62, 3840, 2160, 62, 1920, 1080, 16), ("p010 4K->nv12 1080p", 158, 3840, 2160, 23, 1920, 1080, 16),
                                       ("yuv420p 1080p->yuv420p10 4K", 0, 1920, 1080, 62, 3840, 2160, 16)):
    c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
    s_ = [torch.randint(0, 256, (n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(sf, sw, sh)]
    d_ = [torch.empty((n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(df, dw, dh)]
    for _ in range(2):
        c.scale_batch(s_, d_)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        c.scale_batch(s_, d_)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 5
    byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
    print(json.dumps({"case": key, "frames": n, "ms": round(t, 4), "Mpixels/s": round(n * dw * dh / (t * 1e-3) / 1e6, 1),
                      "GB/s": round(byt / (t * 1e-3) / 1e9, 1), "hbm_frac": round(byt / (t * 1e-3) / 1e9 / 8000, 4)}), flush=True)
    c.close()
