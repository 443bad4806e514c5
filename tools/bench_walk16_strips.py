#!/usr/bin/env python3
"""tools/bench_walk16_strips.py — k_sws_walk16 (LDS-staged rows) at strip heights 64 / 32 / 16 (measure build, FFHIP_W16_STRIP) for the bench's
up-scaling ratios above 8 bits."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib  # noqa: E402
_lib.select("measure")
from ffmpeg_amd import swscale as S  # noqa: E402

dev = torch.device("cuda", 0)
cases = (("p010 720p->1080p", 158, 1280, 720, 158, 1920, 1080, 64), ("yuv420p10 1080p->1440p", 62, 1920, 1080, 62, 2560, 1440, 32),
         ("p010 4K->1440p", 158, 3840, 2160, 158, 2560, 1440, 16))
for key, sf, sw, sh, df, dw, dh, n in cases:
    c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
    s_ = [torch.randint(0, 256, (n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(sf, sw, sh)]
    for t_ in s_:
        t_.view(torch.int16).bitwise_and_(0x03FF if sf == 62 else -64)
    d_ = [torch.zeros((n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(df, dw, dh)]
    byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
    row = {"case": key}
    for p in range(2):
        for strip in ("", "64", "32", "16"):
            if strip:
                os.environ["FFHIP_W16_STRIP"] = strip
            else:
                os.environ.pop("FFHIP_W16_STRIP", None)
            for _ in range(3):
                c.scale_batch(s_, d_)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                c.scale_batch(s_, d_)
            b.record()
            torch.cuda.synchronize()
            row.setdefault(strip or "product", []).append(round(byt / (a.elapsed_time(b) / 10 * 1e-3) / 8e12, 4))
    os.environ.pop("FFHIP_W16_STRIP", None)
    print(json.dumps(row), flush=True)
    c.close()
