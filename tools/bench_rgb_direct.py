"""RGB into planar 4:2:2 / 4:4:4 at the source's size: the one-pass form (every bank the identity) against the two-stage form
(FFHIP_SWS_RGB_DIRECT_OFF=1, the measure build).  python tools/bench_rgb_direct.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffmpeg_amd import _lib, swscale as S
_lib.select("measure")
dev = torch.device("cuda:0")
for name, sf, df in (("bgra->yuv444p", 28, 5), ("rgb24->yuv422p", 2, 4)):
    n, w, h = 32, 1920, 1080
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(sf, w, h)]
    for off in (None, "1"):
        os.environ.pop("FFHIP_SWS_RGB_DIRECT_OFF", None)
        if off: os.environ["FFHIP_SWS_RGB_DIRECT_OFF"] = off
        c = S.SwsContext(w, h, sf, w, h, df, 4)
        dst = [torch.zeros((n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(df, w, h)]
        for _ in range(3): c.scale_batch(src, dst)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): c.scale_batch(src, dst)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        byt = n * (S.frame_bytes(sf, w, h) + S.frame_bytes(df, w, h))
        print(json.dumps({"case": name, "form": "two-stage" if off else "direct", "ms": round(ms, 4), "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
        c.close()
