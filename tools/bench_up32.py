"""Exact 3:2 up-scaling above 8 bits (sws_up32.hip) against the 16-bit walker (FFHIP_SWS_UP32=0), alternating in one process.
python tools/bench_up32.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffmpeg_amd import _lib, swscale as S  # noqa: E402

_lib.select("measure")
dev = torch.device("cuda:0")
P010, YUV420P10, NV12, YUV420P = 158, 62, 23, 0
CASES = [("p010 720p->1080p", P010, 1280, 720, P010, 1920, 1080, 64), ("yuv420p10 720p->1080p", YUV420P10, 1280, 720, YUV420P10, 1920, 1080, 64),
         ("p010 1440p->4K", P010, 2560, 1440, P010, 3840, 2160, 16), ("yuv420p10 1080p->1440p", YUV420P10, 1920, 1080, YUV420P10, 2560, 1440, 32),
         ("p010 1080p->1440p", P010, 1920, 1080, P010, 2560, 1440, 32), ("nv12 720p->p010 1080p", NV12, 1280, 720, P010, 1920, 1080, 64)]
for name, sf, sw, sh, df, dw, dh, n in CASES:
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(sf, sw, sh)]
    if sf in (P010, YUV420P10):
        for t_ in src:
            t_.view(torch.int16).bitwise_and_(0x03FF if sf == YUV420P10 else -64)
    byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
    sums = {}
    for p in range(2):
        for kern, env in (("k_sws_up32", None), ("k_sws_walk16", "0")):
            os.environ.pop("FFHIP_SWS_UP32", None)
            if env:
                os.environ["FFHIP_SWS_UP32"] = env
            c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
            dst = [torch.zeros((n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(df, dw, dh)]
            for _ in range(3):
                c.scale_batch(src, dst)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                c.scale_batch(src, dst)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 10
            sums[kern] = [int(d.to(torch.int64).sum().item()) for d in dst]
            print(json.dumps({"case": name, "frames": n, "pass": p, "kernel": kern, "ms": round(ms, 4), "hbm_frac": round(byt / ms / 1e6 / 8000, 4),
                              "same_pixels": sums[kern] == sums.get("k_sws_up32")}), flush=True)
            c.close()
