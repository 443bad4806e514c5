"""Conversions at the source's size between depths / layouts: which kernel serves them and at what fraction of HBM.  python tools/bench_same_size.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffmpeg_amd import swscale as S  # noqa: E402

dev = torch.device("cuda:0")
P010, YUV420P10, NV12, YUV420P, BGRA, RGB24 = 158, 62, 23, 0, 28, 2
for name, sf, df in (("p010 -> nv12", P010, NV12), ("yuv420p10 -> yuv420p", YUV420P10, YUV420P), ("nv12 -> p010", NV12, P010), ("yuv420p -> yuv420p10", YUV420P, YUV420P10),
                     ("p010 -> yuv420p10", P010, YUV420P10), ("yuv420p10 -> p010", YUV420P10, P010), ("p010 -> bgra", P010, BGRA), ("yuv420p10 -> rgb24", YUV420P10, RGB24)):
    n, w, h = 32, 1920, 1080
    try:
        c = S.SwsContext(w, h, sf, w, h, df, 4)
    except Exception as e:
        print(json.dumps({"case": name, "refused": str(e)[:100]}), flush=True)
        continue
    src = [torch.randint(0, 256, (n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(sf, w, h)]
    if sf in (P010, YUV420P10):
        for t_ in src:
            t_.view(torch.int16).bitwise_and_(0x03FF if sf == YUV420P10 else -64)
    dst = [torch.zeros((n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(df, w, h)]
    byt = n * (S.frame_bytes(sf, w, h) + S.frame_bytes(df, w, h))
    try:
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
    except Exception as e:
        print(json.dumps({"case": name, "error": str(e)[:120]}), flush=True)
    c.close()
