#!/usr/bin/env python3
"""tools/bench_walk16.py — the 16-bit column walker (k_sws_walk16) with its source rows staged through LDS (round 6, the product) against the
per-lane global loads it had (measure build, FFHIP_W16_STAGE=0): the bench's in-between ratios above 8 bits, alternating passes, whole
outputs compared."""
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
cases = (("p010 720p->1080p", 158, 1280, 720, 158, 1920, 1080, 64), ("p010 4K->1440p", 158, 3840, 2160, 158, 2560, 1440, 16),
         ("yuv420p10 1080p->1440p", 62, 1920, 1080, 62, 2560, 1440, 32), ("yuv420p10 720p->p010 1080p", 62, 1280, 720, 158, 1920, 1080, 32),
         ("p010 1080p->nv12 720p", 158, 1920, 1080, 23, 1280, 720, 32), ("yuv420p10 4K->yuv420p10 1080p+1 (3841 wide)", 62, 3841, 2161, 62, 1921, 1081, 8))
for key, sf, sw, sh, df, dw, dh, n in cases:
    c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
    s_ = [torch.randint(0, 256, (n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(sf, sw, sh)]
    if sf in (62, 158):
        for t_ in s_:
            t_.view(torch.int16).bitwise_and_(0x03FF if sf == 62 else -64)
    d_ = [torch.zeros((n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(df, dw, dh)]
    ref, row = None, {"case": key, "frames": n, "walk16": c.walk16_path}
    byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
    for p in range(2):
        for label, val in (("lds", None), ("global", "0")):
            if val:
                os.environ["FFHIP_W16_STAGE"] = val
            else:
                os.environ.pop("FFHIP_W16_STAGE", None)
            for d in d_:
                d.zero_()
            for _ in range(3):
                c.scale_batch(s_, d_)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                c.scale_batch(s_, d_)
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b) / 10
            if ref is None:
                ref = [d.clone() for d in d_]
            row.setdefault(label, []).append(round(byt / (t * 1e-3) / 8e12, 4))
            row["same_" + label] = all(torch.equal(x, y) for x, y in zip(d_, ref))
    os.environ.pop("FFHIP_W16_STAGE", None)
    print(json.dumps(row), flush=True)
    c.close()
