#!/usr/bin/env python3
"""tools/bench_host_face.py — the host-pointer (SwsFunc) face, PCIe included: ffhip_sws_scale on the same pageable host frames call after call;
ms per frame over 64 calls after 6 warm-up calls, two passes, outputs compared.  (profiles/r06_host_face.txt also holds the run that compared
this with buffers registered after their second sighting — "pinned" there; that variant was not kept.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffmpeg_amd import _lib  # noqa: E402
_lib.select("measure")
from ffmpeg_amd import swscale as S  # noqa: E402

cases = (("nv12 1080p -> nv12 4K bicubic", 23, 1920, 1080, 23, 3840, 2160), ("nv12 1080p -> rgb24 1080p", 23, 1920, 1080, 2, 1920, 1080),
         ("yuv420p 4K -> rgb24 4K (table converter)", 0, 3840, 2160, 2, 3840, 2160), ("nv12 4K -> nv12 1080p bicubic", 23, 3840, 2160, 23, 1920, 1080))
for key, sf, sw, sh, df, dw, dh in cases:
    row = {"case": key}
    ref = None
    for label in ("first", "again"):
        c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
        rng = np.random.default_rng(5)
        hs = [rng.integers(0, 256, (r, cc), dtype=np.uint8) for r, cc in S.plane_shapes(sf, sw, sh)]
        hd = [np.zeros((r, cc), np.uint8) for r, cc in S.plane_shapes(df, dw, dh)]
        for _ in range(6):
            c.scale(hs, hd)
        t0 = time.perf_counter()
        for _ in range(64):
            c.scale(hs, hd)
        t = (time.perf_counter() - t0) / 64
        if ref is None:
            ref = [d.copy() for d in hd]
        byt = sum(a.nbytes for a in hs) + sum(a.nbytes for a in hd)
        row[label] = {"ms_per_frame": round(t * 1e3, 4), "PCIe_GB/s": round(byt / t / 1e9, 1), "same": all(np.array_equal(a, b) for a, b in zip(hd, ref))}
        c.close()
    print(json.dumps(row), flush=True)
