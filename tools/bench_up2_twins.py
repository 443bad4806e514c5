#!/usr/bin/env python3
"""tools/bench_up2_twins.py — the exact-2x kernel's twins (above 8 bits; with range conversion) under the measure build's FFHIP_UP2_VAR / FFHIP_UP2_DEPTH:
do the bench kernel's round-6 ingredients (non-temporal stores, six rows in flight, the bank in SGPRs) carry over?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffmpeg_amd import _lib
_lib.select("measure")
from ffmpeg_amd import swscale as S

dev = torch.device("cuda:0")
cases = (("p010 1080p->4K", 158, 158, 64, [("", ""), ("1", ""), ("1", "6"), ("0", "6")]), ("yuv420p10 1080p->4K", 62, 62, 64, [("", ""), ("1", ""), ("1", "6"), ("0", "6")]),
         ("yuvj420p->yuv420p 1080p->4K", 12, 0, 64, [("0", "3"), ("3", "3"), ("3", "6"), ("0", "6")]))
for key, sf, df, n, variants in cases:
    c = S.SwsContext(1920, 1080, sf, 3840, 2160, df, 4)
    s_ = [torch.randint(0, 256, (n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(sf, 1920, 1080)]
    if sf in (62, 158):
        for t_ in s_:
            t_.view(torch.int16).bitwise_and_(0x03FF if sf == 62 else -64)
    d_ = [torch.zeros((n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(df, 3840, 2160)]
    byt = n * (S.frame_bytes(sf, 1920, 1080) + S.frame_bytes(df, 3840, 2160))
    row, ref = {"case": key}, None
    for p in range(3):
        for var, depth in variants:
            for k, v in (("FFHIP_UP2_VAR", var), ("FFHIP_UP2_DEPTH", depth)):
                if v:
                    os.environ[k] = v
                else:
                    os.environ.pop(k, None)
            for d in d_:
                d.zero_()
            for _ in range(5):
                c.scale_batch(s_, d_)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30):
                c.scale_batch(s_, d_)
            b.record()
            torch.cuda.synchronize()
            if ref is None:
                ref = [d.clone() for d in d_]
            assert all(torch.equal(x, y) for x, y in zip(d_, ref)), (key, var, depth)
            row.setdefault("var%s_d%s" % (var or "-", depth or "-"), []).append(round(byt / (a.elapsed_time(b) / 30 * 1e-3) / 8e12, 4))
    print(json.dumps(row), flush=True)
    c.close()
