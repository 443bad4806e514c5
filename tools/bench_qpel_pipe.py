#!/usr/bin/env python3
"""tools/bench_qpel_pipe.py — k_h264_qpel_m against its software-pipelined twin (FFHIP_QPEL_PIPE=1, measure build): every 16x16 block of
PLANES 4K planes, motion +-24, mixed mcXY / copies / the centre position; three alternating passes; the outputs must be identical."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")
from ffmpeg_amd import h264  # noqa: E402

dev = torch.device("cuda", 0)
DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("mcxy", np.uint8), ("size_idx", np.uint8), ("avg", np.uint8),
               ("flags", np.uint8), ("src_x", np.int16), ("src_y", np.int16)])
W, H, P, planes = 3840, 2160, 32, int(os.environ.get("PLANES", "32"))
stride, rows = W + 2 * P, H + 2 * P
rng = np.random.default_rng(5)
ref = torch.randint(0, 256, (planes * rows, stride), dtype=torch.uint8, device=dev)
dst = torch.zeros_like(ref)
my, mx = np.meshgrid(np.arange(H // 16), np.arange(W // 16), indexing="ij")


def blocks(mc):
    out = []
    for p in range(planes):
        b = np.zeros(my.size, DT)
        y = p * rows + P + my.reshape(-1) * 16
        x = P + mx.reshape(-1) * 16
        dy, dx = rng.integers(-24, 25, my.size), rng.integers(-24, 25, my.size)
        b["dst_offset"] = y * stride + x
        b["src_offset"] = (y + dy) * stride + x + dx
        b["mcxy"] = rng.integers(0, 16, my.size) if mc < 0 else mc
        out.append(b)
    b = np.concatenate(out)
    return torch.from_numpy(b.view(np.uint8).reshape(len(b), 16)).to(dev), len(b)


cases = {mc: blocks(mc) for mc in (-1, 0, 10)}
sums = {}
for p in range(3):
    for var in ("0", "1"):
        os.environ["FFHIP_QPEL_PIPE"] = var
        for mc, (d_bl, n) in cases.items():
            for _ in range(3):
                h264.qpel_batch(dst, ref, stride, d_bl, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                h264.qpel_batch(dst, ref, stride, d_bl, n)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            cs = int(dst[::7].to(torch.int64).sum().item())
            same = sums.setdefault(mc, cs) == cs
            print(json.dumps({"pass": p, "pipe": var, "mcxy": "mixed" if mc < 0 else mc, "ms": round(ms, 4),
                              "hbm_frac": round(2 * n * 256 / ms / 1e6 / 8000, 4), "same_pixels": same}), flush=True)
