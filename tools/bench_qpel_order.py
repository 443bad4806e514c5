#!/usr/bin/env python3
"""tools/bench_qpel_order.py — does the ORDER of a mixed-mcXY batch matter?  The same blocks (every 16x16 block of PLANES 4K planes, motion
+-24, random mcXY) in raster order, sorted by mcXY over the whole batch, sorted inside every macroblock row, and sorted inside every group
of 16 consecutive blocks (what one workgroup takes); uniform batches (one position for all) beside them."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
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
out = []
for p in range(planes):
    b = np.zeros(my.size, DT)
    y = p * rows + P + my.reshape(-1) * 16
    x = P + mx.reshape(-1) * 16
    dy, dx = rng.integers(-24, 25, my.size), rng.integers(-24, 25, my.size)
    b["dst_offset"] = y * stride + x
    b["src_offset"] = (y + dy) * stride + x + dx
    b["mcxy"] = rng.integers(0, 16, my.size)
    out.append(b)
base = np.concatenate(out)
n = len(base)


def grouped_sort(b, g):
    idx = np.arange(len(b))
    key = (idx // g).astype(np.int64) * 16 + b["mcxy"]
    return b[np.argsort(key, kind="stable")]


orders = {"raster": base, "sorted_batch": base[np.argsort(base["mcxy"], kind="stable")], "sorted_mb_row": grouped_sort(base, W // 16),
          "sorted_64": grouped_sort(base, 64), "sorted_16": grouped_sort(base, 16)}
for mc in (0, 2, 8, 10, 5):
    u = base.copy()
    u["mcxy"] = mc
    orders["uniform_mc%d" % mc] = u
for p in range(2):
    for name, b in orders.items():
        d_bl = torch.from_numpy(b.view(np.uint8).reshape(len(b), 16)).to(dev)
        for _ in range(3):
            h264.qpel_batch(dst, ref, stride, d_bl, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            h264.qpel_batch(dst, ref, stride, d_bl, n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(json.dumps({"pass": p, "order": name, "ms": round(ms, 4), "hbm_frac": round(2 * n * 256 / ms / 1e6 / 8000, 4)}), flush=True)
