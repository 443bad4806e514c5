#!/usr/bin/env python3
"""tools/bench_qpel.py — A/B of the two luma qpel kernels: every 16x16 MB of 8 4K planes, put; mixed mcXY and per position."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
from ffmpeg_amd import h264  # noqa: E402

dev = torch.device("cuda", 0)
DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("mcxy", np.uint8), ("size_idx", np.uint8), ("avg", np.uint8),
               ("flags", np.uint8), ("src_x", np.int16), ("src_y", np.int16)])
W, H, P, planes = 3840, 2160, 32, int(os.environ.get("PLANES", "8"))
stride = W + 2 * P
rows = H + 2 * P
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
        dy, dx = rng.integers(-16, 17, my.size), rng.integers(-16, 17, my.size)
        b["dst_offset"] = y * stride + x
        b["src_offset"] = (y + dy) * stride + x + dx
        b["mcxy"] = rng.integers(0, 16, my.size) if mc < 0 else mc
        out.append(b)
    b = np.concatenate(out)
    return torch.from_numpy(b.view(np.uint8).reshape(len(b), 16)).to(dev), len(b)


QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"     # the default kernel, mixed positions and plain copies only (PMC runs)
ALL = len(sys.argv) > 1 and sys.argv[1] == "all"         # the default kernel, every position (PLANES=32 for the batch the verdict asks about)
for old, nb in ((("0", "4"),) if QUICK or ALL else (("0", "4"), ("0", "2"), ("0", "1"), ("1", "1"))):
    os.environ["FFHIP_QPEL_OLD"] = old
    os.environ["FFHIP_QPEL_NB"] = nb
    for mc in ((-1, 0) if QUICK else (-1,) + tuple(range(16)) if ALL else (-1, 0, 2, 8, 10, 5, 9)):
        d_bl, n = blocks(mc)
        for _ in range(2):
            h264.qpel_batch(dst, ref, stride, d_bl, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            h264.qpel_batch(dst, ref, stride, d_bl, n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        px = n * 256
        print(json.dumps({"kernel": "regs" if old == "1" else "lds x%s" % nb, "mcxy": "mixed" if mc < 0 else mc, "blocks": n, "ms": round(ms, 4),
                          "Gpixel/s": round(px / ms / 1e6, 1), "hbm_frac": round(2 * px / ms / 1e6 / 8000, 4)}), flush=True)
