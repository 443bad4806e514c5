#!/usr/bin/env python3
"""tools/bench_hevc.py — the hevcdsp batch faces over 4K luma planes (one GPU, HIP events): deblocking (both directions),
SAO (band / edge CTBs), uni-directional quarter-sample MC."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import hevc  # noqa: E402

dev = torch.device("cuda", 0)
W, H, planes = 3840, 2160, 8
rng = np.random.default_rng(2)


def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


pic = torch.randint(100, 140, (planes * H, W), dtype=torch.uint8, device=dev)
# ---- deblocking: every 8x8 grid edge of the planes, vertical edges then horizontal ones ----
segs = []
for vertical in (1, 0):
    ys, xs = (np.arange(0, planes * H, 8), np.arange(8, W, 8)) if vertical else (np.arange(8, planes * H, 8), np.arange(0, W, 8))
    if not vertical:
        ys = ys[ys % H != 0]
    yy, xx = np.meshgrid(ys, xs, indexing="ij")
    ed = np.zeros(yy.size, hevc.EDGE_DTYPE)
    ed["offset"] = (yy * W + xx).reshape(-1)
    ed["kind"] = vertical
    ed["beta"] = 38
    ed["tc"] = 6
    segs.append(torch.from_numpy(ed.view(np.uint8).reshape(-1, 16)).to(dev))
ms = timed(lambda: [hevc.loop_filter_batch(pic, W, s, s.shape[0]) for s in segs])
print(json.dumps({"case": "hevc luma deblocking, %d 4K planes, all 8x8-grid edges (2 launches)" % planes, "segments": int(sum(s.shape[0] for s in segs)),
                  "ms": round(ms, 4), "Gpixel/s": round(planes * W * H / ms / 1e6, 1)}), flush=True)
# ---- SAO: 64x64 CTBs, band and edge alternating, source = a second copy ----
src = torch.randint(0, 256, (planes * H + 2, W + 2), dtype=torch.uint8, device=dev)
blocks = []
for by in range(0, planes * H, 64):
    for bx in range(0, W, 64):
        blocks.append((by * W + bx, (by + 1) * (W + 2) + bx + 1, (0, 2, -1, 1, -2), (by // 64 + bx // 64) & 1, (bx // 64) & 3, 64,
                       min(64, planes * H - by), (0, 0)))
rec = np.array(blocks, hevc.SAO_DTYPE)
d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1, 24)).to(dev)
ms = timed(lambda: hevc.sao_batch(pic, W, src, W + 2, d_rec, len(rec)))
print(json.dumps({"case": "hevc SAO, %d 4K planes of 64x64 CTBs (band/edge alternating)" % planes, "blocks": len(rec), "ms": round(ms, 4),
                  "Gpixel/s": round(planes * W * H / ms / 1e6, 1), "hbm_frac": round(2 * planes * W * H / ms / 1e6 / 8000, 4)}), flush=True)
# ---- uni-directional luma MC: every 16x16 block, random quarter-sample positions and small displacements ----
P = 16
ref = torch.randint(0, 256, (planes * H + 2 * P, W + 2 * P), dtype=torch.uint8, device=dev)
by, bx = np.meshgrid(np.arange(0, planes * H, 16), np.arange(0, W, 16), indexing="ij")
n = by.size
mc = np.zeros(n, hevc.MC_DTYPE)
mc["dst_offset"] = (by * W + bx).reshape(-1)
mc["src_offset"] = ((by + P + rng.integers(-8, 9, by.shape)) * (W + 2 * P) + bx + P + rng.integers(-8, 9, by.shape)).reshape(-1)
mc["width"] = mc["height"] = 16
mc["mx"], mc["my"] = rng.integers(0, 4, n), rng.integers(0, 4, n)
d_mc = torch.from_numpy(mc.view(np.uint8).reshape(-1, 12)).to(dev)
ms = timed(lambda: hevc.mc_batch(0, 1, pic, W, ref, W + 2 * P, d_mc, n))
print(json.dumps({"case": "hevc put_hevc_qpel_uni, every 16x16 block of %d 4K planes, mixed (mx, my)" % planes, "blocks": n, "ms": round(ms, 4),
                  "Gpixel/s": round(planes * W * H / ms / 1e6, 1), "hbm_frac": round(2 * planes * W * H / ms / 1e6 / 8000, 4)}), flush=True)
