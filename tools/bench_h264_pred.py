#!/usr/bin/env python3
"""tools/bench_h264_pred.py — H264PredContext batch kinds over 4K planes (one GPU, HIP events).  Every second block of a
checkerboard per launch (a block's neighbours are final), modes mixed; the figure is the kernel's rate, not a decoder's — a real
picture's wavefront hands over far fewer blocks per launch."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import h264  # noqa: E402

dev = torch.device("cuda", 0)
W, H, planes = 3840, 2160, 4
rng = np.random.default_rng(6)
for kind, n, nmodes, name in ((0, 4, 12, "pred4x4"), (1, 8, 12, "pred8x8l"), (2, 8, 11, "pred8x8"), (3, 16, 7, "pred16x16")):
    hh = (H // n) * n
    by, bx = np.meshgrid(np.arange(n, planes * hh, n), np.arange(n, W - n, n), indexing="ij")
    keep = (((by // n) + (bx // n)) & 1) == 0
    by, bx = by[keep], bx[keep]
    nb = by.size
    rec = np.zeros(nb, h264.PRED_DTYPE)
    rec["offset"] = by * W + bx
    rec["mode"] = rng.integers(0, nmodes, nb)
    rec["flags"] = 3 if kind == 1 else 0
    if kind == 0:
        rec["aux"] = rec["offset"] - W + 4
    d_rec = torch.from_numpy(rec.view(np.uint8).reshape(nb, 12)).to(dev)
    pic = torch.randint(0, 256, (planes * hh, W), dtype=torch.uint8, device=dev)
    h264.pred_batch(kind, pic, W, d_rec, nb)
    ms, reps = 0.0, 10
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h264.pred_batch(kind, pic, W, d_rec, nb)
        e1.record()
        torch.cuda.synchronize()
        ms += e0.elapsed_time(e1) / reps
    px = nb * n * n
    byt = px + nb * (12 + 3 * n + 1)                # block written; record + edge line read
    print(json.dumps({"case": "h264 %s, modes mixed, checkerboard of %d 4K planes" % (name, planes), "blocks": nb, "ms": round(ms, 4),
                      "Gpixel/s": round(px / ms / 1e6, 1), "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
