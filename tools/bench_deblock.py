#!/usr/bin/env python3
"""tools/bench_deblock.py — frame-order luma deblocking of 4K planes: per-frame time vs frames per launch.  DB_DEPTH=10 (9 / 12 / 14):
the same at that depth (uint16 samples) through ffhip_h264_deblock_frames_dev_hbd."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import h264  # noqa: E402

dev = torch.device("cuda", 0)
w, h = 3840, 2160
mbw, mbh = w // 16, h // 16
rng = np.random.default_rng(3)
ed = np.zeros(mbw * mbh * 8, dtype=np.dtype([("o", np.int32), ("k", np.uint8), ("a", np.uint8), ("b", np.uint8), ("p", np.uint8),
                                                 ("tc", np.int8, 4)]))
ed["a"], ed["b"] = 40, 9
# bS = 4 (the strong filter) exists on macroblock EDGES of intra macroblocks only (edge 0 of either direction,
# h264_loopfilter.c check_mv / filter_mb_dir); DB_INTRA = share of macroblocks that are intra (default 0.25), DB_INTRA_ANY=1 puts
# the strong filter on a quarter of ALL edges instead (what the tests do: a stream cannot, the kernel must cope anyway)
intra = float(os.environ.get("DB_INTRA", "0.25"))
if os.environ.get("DB_INTRA_ANY") == "1":
    ed["k"] = np.where(rng.random(ed.size) < .25, 4, 0)
else:
    mb_intra = rng.random(mbw * mbh) < intra
    k = np.zeros((mbw * mbh, 2, 4), np.uint8)
    k[mb_intra, :, 0] = 4
    ed["k"] = k.ravel()
ed["tc"] = rng.integers(0, 4, (ed.size, 4))
ded = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev)
depth = int(os.environ.get("DB_DEPTH", "8"))
if depth > 8:
    for nf in (1, 8, 32):
        batch = (torch.randint(100, 140, (nf, h, w), dtype=torch.int32, device=dev) << (depth - 8)).to(torch.int16)
        dd = ded.repeat(nf, 1)
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            h264.deblock_frames_hbd(depth, batch, 2 * w * h, nf, 2 * w, mbw, mbh, dd)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(json.dumps({"depth": depth, "frames_per_launch": nf, "ms": round(ms, 3), "ms_per_frame": round(ms / nf, 4), "Gpixel/s": round(nf * w * h / ms / 1e6, 2)}), flush=True)
    sys.exit(0)
for nf in (1, 2, 4, 8, 16, 32, 64):
    batch = torch.randint(100, 140, (nf, h, w), dtype=torch.uint8, device=dev)
    dd = ded.repeat(nf, 1)
    h264.deblock_frames(batch, w * h, nf, w, mbw, mbh, dd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    h264.deblock_frames(batch, w * h, nf, w, mbw, mbh, dd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(json.dumps({"frames_per_launch": nf, "ms": round(ms, 3), "ms_per_frame": round(ms / nf, 4),
                      "Gpixel/s": round(nf * w * h / ms / 1e6, 2)}), flush=True)
# one 4:2:0 chroma plane per frame (1920x1080 samples, 8x8 per MB): decoder order, same wavefront
cw, ch = w // 2, h // 2
edc = np.zeros(mbw * mbh * 4, dtype=ed.dtype)
edc["a"], edc["b"] = 40, 9
kc = np.full((mbw * mbh, 2, 2), 2, np.uint8)
kc[rng.random(mbw * mbh) < intra, :, 0] = 6
edc["k"] = kc.ravel()
edc["tc"] = rng.integers(0, 4, (edc.size, 4))
dedc = torch.from_numpy(edc.view(np.uint8).reshape(-1, 12)).to(dev)
for nf in (1, 8, 32, 64):
    batch = torch.randint(100, 140, (nf, ch, cw), dtype=torch.uint8, device=dev)
    dd = dedc.repeat(nf, 1)
    h264.deblock_frames_chroma(batch, cw * ch, nf, cw, mbw, mbh, dd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    h264.deblock_frames_chroma(batch, cw * ch, nf, cw, mbw, mbh, dd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(json.dumps({"chroma_plane_frames_per_launch": nf, "ms": round(ms, 3), "ms_per_plane": round(ms / nf, 4),
                      "Gpixel/s": round(nf * cw * ch / ms / 1e6, 2)}), flush=True)
