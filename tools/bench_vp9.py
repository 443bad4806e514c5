#!/usr/bin/env python3
"""tools/bench_vp9.py — VP9 itxfm_add batches over 4K luma planes (one GPU, HIP events): every block of a size, types mixed."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import vp9  # noqa: E402

dev = torch.device("cuda", 0)
W, H, planes = 3840, 2160, 4
rng = np.random.default_rng(5)
for tx in (0, 1, 2, 3):
    n = 4 << tx
    hh = (H // n) * n
    by, bx = np.meshgrid(np.arange(0, planes * hh, n), np.arange(0, W, n), indexing="ij")
    nb = by.size
    rec = np.zeros(nb, vp9.TU_DTYPE)
    rec["coeff_offset"] = np.arange(nb) * n * n
    rec["dst_offset"] = (by * W + bx).reshape(-1)
    rec["txtp"] = rng.integers(0, 4, nb)
    d_rec = torch.from_numpy(rec.view(np.uint8).reshape(nb, 12)).to(dev)
    pic = torch.randint(0, 256, (planes * hh, W), dtype=torch.uint8, device=dev)
    src = torch.randint(-300, 301, (nb, n * n), dtype=torch.int16, device=dev)
    co = src.clone()
    vp9.itxfm_add_batch(tx, co, pic, W, d_rec, nb)
    ms = 0.0
    reps = 5
    for _ in range(reps):
        co.copy_(src)                              # the transform consumes its coefficients
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        vp9.itxfm_add_batch(tx, co, pic, W, d_rec, nb)
        e1.record()
        torch.cuda.synchronize()
        ms += e0.elapsed_time(e1) / reps
    px = nb * n * n
    byt = px * 6                                   # coefficients read + cleared (2 + 2 B), picture read + written (1 + 1 B)
    print(json.dumps({"case": "vp9 itxfm_add %dx%d, mixed types, %d 4K planes" % (n, n, planes), "blocks": nb, "ms": round(ms, 4),
                      "Gpixel/s": round(px / ms / 1e6, 1), "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
# motion compensation: every 16x16 block of the planes, regular / sharp / smooth 8-tap and bilinear mixed, all (mx, my), put
P = 16
ref = torch.randint(0, 256, (planes * H + 2 * P, W + 2 * P), dtype=torch.uint8, device=dev)
pic = torch.zeros((planes * H, W), dtype=torch.uint8, device=dev)
by, bx = np.meshgrid(np.arange(0, planes * H, 16), np.arange(0, W, 16), indexing="ij")
n = by.size
for name, filt in (("8-tap (3 sets mixed)", lambda k: rng.integers(0, 3, k)), ("bilinear", lambda k: np.full(k, 3))):
    mc = np.zeros(n, vp9.MC_DTYPE)
    mc["dst_offset"] = (by * W + bx).reshape(-1)
    mc["src_offset"] = ((by + P + rng.integers(-8, 9, by.shape)) * (W + 2 * P) + bx + P + rng.integers(-8, 9, by.shape)).reshape(-1)
    mc["width"] = mc["height"] = 16
    mc["filter"] = filt(n)
    mc["mx"], mc["my"] = rng.integers(0, 16, n), rng.integers(0, 16, n)
    d_mc = torch.from_numpy(mc.view(np.uint8).reshape(-1, 16)).to(dev)
    vp9.mc_batch(pic, W, ref, W + 2 * P, d_mc, n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        vp9.mc_batch(pic, W, ref, W + 2 * P, d_mc, n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"case": "vp9 mc put %s, every 16x16 block of %d 4K planes, mixed (mx, my)" % (name, planes), "blocks": n, "ms": round(ms, 4),
                      "Gpixel/s": round(planes * W * H / ms / 1e6, 1), "hbm_frac": round(2 * planes * W * H / ms / 1e6 / 8000, 4)}), flush=True)

def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# ---- loop filters, function level: every 8x8-grid edge of the planes as 8-sample records, 8 wide; column edges, then row edges ----
EDGE_DT = np.dtype([("offset", np.int32), ("wd_idx", np.uint8), ("dir", np.uint8), ("E", np.uint8), ("I", np.uint8), ("H", np.uint8), ("pad", np.uint8, 3)])
pic = torch.from_numpy(np.clip(np.cumsum(np.random.default_rng(3).integers(-2, 3, (planes * H, W)), axis=1) + 128, 0, 255).astype(np.uint8)).to(dev)
for d, name in ((0, "column"), (1, "row")):
    ys, xs = (np.arange(0, planes * H, 8), np.arange(8, W, 8)) if d == 0 else (np.arange(8, planes * H, 8), np.arange(0, W, 8))
    if d:
        ys = ys[ys % H != 0]
    yy, xx = np.meshgrid(ys, xs, indexing="ij")
    ed = np.zeros(yy.size, EDGE_DT)
    ed["offset"] = (yy * W + xx).reshape(-1)
    ed["wd_idx"], ed["dir"], ed["E"], ed["I"], ed["H"] = 1, d, 60, 20, 2
    dev_ed = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev)
    ms = timed(lambda: vp9.loop_filter_batch(pic, W, dev_ed, ed.size))
    print(json.dumps({"case": "vp9 loop_filter_8 (8 wide), every 8x8-grid %s edge of %d 4K planes" % (name, planes), "segments": int(ed.size),
                      "ms": round(ms, 4), "Gpixel/s": round(planes * W * H / ms / 1e6, 1)}), flush=True)
