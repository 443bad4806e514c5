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
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
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
for name, sg in (("vertical", segs[0]), ("horizontal", segs[1])):
    ms1 = timed(lambda: hevc.loop_filter_batch(pic, W, sg, sg.shape[0]))
    print(json.dumps({"case": "  the %s edges alone" % name, "segments": int(sg.shape[0]), "ms": round(ms1, 4),
                      "Gpixel/s": round(planes * W * H / ms1 / 1e6, 1)}), flush=True)
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
os.environ["FFHIP_HEVC_MC_OLD"] = "1"
ms_old = timed(lambda: hevc.mc_batch(0, 1, pic, W, ref, W + 2 * P, d_mc, n))
os.environ["FFHIP_HEVC_MC_OLD"] = "0"
print(json.dumps({"case": "  same, the first (sample-per-lane) kernel", "ms": round(ms_old, 4), "Gpixel/s": round(planes * W * H / ms_old / 1e6, 1)}),
      flush=True)
# ---- bi-directional prediction as the decoder runs it: put_hevc_qpel (list 0 -> int16) then put_hevc_qpel_bi_w / _bi (list 1 + int16 -> pixels)
# the int16 intermediates packed four 16-wide blocks to a 64-element row group (rows are MAX_PB_SIZE = 64 elements apart)
tmp16 = torch.empty((n // 4 + 1, 16, 64), dtype=torch.int16, device=dev)
put = mc.copy()
put["dst_offset"] = (np.arange(n) // 4) * 1024 + (np.arange(n) % 4) * 16
d_put = torch.from_numpy(put.view(np.uint8).reshape(-1, 12)).to(dev)
mw = np.zeros(n, hevc.MCW_DTYPE)
for f in ("dst_offset", "width", "height"):
    mw[f] = mc[f]
mw["src_offset"] = ((by + P + rng.integers(-8, 9, by.shape)) * (W + 2 * P) + bx + P + rng.integers(-8, 9, by.shape)).reshape(-1)
mw["mx"], mw["my"] = rng.integers(0, 4, n), rng.integers(0, 4, n)
mw["src2_offset"] = put["dst_offset"]
mw["denom"], mw["wx0"], mw["wx1"], mw["ox"] = 6, 70, 58, 3
d_mw = torch.from_numpy(mw.view(np.uint8).reshape(-1, 24)).to(dev)
for name, mode in (("put_hevc_qpel + put_hevc_qpel_bi", hevc.MC_BI), ("put_hevc_qpel + put_hevc_qpel_bi_w", hevc.MC_BI_W)):
    def both():
        hevc.mc_batch(0, 0, tmp16, 0, ref, W + 2 * P, d_put, n)
        hevc.mc_w_batch(0, mode, pic, W, ref, W + 2 * P, tmp16, d_mw, n)
    ms = timed(both)
    print(json.dumps({"case": "hevc %s, every 16x16 block of %d 4K planes" % (name, planes), "blocks": n, "ms": round(ms, 4),
                      "Gpixel/s": round(planes * W * H / ms / 1e6, 1)}), flush=True)
ms = timed(lambda: hevc.mc_w_batch(0, hevc.MC_UNI_W, pic, W, ref, W + 2 * P, None, d_mw, n))
print(json.dumps({"case": "hevc put_hevc_qpel_uni_w, every 16x16 block of %d 4K planes" % planes, "blocks": n, "ms": round(ms, 4),
                  "Gpixel/s": round(planes * W * H / ms / 1e6, 1)}), flush=True)
# chroma: 8x8 blocks of the half-size planes, eighth-sample positions
cm = np.zeros(n, hevc.MC_DTYPE)
cm["dst_offset"] = ((by // 2) * W + bx // 2).reshape(-1)
cm["src_offset"] = ((by // 2 + P + rng.integers(-4, 5, by.shape)) * (W + 2 * P) + bx // 2 + P + rng.integers(-4, 5, by.shape)).reshape(-1)
cm["width"] = cm["height"] = 8
cm["mx"], cm["my"] = rng.integers(0, 8, n), rng.integers(0, 8, n)
d_cm = torch.from_numpy(cm.view(np.uint8).reshape(-1, 12)).to(dev)
ms = timed(lambda: hevc.mc_batch(1, 1, pic, W, ref, W + 2 * P, d_cm, n))
print(json.dumps({"case": "hevc put_hevc_epel_uni, %d 8x8 chroma blocks, mixed (mx, my)" % n, "blocks": n, "ms": round(ms, 4),
                  "Gpixel/s": round(n * 64 / ms / 1e6, 1)}), flush=True)
