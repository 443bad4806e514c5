#!/usr/bin/env python3
"""tools/bench_h264_picture.py — the caller-side batching layer on a 4K P-picture (240x135 MBs, 4:2:0): every macroblock 16x16
uni-predicted at a random quarter-sample position, about half of its 8x8 luma / a third of its 4x4 chroma blocks carry a
residual, every edge is filtered.  Times flush() — one record upload + the launches — with HIP events; the recording itself is
done here through ctypes (a decoder appends records in C)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ffmpeg_amd import _lib, h264  # noqa: E402
if os.environ.get("FFHIP_MEASURE_LIB") == "1":   # the -DFFHIP_MEASURE build: FFHIP_INTRA_WPB and the other knobs are live there
    _lib.select("measure")
import h264_intra_gen as G  # noqa: E402
from test_gpu_h264_picture import QPEL_DT, CHROMA_DT, EDGE_DT  # noqa: E402

dev = torch.device("cuda", 0)
mb_w, mb_h = (int(v) for v in os.environ.get("BENCH_MB", "240x135").split("x"))   # 4K unless told otherwise
P = 32
MTYPE = int(os.environ["BENCH_MTYPE"]) if "BENCH_MTYPE" in os.environ else None   # 0 Intra16x16, 1 4x4, 2 8x8, 3 I_PCM; default mixed
NODEBLOCK = os.environ.get("BENCH_NODEBLOCK") == "1"   # the reconstruction wavefront alone
W, H = mb_w * 16, mb_h * 16
sy, sc = W + 2 * P, W // 2 + P
rng = np.random.default_rng(4)
DEPTH = int(os.environ.get("BENCH_DEPTH", "8"))   # 9 / 10 / 12 / 14: uint16_t samples, int32 coefficients, byte offsets and strides
PS = 2 if DEPTH > 8 else 1
CDT = np.int32 if DEPTH > 8 else np.int16


def plane(rows, cols, rand):
    if PS == 1:
        return torch.randint(0, 256, (rows, cols), dtype=torch.uint8, device=dev) if rand else torch.zeros((rows, cols), dtype=torch.uint8, device=dev)
    a = torch.randint(0, 1 << DEPTH, (rows, cols), dtype=torch.int16, device=dev) if rand else torch.zeros((rows, cols), dtype=torch.int16, device=dev)
    return a.view(torch.uint8)


refs = [plane(H + 2 * P, sy, 1), plane(H // 2 + P, sc, 1), plane(H // 2 + P, sc, 1)]
dst = [plane(H, sy, 0), plane(H // 2, sc, 0), plane(H // 2, sc, 0)]
pic = h264.Picture(mb_w, mb_h, bit_depth=DEPTH)


def record():
    pic.begin()
    n = {"mc": 0, "idct": 0}
    q, c = np.zeros(1, QPEL_DT), np.zeros(1, CHROMA_DT)
    ed8, ed4 = np.zeros(8, EDGE_DT), np.zeros(4, EDGE_DT)
    for e in (ed8, ed4):
        e["alpha"], e["beta"] = 40, 9
        e["tc0"] = 1
    ed4["kind"] = 2
    blk8, blk4 = np.zeros(64, CDT), np.zeros(16, CDT)
    for my in range(mb_h):
        for mx in range(mb_w):
            x, y = mx * 16, my * 16
            dy, dx = (int(v) for v in rng.integers(-16, 17, 2))
            q[0] = (PS * (y * sy + x), PS * ((P + y + dy) * sy + P + x + dx), int(rng.integers(0, 16)), 0, 0, 0, 0, 0)
            pic.mc_luma(h264.MC_PUT, q)
            for pl in (1, 2):
                c[0] = (PS * ((y // 2) * sc + x // 2), PS * ((P // 2 + y // 2 + dy // 2) * sc + P // 2 + x // 2 + dx // 2), 0, 8, int(rng.integers(0, 8)),
                        int(rng.integers(0, 8)), 0, 0, 0, 0, 0)
                pic.mc_chroma(pl, h264.MC_PUT, c)
            n["mc"] += 3
            for by in (0, 8):
                for bx in (0, 8):
                    if rng.random() < .5:
                        blk8[:] = 0
                        blk8[:6] = rng.integers(-80, 81, 6)
                        pic.idct_add(0, 1, PS * ((y + by) * sy + x + bx), blk8)
                        n["idct"] += 1
            for pl in (1, 2):
                for by in (0, 4):
                    for bx in (0, 4):
                        if rng.random() < .3:
                            blk4[:] = 0
                            blk4[:3] = rng.integers(-80, 81, 3)
                            pic.idct_add(pl, 0, PS * ((y // 2 + by) * sc + x // 2 + bx), blk4)
                            n["idct"] += 1
            pic.deblock_mb(0, mx, my, ed8)
            pic.deblock_mb(1, mx, my, ed4)
            pic.deblock_mb(2, mx, my, ed4)
    return n


def record_intra(frac):
    """an I-picture (frac = 1) or the intra macroblocks of a P-picture: intra macroblocks only, deblocking recorded for all"""
    pic.begin()
    n = 0
    ed8, ed4 = np.zeros(8, EDGE_DT), np.zeros(4, EDGE_DT)
    for e in (ed8, ed4):
        e["alpha"], e["beta"] = 40, 9
        e["kind"] = 4
    ed4["kind"] = 6
    for my in range(mb_h):
        for mx in range(mb_w):
            if rng.random() < frac:
                d = G.make_intra_mb(rng, mx, my, mb_w, mb_h, mtype=MTYPE, depth=DEPTH)
                pic.intra_mb(G.to_record(d), d["nnzc"], d["mb"], d["luma_dc"], d["pcm"])
                n += 1
            if NODEBLOCK:
                continue
            pic.deblock_mb(0, mx, my, ed8)
            pic.deblock_mb(1, mx, my, ed4)
            pic.deblock_mb(2, mx, my, ed4)
    return n


def timed(reps=10):
    pic.flush(dst, strides, refs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pic.flush(dst, strides, refs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


strides = [PS * sy, PS * sc, PS * sc]
if "--intra" in sys.argv:
    for frac in (1.0, .1):
        n = record_intra(frac)
        ms = timed()
        print(json.dumps({"case": "h264 4K picture at %d bits, %d%% intra macroblocks (Intra16x16 / 4x4 / 8x8 mixed, residuals): reconstruction "
                                  "wavefront + frame-order deblock through ffhip_h264_picture_flush" % (DEPTH, round(100 * frac)),
                          "intra_macroblocks": n, "ms_per_picture_gpu": round(ms, 3), "pictures_per_s": round(1e3 / ms, 1)}), flush=True)
    sys.exit(0)

t0 = time.perf_counter()
counts = record()
t_rec = time.perf_counter() - t0
pic.flush(dst, strides, refs)
torch.cuda.synchronize()
reps = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(reps):
    pic.flush(dst, strides, refs)          # the same records again: flush() does not clear
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
ms = e0.elapsed_time(e1) / reps
print(json.dumps({"case": "h264 4K P-picture at %d bits through ffhip_h264_picture_flush (MC + IDCT + frame-order deblock, Y/Cb/Cr)" % DEPTH,
                  "macroblocks": mb_w * mb_h, "mc_calls": counts["mc"], "idct_calls": counts["idct"], "ms_per_picture_gpu": round(ms, 3),
                  "ms_per_picture_wall": round(wall, 3), "pictures_per_s": round(1e3 / ms, 1),
                  "python_recording_s": round(t_rec, 2)}), flush=True)
pic.close()
