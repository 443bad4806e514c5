#!/usr/bin/env python3
"""tools/bench_more.py — the conversions and h264 faces bench.py's headline does not time (one GPU, HIP events)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
from ffmpeg_amd import swscale as S, h264  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def sws(name, sf, sw, sh, df, dw, dh, flags, n):
    ctx = S.SwsContext(sw, sh, S.PIX_FMT[sf], dw, dh, S.PIX_FMT[df], flags)
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(S.PIX_FMT[sf], sw, sh)]
    dst = [torch.empty((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(S.PIX_FMT[df], dw, dh)]
    ms = timed(lambda: ctx.scale_batch(src, dst))
    byt = n * (S.frame_bytes(S.PIX_FMT[sf], sw, sh) + S.frame_bytes(S.PIX_FMT[df], dw, dh))
    print(json.dumps({"case": name, "fast_path": ctx.fast_path, "frames": n, "ms": round(ms, 4),
                      "Mpix/s": round(n * dw * dh / ms / 1e3, 1), "GB/s": round(byt / ms / 1e6, 1),
                      "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
    ctx.close()


sws("nv12 1080p->4K bilinear (padded 2-tap banks)", "nv12", 1920, 1080, "nv12", 3840, 2160, S.SWS_BILINEAR, 128)
sws("nv12 1080p -> yuv420p 1080p (1:1 re-pack)", "nv12", 1920, 1080, "yuv420p", 1920, 1080, S.SWS_BICUBIC, 128)
sws("yuv420p 1080p->4K bicubic", "yuv420p", 1920, 1080, "yuv420p", 3840, 2160, S.SWS_BICUBIC, 128)
sws("nv12 720p->1080p bicubic (1.5x, byte-aligned spans)", "nv12", 1280, 720, "nv12", 1920, 1080, S.SWS_BICUBIC, 256)
sws("yuv420p 720p->1080p rgb24 bicubic (1.5x)", "yuv420p", 1280, 720, "rgb24", 1920, 1080, S.SWS_BICUBIC, 128)
sws("nv12 4K->1080p bicubic (8 x 8 taps, k_sws_lwalk)", "nv12", 3840, 2160, "nv12", 1920, 1080, S.SWS_BICUBIC, 32)
sws("yuv420p 1080p->4K rgb24 bicubic (k_sws_colwalk_rgb)", "yuv420p", 1920, 1080, "rgb24", 3840, 2160, S.SWS_BICUBIC, 32)
sws("nv12 1080p->4K bgr24 bicubic (k_sws_colwalk_rgb)", "nv12", 1920, 1080, "bgr24", 3840, 2160, S.SWS_BICUBIC, 32)
os.environ["FFHIP_CWRGB_DIRECT"] = "1"
sws("yuv420p 1080p->4K rgb24 bicubic (k_sws_colwalk_rgb, direct stores)", "yuv420p", 1920, 1080, "rgb24", 3840, 2160, S.SWS_BICUBIC, 32)
del os.environ["FFHIP_CWRGB_DIRECT"]
os.environ["FFHIP_SWS_FAST"] = "0"
sws("yuv420p 1080p->4K rgb24 bicubic (k_scale_rgb, LDS-tiled)", "yuv420p", 1920, 1080, "rgb24", 3840, 2160, S.SWS_BICUBIC, 32)
del os.environ["FFHIP_SWS_FAST"]
sws("yuv420p 1080p->rgb24 accurate_rnd (1-tap luma + 4-tap chroma banks: k_sws_colwalk_rgb)", "yuv420p", 1920, 1080, "rgb24", 1920, 1080,
    S.SWS_BICUBIC | S.SWS_ACCURATE_RND | S.SWS_BITEXACT, 64)

# the SwsFunc-shaped host face: PCIe + staging included
ctx = S.SwsContext(1920, 1080, 23, 3840, 2160, 23, S.SWS_BICUBIC)
hs = [np.random.default_rng(1).integers(0, 256, (r, c), dtype=np.uint8) for r, c in S.plane_shapes(23, 1920, 1080)]
hd = [np.zeros((r, c), np.uint8) for r, c in S.plane_shapes(23, 3840, 2160)]
ctx.scale(hs, hd)
t0 = time.perf_counter()
for _ in range(10):
    ctx.scale(hs, hd)
ms = (time.perf_counter() - t0) * 100
print(json.dumps({"case": "ffhip_sws_scale host face nv12 1080p->4K (pageable host memory, PCIe + staging included)",
                  "ms_per_frame": round(ms, 3), "Mpix/s": round(3840 * 2160 / ms / 1e3, 1)}), flush=True)
ctx.close()

# h264 batch faces
rng = np.random.default_rng(2)
w, h, P = 1920, 1080, 16
stride = w + 2 * P
ref = torch.randint(0, 256, (h + 2 * P, stride), dtype=torch.uint8, device=dev)
dst = torch.zeros_like(ref)
by, bx = np.meshgrid(np.arange(h // 8), np.arange(w // 8), indexing="ij")
cb = np.zeros(by.size, np.dtype([("d", np.int32), ("s", np.int32), ("w", np.uint8), ("h", np.uint8), ("x", np.uint8), ("y", np.uint8),
                                 ("avg", np.uint8), ("flags", np.uint8), ("sx", np.int16), ("sy", np.int16), ("pad", np.int16)]))
cb["d"] = (P + by.ravel() * 8) * stride + P + bx.ravel() * 8
cb["s"] = cb["d"] + rng.integers(-8, 9, by.size) * stride + rng.integers(-8, 9, by.size)
cb["h"], cb["x"], cb["y"] = 8, rng.integers(0, 8, by.size), rng.integers(0, 8, by.size)
dcb = torch.from_numpy(cb.view(np.uint8).reshape(-1, 20)).to(dev)
ms = timed(lambda: h264.chroma_mc_batch(dst, ref, stride, dcb, cb.size))
print(json.dumps({"case": "h264 chroma mc8, every 8x8 block of a 1920x1080 plane", "blocks": int(cb.size), "ms": round(ms, 4),
                  "Mpix/s": round(cb.size * 64 / ms / 1e3, 1)}), flush=True)
wb = np.zeros((h // 16) * (w // 16), np.dtype([("d", np.int32), ("s", np.int32), ("w", np.uint8), ("h", np.uint8), ("ld", np.uint8),
                                                 ("bi", np.uint8), ("wd", np.int16), ("ws", np.int16), ("of", np.int16), ("pad", np.int16)]))
my, mx = np.meshgrid(np.arange(h // 16), np.arange(w // 16), indexing="ij")
wb["d"] = (P + my.ravel() * 16) * stride + P + mx.ravel() * 16
wb["s"] = wb["d"]
wb["h"], wb["ld"], wb["bi"], wb["wd"], wb["ws"], wb["of"] = 16, 5, 1, 20, 12, 3
dwb = torch.from_numpy(wb.view(np.uint8).reshape(-1, 20)).to(dev)
ms = timed(lambda: h264.weight_batch(dst, ref, stride, dwb, wb.size))
print(json.dumps({"case": "h264 biweight 16x16, every MB of a 1920x1080 plane", "blocks": int(wb.size), "ms": round(ms, 4),
                  "Mpix/s": round(wb.size * 256 / ms / 1e3, 1)}), flush=True)
# loop filter batch: 4 vertical + 4 horizontal luma edges per MB laid out on disjoint 32x16 tiles
ne = (h // 16) * (w // 32)
ed = np.zeros(ne, np.dtype([("o", np.int32), ("k", np.uint8), ("a", np.uint8), ("b", np.uint8), ("p", np.uint8), ("tc", np.int8, 4)]))
ty, tx = np.meshgrid(np.arange(h // 16), np.arange(w // 32), indexing="ij")
ed["o"] = (P + ty.ravel() * 16) * stride + P + tx.ravel() * 32 + 16
ed["k"], ed["a"], ed["b"] = 1, 40, 9
ed["tc"] = 2
ded = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev)
ms = timed(lambda: h264.loop_filter_batch(dst, stride, ded, ne))
print(json.dumps({"case": "h264 h_loop_filter_luma, one edge per 32x16 tile of a 1920x1080 plane", "edges": int(ne), "ms": round(ms, 4),
                  "Medges/s": round(ne / ms / 1e3, 1)}), flush=True)
