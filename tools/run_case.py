#!/usr/bin/env python3
"""tools/run_case.py SRCFMT W H DSTFMT W H [flags=4] [frames=32] [reps=5] — run one scaler configuration a few times on
cuda:0 (for rocprofv3 kernel-trace / PMC passes of a single kernel) and print its HIP-event time."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import swscale as S  # noqa: E402

a = sys.argv[1:]
sf, sw, sh, df, dw, dh = a[0], int(a[1]), int(a[2]), a[3], int(a[4]), int(a[5])
S.PIX_FMT.setdefault("p010", 158)        # AV_PIX_FMT_P010LE / YUV420P10LE (include/ffhip.h FFHIP_PIX_FMT_*)
S.PIX_FMT.setdefault("yuv420p10", 62)
flags = int(a[6], 0) if len(a) > 6 else S.SWS_BICUBIC
n = int(a[7]) if len(a) > 7 else 32
reps = int(a[8]) if len(a) > 8 else 5
dev = torch.device("cuda", 0)
ctx = S.SwsContext(sw, sh, S.PIX_FMT[sf], dw, dh, S.PIX_FMT[df], flags)
g = torch.Generator(device=dev).manual_seed(1)
# line pitches rounded up to 64 bytes, as av_frame_get_buffer() lays frames out (the fast kernels want dword-aligned lines)
def planes(fmt, w, h, fill):
    out = []
    for r, c in S.plane_shapes(S.PIX_FMT[fmt], w, h):
        pitch = (c + 63) // 64 * 64
        t_ = torch.randint(0, 256, (n, r, pitch), dtype=torch.uint8, device=dev, generator=g) if fill else torch.zeros((n, r, pitch), dtype=torch.uint8, device=dev)
        out.append(t_[:, :, :c])
    return out


src = planes(sf, sw, sh, True)
dst = planes(df, dw, dh, False)
for name, planes in ((sf, src),):
    if name in ("p010", "yuv420p10"):    # valid 10-bit samples: the low 10 bits of the word (planar) or the high 10 (P010)
        for t_ in planes:
            t_.view(torch.int16).bitwise_and_(0x03FF if name == "yuv420p10" else -64)
ctx.scale_batch(src, dst)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ctx.scale_batch(src, dst)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
byt = n * (S.frame_bytes(S.PIX_FMT[sf], sw, sh) + S.frame_bytes(S.PIX_FMT[df], dw, dh))
print(json.dumps({"case": " ".join(a[:6]), "fast_path": ctx.fast_path, "paths": ctx.paths, "frames": n, "ms": round(ms, 4),
                  "Mpix/s": round(n * dw * dh / ms / 1e3, 1), "GB/s": round(byt / ms / 1e6, 1), "hbm_frac": round(byt / ms / 8e9, 4)}))
