#!/usr/bin/env python3
"""tools/xcd_numbering_sweep.py — workgroup numbering against where the buffers lie: yuv420p -> rgb24 4K (64 frames)
and the headline nv12 1080p -> 4K launch (256 frames), each with plain numbering, an eighth of the units per XCD, and XCD-contiguous CHUNKS of
2^k workgroups dealt round-robin (round 6), over several fresh allocations in one process.  Per trial: the buffers' addresses and hbm_frac per
numbering (mean of two measurements of 60 launches); the first trial compares every variant's whole output with the product's."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffmpeg_amd import _lib
_lib.select("measure")
from ffmpeg_amd import swscale as S

dev = torch.device("cuda:0")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
which = sys.argv[2] if len(sys.argv) > 2 else "rgb24,up2"


def timed(ctx, src, dst, reps=60):
    for _ in range(10):
        ctx.scale_batch(src, dst)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ctx.scale_batch(src, dst)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def sweep(name, ctx, mk, knob, variants, byts):
    keep = []
    for t in range(trials):
        src, dst = mk()
        row = {}
        ref = None
        for rep in range(2):
            for label, val in variants:
                if val is None:
                    os.environ.pop(knob, None)
                else:
                    os.environ[knob] = val
                ms = timed(ctx, src, dst)
                row.setdefault(label, []).append(round(byts / (ms * 1e-3) / 8e12, 4))
                if t == 0 and rep == 0:
                    if ref is None:
                        ref = [d.clone() for d in dst]
                    else:
                        assert all(torch.equal(x, y) for x, y in zip(dst, ref)), (name, label, "output differs from the product's")
        os.environ.pop(knob, None)
        print(json.dumps({"kernel": name, "trial": t, "dst": hex(dst[0].data_ptr()), "src": hex(src[0].data_ptr()),
                          **{k: round(sum(v) / len(v), 4) for k, v in row.items()}}), flush=True)
        keep.append((src, dst))          # keep earlier allocations alive: the next trial's buffers land elsewhere
        if len(keep) > 3:
            keep.pop(0)


if "rgb24" in which:
    n, w, h = 64, 3840, 2160
    ctx = S.SwsContext(w, h, 0, w, h, 2, 4)
    sweep("k_yuv420p_rgb24_t", ctx,
          lambda: ([torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(0, w, h)],
                   [torch.empty((n, h, 3 * w), dtype=torch.uint8, device=dev)]),
          "FFHIP_YUV2RGB_VARIANT", [("plain", None), ("eighth", "xcd"), ("c4", "xcd2"), ("c16", "xcd4"), ("c64", "xcd6"), ("c256", "xcd8"), ("c1024", "xcd10")],
          n * w * h * 4.5)
    ctx.close()
if "up2" in which:
    n = 256
    ctx = S.SwsContext(1920, 1080, 23, 3840, 2160, 23, 4)
    sweep("k_sws_up2", ctx,
          lambda: ([torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 1920, 1080)],
                   [torch.empty((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 3840, 2160)]),
          "FFHIP_UP2_XCD", [("eighth", None), ("plain", "0"), ("c4", "3"), ("c16", "5"), ("c64", "7"), ("c256", "9"), ("c1024", "11")],
          n * 15552000)
    ctx.close()
