#!/usr/bin/env python3
"""tools/arena_offset_sweep.py — is the allocation dependence of the streaming kernels a function of ADDRESS BITS or of where the driver
found physical memory?  One arena (one hipMalloc), the source planes at its start, the destination at arena + D0 + off for a list of offsets;
yuv420p -> rgb24 4K (64 frames) with plain and eighth-per-XCD numbering, and the headline nv12 1080p -> 4K launch (256 frames).  Then the same
offsets in a SECOND arena.  If hbm_frac follows the offset in both arenas alike, virtual address bits decide (and a launch can choose its
numbering from them); if it follows the arena, physical placement does."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffmpeg_amd import _lib
_lib.select("measure")
from ffmpeg_amd import swscale as S

dev = torch.device("cuda:0")
MB = 1 << 20
offs = [0, 2 * MB, 4 * MB, 8 * MB, 16 * MB, 32 * MB, 64 * MB, 128 * MB, 256 * MB, 512 * MB, 1024 * MB, 6 * MB, 74 * MB, 1030 * MB]


def timed(ctx, src, dst, reps=40):
    for _ in range(8):
        ctx.scale_batch(src, dst)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ctx.scale_batch(src, dst)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def carve(arena, at, shape):
    n = 1
    for s in shape:
        n *= s
    return arena[at:at + n].view(*shape), at + ((n + 2 * MB - 1) // (2 * MB)) * 2 * MB


for arena_no in range(2):
    arena = torch.empty(9 << 30, dtype=torch.uint8, device=dev)
    arena.random_(0, 256)
    base = arena.data_ptr()
    # ---- rgb24
    n, w, h = 64, 3840, 2160
    ctx = S.SwsContext(w, h, 0, w, h, 2, 4)
    at, src = 0, []
    for r, c in S.plane_shapes(0, w, h):
        t, at = carve(arena, at, (n, r, c))
        src.append(t)
    d0 = at
    for off in offs:
        dst, _ = carve(arena, d0 + off, (n, h, 3 * w))
        row = {}
        for label, val in (("plain", None), ("eighth", "xcd")):
            if val:
                os.environ["FFHIP_YUV2RGB_VARIANT"] = val
            else:
                os.environ.pop("FFHIP_YUV2RGB_VARIANT", None)
            ms = timed(ctx, src, [dst])
            row[label] = round(n * w * h * 4.5 / (ms * 1e-3) / 8e12, 4)
        os.environ.pop("FFHIP_YUV2RGB_VARIANT", None)
        print(json.dumps({"kernel": "rgb24", "arena": arena_no, "base": hex(base), "dst_off_MB": off // MB, "dst": hex(dst.data_ptr()), **row}), flush=True)
    ctx.close()
    # ---- up2
    n = 256
    ctx = S.SwsContext(1920, 1080, 23, 3840, 2160, 23, 4)
    at, src = 0, []
    for r, c in S.plane_shapes(23, 1920, 1080):
        t, at = carve(arena, at, (n, r, c))
        src.append(t)
    d0 = at
    for off in offs:
        a2, dst = d0 + off, []
        for r, c in S.plane_shapes(23, 3840, 2160):
            t, a2 = carve(arena, a2, (n, r, c))
            dst.append(t)
        row = {}
        for label, val in (("eighth", None), ("plain", "0")):
            if val:
                os.environ["FFHIP_UP2_XCD"] = val
            else:
                os.environ.pop("FFHIP_UP2_XCD", None)
            ms = timed(ctx, src, dst)
            row[label] = round(n * 15552000 / (ms * 1e-3) / 8e12, 4)
        os.environ.pop("FFHIP_UP2_XCD", None)
        print(json.dumps({"kernel": "up2", "arena": arena_no, "base": hex(base), "dst_off_MB": off // MB, "dst": hex(dst[0].data_ptr()), **row}), flush=True)
    ctx.close()
    keep = arena if arena_no == 0 else None      # the second arena must not reuse the first one's memory
