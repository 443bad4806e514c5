#!/usr/bin/env python3
"""tools/plane_skew_sweep.py — the headline launch (nv12 1080p -> 4K, 256 frames) with all four plane batches carved from ONE allocation:
source luma, source chroma and destination luma at fixed places, the destination CHROMA batch at a varying distance behind the luma batch.
If the rate follows the distance, the relative placement of the two destination streams decides between the kernel's two modes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffmpeg_amd import swscale as S

dev = torch.device("cuda:0")
MB = 1 << 20
n = 256
ctx = S.SwsContext(1920, 1080, 23, 3840, 2160, 23, 4)


def timed(src, dst, reps=40):
    for _ in range(8):
        ctx.scale_batch(src, dst)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ctx.scale_batch(src, dst)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def view(arena, at, shape):
    k = 1
    for s_ in shape:
        k *= s_
    return arena[at:at + k].view(*shape)


def up(x, a):
    return (x + a - 1) // a * a


for arena_no in range(2):
    arena = torch.empty(12 << 30, dtype=torch.uint8, device=dev)
    arena[: 1 << 30].random_(0, 256)
    sy, sc = (n, 1080, 1920), (n, 540, 1920)
    dy, dc = (n, 2160, 3840), (n, 1080, 3840)
    at = 0
    src_y = view(arena, at, sy); at = up(at + n * 1080 * 1920, 2 * MB)
    src_c = view(arena, at, sc); at = up(at + n * 540 * 1920, 2 * MB)
    dst_y = view(arena, at, dy); y_end = at + n * 2160 * 3840
    base_c = up(y_end, 2 * MB)
    for skew in [0, 4096, 65536, 256 * 1024, MB, 2 * MB, 3 * MB, 4 * MB, 6 * MB, 8 * MB, 12 * MB, 16 * MB, 24 * MB, 32 * MB, 48 * MB, 64 * MB, 96 * MB, 128 * MB,
                 192 * MB, 256 * MB, 384 * MB, 512 * MB, 768 * MB, 1024 * MB, 1536 * MB, 2048 * MB]:
        dst_c = view(arena, base_c + skew, dc)
        ms = timed([src_y, src_c], [dst_y, dst_c])
        print(json.dumps({"arena": arena_no, "base": hex(arena.data_ptr()), "chroma_minus_luma_MB": round((base_c + skew - (y_end - n * 2160 * 3840)) / MB, 3),
                          "skew": skew, "hbm_frac": round(n * 15552000 / (ms * 1e-3) / 8e12, 4)}), flush=True)
    keep = arena
