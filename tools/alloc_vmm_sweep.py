#!/usr/bin/env python3
"""tools/alloc_vmm_sweep.py — does the alignment the virtual and physical addresses SHARE decide the streaming rate?  The headline launch
(nv12 1080p -> 4K, 256 frames) and yuv420p -> rgb24 4K (64 frames), several rounds in one process, on buffers from (a) torch's allocator
(hipMalloc: 2 MiB-aligned virtual addresses, whatever physical ones), (b) ffhip_frames_alloc with chunks of 2 MiB / 64 MiB / 1 GiB
(virtual range and every physical chunk aligned to the chunk size)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffmpeg_amd import _lib
_lib.select("measure")
from ffmpeg_amd import swscale as S

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def timed(ctx, src, dst, reps=60):
    for _ in range(10):
        ctx.scale_batch(src, dst)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ctx.scale_batch(src, dst)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def carve(mem, shapes):
    out, at = [], 0
    for sh in shapes:
        n = 1
        for s_ in sh:
            n *= s_
        out.append(mem.tensor(sh, at))
        at += (n + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    return out


def total(shapes):
    t = 0
    for sh in shapes:
        n = 1
        for s_ in sh:
            n *= s_
        t += (n + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    return t


cases = (("up2 nv12 1080p->4K x256", 23, 1920, 1080, 23, 3840, 2160, 256, 256 * 15552000), ("rgb24 yuv420p 4K x64", 0, 3840, 2160, 2, 3840, 2160, 64, 64 * 3840 * 2160 * 4.5))
for name, sf, sw, sh, df, dw, dh, n, byts in cases:
    sshapes = [(n, r, c) for r, c in S.plane_shapes(sf, sw, sh)]
    dshapes = [(n, r, c) for r, c in S.plane_shapes(df, dw, dh)]
    keep = []
    for rnd in range(rounds):
        row = {"case": name, "round": rnd}
        for label, chunk, order in (("torch", None, None), ("lin_2M", 2 << 20, "0"), ("mix_2M", 2 << 20, None), ("lin_16M", 16 << 20, "0"), ("mix_16M", 16 << 20, None),
                                    ("mix_64M", 64 << 20, None), ("mix_256M", 256 << 20, None)):
            if order:
                os.environ["FFHIP_FRAMES_ORDER"] = order
            else:
                os.environ.pop("FFHIP_FRAMES_ORDER", None)
            if chunk is None:
                src = [torch.randint(0, 256, s_, dtype=torch.uint8, device=dev) for s_ in sshapes]
                dst = [torch.empty(s_, dtype=torch.uint8, device=dev) for s_ in dshapes]
                mem = None
            else:
                mem = (_lib.FrameMemory(total(sshapes), chunk), _lib.FrameMemory(total(dshapes), chunk))
                src, dst = carve(mem[0], sshapes), carve(mem[1], dshapes)
                for t in src:
                    t.random_(0, 256)
            ctx = S.SwsContext(sw, sh, sf, dw, dh, df, 4)     # a context per buffer set: the launch tuner decides for THESE buffers
            ms = timed(ctx, src, dst)
            row[label] = round(byts / (ms * 1e-3) / 8e12, 4)
            if df == 2:
                row[label + "_numbering"] = ctx.tuned_numbering
            ctx.close()
            if rnd % 2 == 0:
                keep.append((src, dst, mem))   # every other round leaves its buffers allocated: the next round's land elsewhere
            else:
                del src, dst, mem
        print(json.dumps(row), flush=True)
    del keep
    torch.cuda.empty_cache()
