"""tools/bench_hevc_qm.py [planes] — put_hevc_qpel_uni over every 16 x 16 block of N 4K planes, mixed (mx, my), displacements +-8:
k_hevc_qpel_m (matrix cores) against k_hevc_mc (FFHIP_HEVC_MC_M=0) and its whole-footprint variant (FFHIP_HEVC_QM_FULL=1), alternating
passes in one process (the measure build reads the knobs at every launch); per position as well."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffmpeg_amd import _lib, hevc

_lib.select("measure")
planes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W, H, P = 3840, 2160, 16
dev = "cuda:0"
rng = np.random.default_rng(7)
pic = torch.zeros((planes * H, W), dtype=torch.uint8, device=dev)
ref = torch.randint(0, 256, (planes * H + 2 * P, W + 2 * P), dtype=torch.uint8, device=dev)
by, bx = np.meshgrid(np.arange(0, planes * H, 16), np.arange(0, W, 16), indexing="ij")
n = by.size


def records(mx=None, my=None):
    mc = np.zeros(n, hevc.MC_DTYPE)
    mc["dst_offset"] = (by * W + bx).reshape(-1)
    mc["src_offset"] = ((by + P + rng.integers(-8, 9, by.shape)) * (W + 2 * P) + bx + P + rng.integers(-8, 9, by.shape)).reshape(-1)
    mc["width"] = mc["height"] = 16
    mc["mx"] = rng.integers(0, 4, n) if mx is None else mx
    mc["my"] = rng.integers(0, 4, n) if my is None else my
    return torch.from_numpy(mc.view(np.uint8).reshape(-1, 12)).to(dev)


def timed(d_mc, iters=20):
    for _ in range(3):
        hevc.mc_batch(0, 1, pic, W, ref, W + 2 * P, d_mc, n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        hevc.mc_batch(0, 1, pic, W, ref, W + 2 * P, d_mc, n)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


mixed = records()
variants = (("matrix cores", {}), ("matrix cores, whole footprints", {"FFHIP_HEVC_QM_FULL": "1"}), ("k_hevc_mc", {"FFHIP_HEVC_MC_M": "0"}))
want = None
for p in range(3):
    for name, env in variants:
        for k in ("FFHIP_HEVC_QM_FULL", "FFHIP_HEVC_MC_M"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ms = timed(mixed)
        got = int(pic.to(torch.int64).sum().item())
        want = got if want is None else want
        print(json.dumps({"pass": p, "kernel": name, "planes": planes, "ms": round(ms, 4), "hbm_frac": round(2 * planes * W * H / ms / 1e6 / 8000, 4),
                          "same_pixels": got == want}), flush=True)
for k in ("FFHIP_HEVC_QM_FULL", "FFHIP_HEVC_MC_M"):
    os.environ.pop(k, None)
for mx, my in ((0, 0), (2, 0), (0, 2), (2, 2), (1, 3)):
    ms = timed(records(mx, my))
    print(json.dumps({"position": [mx, my], "ms": round(ms, 4), "hbm_frac": round(2 * planes * W * H / ms / 1e6 / 8000, 4)}), flush=True)

# a quadtree-like partition (round 6, second step): every 32 x 32 cell is one 32 x 32 block, four 16 x 16, two 32 x 16 / 16 x 32 or sixteen 8 x 8
cells_y, cells_x = np.meshgrid(np.arange(0, planes * H, 32), np.arange(0, W, 32), indexing="ij")
kind = rng.integers(0, 5, cells_y.shape)
recs = []
for k, parts in enumerate(([(0, 0, 32, 32)], [(0, 0, 16, 16), (0, 16, 16, 16), (16, 0, 16, 16), (16, 16, 16, 16)], [(0, 0, 32, 16), (16, 0, 32, 16)],
                           [(0, 0, 16, 32), (0, 16, 16, 32)], [(y, x, 8, 8) for y in range(0, 32, 8) for x in range(0, 32, 8)])):
    cy, cx = cells_y[kind == k], cells_x[kind == k]
    for (oy, ox, w, h) in parts:
        m = cy + oy < planes * H
        recs.append(np.stack([cy[m] + oy, cx[m] + ox, np.full(m.sum(), w), np.full(m.sum(), h)], 1))
recs = np.concatenate(recs)
recs = recs[np.lexsort((recs[:, 1], recs[:, 0]))]
nq = len(recs)
mcq = np.zeros(nq, hevc.MC_DTYPE)
mcq["dst_offset"] = recs[:, 0] * W + recs[:, 1]
mcq["src_offset"] = (recs[:, 0] + P + rng.integers(-8, 9, nq)) * (W + 2 * P) + recs[:, 1] + P + rng.integers(-8, 9, nq)
mcq["width"], mcq["height"] = recs[:, 2], recs[:, 3]
mcq["mx"], mcq["my"] = rng.integers(0, 4, nq), rng.integers(0, 4, nq)
d_q = torch.from_numpy(mcq.view(np.uint8).reshape(-1, 12)).to(dev)


def timed_q(iters=20):
    for _ in range(3):
        hevc.mc_batch(0, 1, pic, W, ref, W + 2 * P, d_q, nq)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        hevc.mc_batch(0, 1, pic, W, ref, W + 2 * P, d_q, nq)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


want = None
for p in range(2):
    for name, env in (("product (16 x 16 on the matrix cores, the rest on k_hevc_mc)", {}), ("k_hevc_mc", {"FFHIP_HEVC_MC_M": "0"})):
        for k in ("FFHIP_HEVC_QM_FULL", "FFHIP_HEVC_MC_M"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ms = timed_q()
        got = int(pic.to(torch.int64).sum().item())
        want = got if want is None else want
        print(json.dumps({"partition": "8x8 .. 32x32 mixed", "blocks": nq, "pass": p, "kernel": name, "ms": round(ms, 4),
                          "hbm_frac": round(2 * planes * W * H / ms / 1e6 / 8000, 4), "same_pixels": got == want}), flush=True)
