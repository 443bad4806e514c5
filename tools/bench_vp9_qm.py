"""tools/bench_vp9_qm.py [planes] — VP9 motion compensation over every 16 x 16 block of N 4K planes (8-tap sets mixed / bilinear, all
(mx, my), put, displacements +-8): k_vp9_mc_m (matrix cores) against k_vp9_mc (FFHIP_VP9_MC_M=0), alternating passes in one process."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffmpeg_amd import _lib, vp9

_lib.select("measure")
planes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W, H, P = 3840, 2160, 16
dev = "cuda:0"
rng = np.random.default_rng(7)
pic = torch.zeros((planes * H, W), dtype=torch.uint8, device=dev)
ref = torch.randint(0, 256, (planes * H + 2 * P, W + 2 * P), dtype=torch.uint8, device=dev)
by, bx = np.meshgrid(np.arange(0, planes * H, 16), np.arange(0, W, 16), indexing="ij")
n = by.size


def records(filt, mx=None, my=None):
    mc = np.zeros(n, vp9.MC_DTYPE)
    mc["dst_offset"] = (by * W + bx).reshape(-1)
    mc["src_offset"] = ((by + P + rng.integers(-8, 9, by.shape)) * (W + 2 * P) + bx + P + rng.integers(-8, 9, by.shape)).reshape(-1)
    mc["width"] = mc["height"] = 16
    mc["filter"] = rng.integers(0, 3, n) if filt is None else filt
    mc["mx"] = rng.integers(0, 16, n) if mx is None else mx
    mc["my"] = rng.integers(0, 16, n) if my is None else my
    return torch.from_numpy(mc.view(np.uint8).reshape(-1, 16)).to(dev)


def timed(d_mc, iters=20):
    for _ in range(3):
        vp9.mc_batch(pic, W, ref, W + 2 * P, d_mc, n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        vp9.mc_batch(pic, W, ref, W + 2 * P, d_mc, n)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for name, filt in (("8-tap (3 sets mixed)", None), ("bilinear", 3)):
    d_mc = records(filt)
    want = None
    for p in range(3):
        for kern, env in (("matrix cores", None), ("k_vp9_mc", "0")):
            os.environ.pop("FFHIP_VP9_MC_M", None)
            if env:
                os.environ["FFHIP_VP9_MC_M"] = env
            ms = timed(d_mc)
            got = int(pic.to(torch.int64).sum().item())
            want = got if want is None else want
            print(json.dumps({"filter": name, "pass": p, "kernel": kern, "planes": planes, "ms": round(ms, 4),
                              "hbm_frac": round(2 * planes * W * H / ms / 1e6 / 8000, 4), "same_pixels": got == want}), flush=True)
os.environ.pop("FFHIP_VP9_MC_M", None)
for mx, my in ((0, 0), (8, 0), (0, 8), (8, 8)):
    ms = timed(records(1, mx, my))
    print(json.dumps({"filter": "regular", "position": [mx, my], "ms": round(ms, 4), "hbm_frac": round(2 * planes * W * H / ms / 1e6 / 8000, 4)}), flush=True)
