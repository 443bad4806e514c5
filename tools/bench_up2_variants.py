#!/usr/bin/env python3
"""tools/bench_up2_variants.py — the headline launch (nv12 1080p -> 4K bicubic, 256 frames) with the product kernel and the measure build's
FFHIP_UP2_VAR variants (1 non-temporal stores; 2 the horizontal bank in SGPRs; 3 the same with six rows in flight), alternating passes,
20 warm-up + 100 timed launches each; every variant's whole output is compared with the product's."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffmpeg_amd import _lib
_lib.select("measure")
from ffmpeg_amd import swscale as S

variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["", "1", "2", "3"]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
n = 256
ctx = S.SwsContext(1920, 1080, 23, 3840, 2160, 23, 4)
src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 1920, 1080)]
dst = [torch.empty((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 3840, 2160)]
os.environ.pop("FFHIP_UP2_VAR", None)
os.environ.pop("FFHIP_UP2_DEPTH", None)
ctx.scale_batch(src, dst)
ref = [d.clone() for d in dst]
for p in range(passes):
    for var in variants:
        v, _, depth = var.partition("d")
        if v:
            os.environ["FFHIP_UP2_VAR"] = v
        else:
            os.environ.pop("FFHIP_UP2_VAR", None)
        if depth:
            os.environ["FFHIP_UP2_DEPTH"] = depth
        else:
            os.environ.pop("FFHIP_UP2_DEPTH", None)
        for d in dst:
            d.zero_()
        for _ in range(20):
            ctx.scale_batch(src, dst)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            ctx.scale_batch(src, dst)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 100
        same = all(torch.equal(x, y) for x, y in zip(dst, ref))
        print(json.dumps({"pass": p, "variant": var or "product", "ms": round(ms, 4), "hbm_frac": round(n * 15552000 / (ms * 1e-3) / 8e12, 4),
                          "same_pixels": same}))
