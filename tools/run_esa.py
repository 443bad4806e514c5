#!/usr/bin/env python3
"""tools/run_esa.py [kind=0] [R=7] [pairs=8] — the full-search kernel alone on 4K frame pairs (for rocprofv3 passes)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib, me  # noqa: E402

if any(k.startswith("FFHIP_ME_") for k in os.environ):   # a knob: the measure build (the product library reads no environment)
    _lib.select("measure")

kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
R = int(sys.argv[2]) if len(sys.argv) > 2 else 7
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 8
w, h, mb = 3840, 2160, 16
dev = torch.device("cuda", 0)
cur = torch.randint(0, 256, (nf, h, w), dtype=torch.uint8, device=dev)
ref = torch.roll(cur, shifts=(2, -3), dims=(1, 2)).contiguous()
mv = torch.zeros((nf, (h // mb) * (w // mb), 2), dtype=torch.int16, device=dev)
cost = torch.zeros((nf, (h // mb) * (w // mb)), dtype=torch.int32, device=dev)
for _ in range(2):
    me.esa_batch(cur, ref, w, h, w, h * w, nf, mb, R, kind, mv, cost)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    me.esa_batch(cur, ref, w, h, w, h * w, nf, mb, R, kind, mv, cost)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
n = nf * (h // mb) * (w // mb)
print(json.dumps({"kind": kind, "R": R, "pairs": nf, "ms": round(ms, 4), "M MB-searches/s": round(n / ms / 1e3, 1)}))
