#!/usr/bin/env python3
"""tools/run_deblock.py [frames] — one configuration of the frame-order luma deblocking of 4K planes, a few launches: what the
rocprofv3 passes of tools/gpu.sh profile (kernel stats / PMC) for k_h264_deblock_skew."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import h264  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
w, h = 3840, 2160
mbw, mbh = w // 16, h // 16
rng = np.random.default_rng(3)
ed = np.zeros(mbw * mbh * 8, dtype=np.dtype([("o", np.int32), ("k", np.uint8), ("a", np.uint8), ("b", np.uint8), ("p", np.uint8), ("tc", np.int8, 4)]))
ed["a"], ed["b"] = 40, 9
k = np.zeros((mbw * mbh, 2, 4), np.uint8)
k[rng.random(mbw * mbh) < .25, :, 0] = 4
ed["k"] = k.ravel()
ed["tc"] = rng.integers(0, 4, (ed.size, 4))
ded = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev).repeat(nf, 1)
batch = torch.randint(100, 140, (nf, h, w), dtype=torch.uint8, device=dev)
for _ in range(4):
    h264.deblock_frames(batch, w * h, nf, w, mbw, mbh, ded)
torch.cuda.synchronize()
print("ok", nf)
