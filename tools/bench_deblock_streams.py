#!/usr/bin/env python3
"""tools/bench_deblock_streams.py — N lone 1080p luma planes deblocked in frame order on N streams at once (the picture pipeline's
shape: one launch per plane and picture): round time vs N, product build; FFHIP_DEBLOCK_WPB (measure build) with arg 'measure'."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib  # noqa: E402
if len(sys.argv) > 1 and sys.argv[1] == "measure":
    _lib.select("measure")
from ffmpeg_amd import h264  # noqa: E402

dev = torch.device("cuda", 0)
w, h = 1920, 1088
mbw, mbh = w // 16, h // 16
rng = np.random.default_rng(3)
ed = np.zeros(mbw * mbh * 8, dtype=np.dtype([("o", np.int32), ("k", np.uint8), ("a", np.uint8), ("b", np.uint8), ("p", np.uint8), ("tc", np.int8, 4)]))
ed["a"], ed["b"] = 40, 9
mb_intra = rng.random(mbw * mbh) < .25
k = np.zeros((mbw * mbh, 2, 4), np.uint8)
k[mb_intra, :, 0] = 4
ed["k"] = k.ravel()
ed["tc"] = rng.integers(0, 4, (ed.size, 4))
ded = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev)
NMAX = 32
planes = [torch.randint(100, 140, (h, w), dtype=torch.uint8, device=dev) for _ in range(NMAX)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NMAX)]
torch.cuda.synchronize()
for n in (1, 2, 4, 8, 16, 32):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        evs = []
        for i in range(n):
            streams[i].wait_event(e0)
            h264.deblock_frame(planes[i], w, mbw, mbh, ded, stream=streams[i].cuda_stream)
            ev = torch.cuda.Event()
            ev.record(streams[i])
            evs.append(ev)
        for ev in evs:
            torch.cuda.current_stream().wait_event(ev)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(json.dumps({"planes_in_flight": n, "ms_per_round": round(ms, 3), "ms_per_plane": round(ms / n, 4), "knobs": {k: v for k, v in os.environ.items() if k.startswith("FFHIP_") or k == "GPU_MAX_HW_QUEUES"}}), flush=True)
