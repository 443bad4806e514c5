#!/usr/bin/env python3
"""tools/run_aac.py [tns] — 6 calls of the AAC imdct_and_windowing batch (65,536 all-long channel-frames), or with `tns` of the
apply_tns batch (131,072 filters), for profiler passes"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import aac  # noqa: E402

nfr = 65536
if len(sys.argv) > 1 and sys.argv[1] == "tns":
    rng = np.random.default_rng(3)
    rec = np.zeros(2 * nfr, aac.TNS_FILTER_DTYPE)
    rec["frame"] = np.repeat(np.arange(nfr), 2)
    rec["start"][0::2], rec["size"][0::2], rec["inc"][0::2], rec["order"][0::2] = 799, 400, -1, 12
    rec["start"][1::2], rec["size"][1::2], rec["inc"][1::2], rec["order"][1::2] = 100, 300, 1, 7
    rec["coef"] = np.sin(rng.uniform(-1.0, 1.0, (2 * nfr, 20))).astype(np.float32) * 0.5
    d_rec = torch.from_numpy(rec.view(np.uint8).reshape(2 * nfr, 92)).cuda()
    co = torch.randn((nfr, 1024), dtype=torch.float32, device="cuda:0")
    for _ in range(6):
        aac.apply_tns_batch(co, d_rec, 2 * nfr, 1)
else:
    d = np.load(os.path.join(ROOT, "tests", "golden", "aac.npz"))
    ctx = aac.AacImdct([d[k] for k in ("sine_1024", "sine_128", "kbd_long_1024", "kbd_short_128")])
    nch = 2
    co = torch.randn((nfr // nch, nch, 1024), dtype=torch.float32, device="cuda:0") * 1000
    out = torch.empty_like(co)
    saved = torch.zeros((nch, 512), dtype=torch.float32, device="cuda:0")
    seq = np.zeros((nfr // nch, nch), np.uint8)
    z = np.zeros(nch, np.uint8)
    for _ in range(6):
        ctx.batch(co, out, saved, seq, seq + 1, z, z)
torch.cuda.synchronize()
print("ok")
