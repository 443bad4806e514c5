#!/usr/bin/env python3
"""tools/bench_aac.py — AACDecDSP.imdct_and_windowing, 65,536 channel-frames per call (one GPU, HIP events): long windows only and
a realistic mix with transient (eight-short) frames."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ffmpeg_amd import aac  # noqa: E402

d = np.load(os.path.join(ROOT, "tests", "golden", "aac.npz"))
ctx = aac.AacImdct([d[k] for k in ("sine_1024", "sine_128", "kbd_long_1024", "kbd_short_128")])
nch, nframes = 2, 32768
n = nch * nframes
rng = np.random.default_rng(3)
co = torch.randn((nframes, nch, 1024), dtype=torch.float32, device="cuda:0") * 1000
out = torch.empty_like(co)
saved = torch.zeros((nch, 512), dtype=torch.float32, device="cuda:0")
for name, mix in (("ONLY_LONG", False), ("5 % transients (LONG_START, EIGHT_SHORT x2, LONG_STOP)", True)):
    seq = np.zeros((nframes, nch), np.uint8)
    if mix:
        for c in range(nch):
            f = 0
            while f < nframes - 4:
                if rng.random() < 0.0125:
                    seq[f:f + 4, c] = (1, 2, 2, 3)
                    f += 4
                else:
                    f += 1
    kb = np.ones((nframes, nch), np.uint8)
    z = np.zeros(nch, np.uint8)
    for _ in range(2):
        ctx.batch(co, out, saved, seq, kb, z, z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ctx.batch(co, out, saved, seq, kb, z, z)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byt = n * 18432                           # coeffs in, buf out, buf + the previous buf's upper half in, out (aac_api.hip)
    print(json.dumps({"case": "aac imdct_and_windowing, %s" % name, "channel_frames": n, "short_frames": int((seq == 2).sum()), "ms": round(ms, 4),
                      "Mframes/s": round(n / ms / 1e3, 1), "GB/s": round(byt / ms / 1e6, 1), "hbm_frac": round(byt / ms / 1e6 / 8000, 4),
                      "realtime_48k_channels": round(n / ms * 1e3 / (48000 / 1024))}), flush=True)
# apply_tns: two filters per long channel-frame (orders 12 and 7, ~700 of the 1024 coefficients covered), all-pole (decode)
nfr = 65536
rec = np.zeros(2 * nfr, aac.TNS_FILTER_DTYPE)
rec["frame"] = np.repeat(np.arange(nfr), 2)
rec["start"][0::2], rec["size"][0::2], rec["inc"][0::2], rec["order"][0::2] = 799, 400, -1, 12
rec["start"][1::2], rec["size"][1::2], rec["inc"][1::2], rec["order"][1::2] = 100, 300, 1, 7
rec["coef"] = np.sin(rng.uniform(-1.0, 1.0, (2 * nfr, 20))).astype(np.float32) * 0.5
d_rec = torch.from_numpy(rec.view(np.uint8).reshape(2 * nfr, 92)).cuda()
src = torch.randn((nfr, 1024), dtype=torch.float32, device="cuda:0")
co = src.clone()
aac.apply_tns_batch(co, d_rec, 2 * nfr, 1)
ms, reps = 0.0, 5
for _ in range(reps):
    co.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    aac.apply_tns_batch(co, d_rec, 2 * nfr, 1)
    e1.record()
    torch.cuda.synchronize()
    ms += e0.elapsed_time(e1) / reps
byt = nfr * 700 * 8 + 2 * nfr * 92
print(json.dumps({"case": "aac apply_tns, 2 filters per frame (orders 12 / 7, 400 + 300 coefficients)", "channel_frames": nfr, "ms": round(ms, 4),
                  "Mframes/s": round(nfr / ms / 1e3, 1), "GB/s": round(byt / ms / 1e6, 1), "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
