#!/usr/bin/env python3
"""tools/bench_vp9_lf_frame.py — ffhip_vp9_loopfilter_frame_dev on a 4K picture (60 x 34 superblocks, 4:2:0): masks and levels from
a random block / transform partition by the decoder's rules (tests/vp9_lf_gen.structured), smooth content so the filters fire.
HIP-event time of the one launch; the per-edge batch face on the same plane beside it (every 8-sample segment of the luma column
edges as one FFHipVp9Edge record — what a caller that re-orders the edges itself would launch)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ffmpeg_amd import _lib  # noqa: E402
if os.environ.get("FFHIP_MEASURE_LIB") == "1":
    _lib.select("measure")  # the FFHIP_VP9_LF_* knobs exist only there
from ffmpeg_amd import vp9  # noqa: E402
import vp9_lf_gen as G  # noqa: E402

bd = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sbc, sbr = 60, 34
rng = np.random.default_rng(9)
lim, mblim = G.filter_lut(2)
filt = np.zeros(sbr * sbc, G.FILTER_DT)
for r in range(sbr):
    for c in range(sbc):
        filt[r * sbc + c] = G.structured(rng, r, c, 8 * sbc, 8 * sbr)
tabs = vp9.lf_sb_tables(filt.view(np.uint8).reshape(sbr * sbc, 192), sbc, sbr, lim, mblim)
entries = int((tabs >> 31).sum())
ps = 1 if bd == 8 else 2
dt = torch.uint8


def plane(h, w):
    base = np.cumsum(rng.integers(-2, 3, (h, w)), axis=1) + 128
    a = np.clip(base << (bd - 8), 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16)
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()


y, u, v = plane(64 * sbr, 64 * sbc), plane(32 * sbr, 32 * sbc), plane(32 * sbr, 32 * sbc)
d_tabs = torch.from_numpy(tabs.view(np.int32)).cuda()
sy, suv = 64 * sbc * ps, 32 * sbc * ps
for _ in range(3):
    vp9.loopfilter_frame(y, u, v, sy, suv, 8 * sbc, 8 * sbr, d_tabs, bit_depth=bd)
torch.cuda.synchronize()
N = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    vp9.loopfilter_frame(y, u, v, sy, suv, 8 * sbc, 8 * sbr, d_tabs, bit_depth=bd)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print(json.dumps({"case": "vp9 loop filter, 4K picture (3840x2176, 4:2:0, %d bits) in superblock order, one launch" % bd, "superblocks": sbc * sbr,
                  "filtered_8_sample_segments": entries, "ms_per_picture_gpu": round(ms, 3), "pictures_per_s": round(1e3 / ms, 1),
                  "us_per_superblock_step": round(1e3 * ms / (sbc + 2 * sbr), 2)}))
# the same picture N times in ONE launch (ffhip_vp9_loopfilter_frames_dev): each copy its own planes
for npl in (8, 16, 32):
    pics = [(y.clone(), u.clone(), v.clone(), d_tabs) for _ in range(npl)]
    for _ in range(2):
        vp9.loopfilter_frames(pics, sy, suv, 8 * sbc, 8 * sbr, bit_depth=bd)
    e0.record()
    for _ in range(4):
        vp9.loopfilter_frames(pics, sy, suv, 8 * sbc, 8 * sbr, bit_depth=bd)
    e1.record()
    torch.cuda.synchronize()
    msb = e0.elapsed_time(e1) / 4
    print(json.dumps({"pictures_per_launch": npl, "ms_per_launch": round(msb, 3), "pictures_per_s": round(npl * 1e3 / msb, 1)}))
    del pics
