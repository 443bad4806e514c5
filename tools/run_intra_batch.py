#!/usr/bin/env python3
"""tools/run_intra_batch.py [npics=32] [launches=4] [mb_w=120 mb_h=68 [type]] — ffhip_h264_intra_frames_dev on npics all-intra pictures
(1080p unless given; the same records, each picture its own planes), for profiler passes and timing.  type 0..3 = Intra16x16 / Intra4x4 /
Intra4x4 with the 8x8 transform / I_PCM only (a picture of ONE macroblock row has no hand-off: the time per macroblock of the chain itself).
FFHIP_BUILD=measure selects the build in which the FFHIP_* knobs are live."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ffmpeg_amd import _lib  # noqa: E402
import h264_intra_gen as G  # noqa: E402

npics = int(sys.argv[1]) if len(sys.argv) > 1 else 32
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mb_w, mb_h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (120, 68)
mtype = [G.I16, G.I4, G.I8, G.PCM][int(sys.argv[5])] if len(sys.argv) > 5 else None
if os.environ.get("FFHIP_BUILD") == "measure":  # the build in which the FFHIP_* knobs are live
    _lib.select("measure")
L = _lib.lib()
L.ffhip_h264_intra_pack.restype = C.c_int


class IntraPic(C.Structure):
    _fields_ = [("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p), ("recs", C.c_void_p), ("row_start", C.c_void_p), ("coefs", C.c_void_p)]


rng = np.random.default_rng(6)
coefs, ncoef, recs = np.zeros(mb_w * mb_h * 400, np.int16), 0, []
for my in range(mb_h):
    for mx in range(mb_w):
        d = G.make_intra_mb(rng, mx, my, mb_w, mb_h, mtype)
        rec = G.to_record(d)
        mb = d["mb"].copy()
        n = C.c_int32(ncoef)
        assert L.ffhip_h264_intra_pack(rec.ctypes.data, d["nnzc"].ctypes.data, mb.ctypes.data, d["luma_dc"].ctypes.data, G._p(d["pcm"], C.c_uint8),
                                       G._p(coefs, C.c_int16), C.byref(n), C.c_int32(coefs.size)) == 0
        ncoef = n.value
        recs.append(rec)
d_rec = torch.from_numpy(np.concatenate(recs).view(np.uint8).reshape(-1, 108).copy()).cuda()
d_rows = torch.from_numpy(np.arange(mb_h + 1, dtype=np.int32) * mb_w).cuda()
d_coef = torch.from_numpy(coefs[:ncoef].copy()).cuda()
sy, sc = mb_w * 16, mb_w * 8
planes = [[torch.zeros((mb_h * 16, sy), dtype=torch.uint8, device="cuda"), torch.zeros((mb_h * 8, sc), dtype=torch.uint8, device="cuda"),
           torch.zeros((mb_h * 8, sc), dtype=torch.uint8, device="cuda")] for _ in range(npics)]
arr = (IntraPic * npics)(*[IntraPic(p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), d_rec.data_ptr(), d_rows.data_ptr(), d_coef.data_ptr()) for p in planes])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
_lib.check(L.ffhip_h264_intra_frames_dev(8, npics, C.cast(arr, C.c_void_p), sy, sc, mb_w, mb_h, None), "ffhip_h264_intra_frames_dev")
torch.cuda.synchronize()
e0.record()
for _ in range(launches):
    _lib.check(L.ffhip_h264_intra_frames_dev(8, npics, C.cast(arr, C.c_void_p), sy, sc, mb_w, mb_h, None), "ffhip_h264_intra_frames_dev")
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / launches
print(json.dumps({"pictures_per_launch": npics, "ms_per_launch": round(ms, 4), "pictures_per_s": round(1e3 * npics / ms, 1),
                  "us_per_step": round(1e3 * ms / (mb_w + 2 * (mb_h - 1)), 3)}))
