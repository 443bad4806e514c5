#!/usr/bin/env python3
"""tools/bench_hevc_idct.py — HEVC inverse transform + add_residual over 16 4K luma planes (6 B per sample: coefficients read,
residual written in place, picture read + written): 32x32 on the matrix cores vs the dot2 kernel (FFHIP_HEVC_IDCT32_VALU=1), 16x16 on the
dot2 kernel vs two units per MFMA (FFHIP_HEVC_IDCT16_MFMA=1), 8x8."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
from ffmpeg_amd import hevc  # noqa: E402

dev = torch.device("cuda", 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
for lg, planes, valu, bd in ((5, 16, "0", 8), (5, 16, "1", 8), (5, 16, "0", 10), (4, 16, "0", 8), (4, 16, "m", 8), (4, 16, "0", 10), (3, 8, "0", 8)):
    os.environ["FFHIP_HEVC_IDCT32_VALU"] = valu
    os.environ["FFHIP_HEVC_IDCT16_MFMA"] = "1" if valu == "m" else "0"
    nsz, ps = 1 << lg, 2 if bd > 8 else 1
    bw, bh = 3840 // nsz, 2160 // nsz
    ntu = planes * bw * bh
    tus = np.zeros(ntu, hevc.TU_DTYPE)
    idx = np.arange(ntu)
    pl, rem = idx // (bw * bh), idx % (bw * bh)
    tus["coeff_offset"] = idx * nsz * nsz
    tus["dst_offset"] = ps * (pl * 3840 * 2160 + (rem // bw) * nsz * 3840 + (rem % bw) * nsz)
    tus["col_limit"] = nsz
    d_t = torch.from_numpy(tus.view(np.uint8).reshape(ntu, 12).copy()).to(dev)
    c0 = torch.randint(-512, 512, (ntu, nsz * nsz), dtype=torch.int16, device=dev)
    pic = torch.randint(0, 256, (planes * 2160, 3840 * ps), dtype=torch.uint8, device=dev)
    cc = c0.clone()
    hevc.idct_batch(hevc.IDCT, lg, cc, pic, 3840 * ps, d_t, ntu, bit_depth=bd)
    tot = 0.0
    for _ in range(5):
        cc.copy_(c0)
        e0, e1 = ev(), ev()
        e0.record()
        hevc.idct_batch(hevc.IDCT, lg, cc, pic, 3840 * ps, d_t, ntu, bit_depth=bd)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    ms = tot / 5
    gbs = ntu * nsz * nsz * (4 + 2 * ps) / (ms * 1e-3) / 1e9
    print(json.dumps({"case": "hevc idct%d + add_residual, %d-bit%s" % (nsz, bd, " (dot2 kernel)" if valu == "1" else " (two units per MFMA)" if valu == "m" else ""),
                      "Mblocks/s": round(ntu / (ms * 1e-3) / 1e6, 1), "GB/s": round(gbs, 1), "hbm_frac": round(gbs / 8000, 4), "blocks": ntu,
                      "ms": round(ms, 4)}), flush=True)
    del cc, c0, pic, d_t
