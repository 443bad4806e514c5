#!/usr/bin/env python3
"""tools/bisect_bench.py <libffhip.so> [reps] — the four extras whose driver numbers slipped between rounds 3 and 4,
timed on ONE library build given by path: sws nv12 4K -> 1080p (k_sws_down2), sws yuv444p 1080p -> 4K (k_sws_up2), h264 idct8_add,
hevc idct32 + add.  20 warm-up launches, then `reps` launches between two events, per case; one JSON line.  tools/bisect_regression.sh runs
it for HEAD's library, HEAD built without -mllvm -amdgpu-mfma-vgpr-form and round 3's library (590d041), alternating, on one box."""
import json
import os
import sys

import numpy as np


def main():
    so = os.path.abspath(sys.argv[1])
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from ffmpeg_amd import _lib
    _lib.SO = so
    from ffmpeg_amd import swscale as S, h264
    dev = torch.device("cuda:0")
    ev = lambda: torch.cuda.Event(enable_timing=True)
    out = {"lib": os.path.basename(so)}

    def timed(fn, prep=None):
        for _ in range(20):
            if prep:
                prep()
            fn()
        if prep:   # the call consumes its input: time launches one by one
            tot = 0.0
            for _ in range(reps):
                prep()
                a, b = ev(), ev()
                a.record(); fn(); b.record()
                torch.cuda.synchronize()
                tot += a.elapsed_time(b)
            return tot / reps
        a, b = ev(), ev()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    def sws_case(key, sf, sw, sh, df, dw, dh, n):
        c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
        s_ = [torch.randint(0, 256, (n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(sf, sw, sh)]
        d_ = [torch.empty((n, r, cc), dtype=torch.uint8, device=dev) for r, cc in S.plane_shapes(df, dw, dh)]
        t = timed(lambda: c.scale_batch(s_, d_))
        byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
        out[key] = {"ms": round(t, 4), "hbm_frac": round(byt / (t * 1e-3) / 1e9 / 8000.0, 4)}
        c.close()

    sws_case("sws_nv12_4k_to_1080p", 23, 3840, 2160, 23, 1920, 1080, 64)
    sws_case("sws_yuv444p_1080p_to_4k", 5, 1920, 1080, 5, 3840, 2160, 32)
    sws_case("sws_nv12_1080p_to_4k_256", 23, 1920, 1080, 23, 3840, 2160, 256)
    planes, stride = 32, 3840
    nb = planes * 129600
    plane = torch.randint(0, 256, (planes * 2160, stride), dtype=torch.uint8, device=dev)
    by, bx = torch.meshgrid(torch.arange(planes * 270, device=dev), torch.arange(480, device=dev), indexing="ij")
    offs = (by * 8 * stride + bx * 8).to(torch.int32).reshape(-1).contiguous()
    coefs0 = torch.randint(-512, 512, (nb, 64), dtype=torch.int16, device=dev)
    coefs = coefs0.clone()
    t = timed(lambda: h264.idct_add_batch(h264.IDCT8, plane, stride, offs, coefs), prep=lambda: coefs.copy_(coefs0))
    out["h264_idct8_add"] = {"ms": round(t, 4), "hbm_frac": round(nb * 384 / (t * 1e-3) / 1e9 / 8000.0, 4)}
    del plane, coefs, coefs0, offs
    from ffmpeg_amd import hevc
    nsz, planes = 32, 16
    bw, bh = 3840 // nsz, 2160 // nsz
    ntu = planes * bw * bh
    tus = np.zeros(ntu, hevc.TU_DTYPE)
    idx = np.arange(ntu)
    pl, rem = idx // (bw * bh), idx % (bw * bh)
    tus["coeff_offset"] = idx * nsz * nsz
    tus["dst_offset"] = pl * 3840 * 2160 + (rem // bw) * nsz * 3840 + (rem % bw) * nsz
    tus["col_limit"] = nsz
    try:
        d_t = torch.from_numpy(tus.view(np.uint8).reshape(ntu, 12).copy()).to(dev)
        c0 = torch.randint(-512, 512, (ntu, nsz * nsz), dtype=torch.int16, device=dev)
        cc = c0.clone()
        pic = torch.randint(0, 256, (planes * 2160, 3840), dtype=torch.uint8, device=dev)
        t = timed(lambda: hevc.idct_batch(hevc.IDCT, 5, cc, pic, 3840, d_t, ntu), prep=lambda: cc.copy_(c0))
        out["hevc_idct32_add"] = {"ms": round(t, 4), "hbm_frac": round(ntu * nsz * nsz * 6 / (t * 1e-3) / 1e9 / 8000.0, 4)}
    except Exception as e:   # the python face of an older library may differ
        out["hevc_idct32_add"] = {"error": str(e)[:120]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
