"""__graft_entry__.smoke(): one tiny invocation of the hot path on cuda:0, checked against the oracle.
(The oracle is only the checker here; every pixel below is produced by libffhip.so's HIP kernels.)"""
import ctypes as C
import os
import sys

import numpy as np


def run():
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    import ffi
    from ffmpeg_amd import swscale as S, h264, _lib

    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    assert _lib.lib().ffhip_device_count() > 0, "libffhip.so sees no HIP device"
    torch.cuda.set_device(0)
    rng = np.random.default_rng(1)

    # 1. nv12 bicubic 2x upscale (the BASELINE configs[1] shape, shrunk)
    sw, sh, dw, dh = 192, 108, 384, 216
    src = ffi.alloc_frame(23, sw, sh, rng)
    ht = S.HostTables(sw, sh, 23, dw, dh, 23, S.SWS_BICUBIC)
    t = ffi.make_otables(sw, sh, 23, dw, dh, 23, S.SWS_BICUBIC, ht.banks(), ht.coeffs())
    want = ffi.alloc_frame(23, dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    ctx = S.SwsContext(sw, sh, 23, dw, dh, 23, S.SWS_BICUBIC)
    dsrc = [torch.from_numpy(a).cuda().unsqueeze(0) for a in src]
    ddst = [torch.zeros((1,) + a.shape, dtype=torch.uint8, device="cuda:0") for a in want]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for p, a in enumerate(want):
        assert np.array_equal(ddst[p][0].cpu().numpy(), a), "smoke: nv12 scale plane %d differs from the oracle" % p

    # 2. one 8x8 IDCT batch
    n, stride = 64, 64
    plane = rng.integers(0, 256, (64, stride), dtype=np.uint8)
    offs = (np.arange(8)[:, None] * 8 * stride + np.arange(8)[None, :] * 8).astype(np.int32).ravel()
    coefs = rng.integers(-1024, 1024, (n, 64)).astype(np.int16)
    wp, wc = plane.copy(), coefs.copy()
    for i in range(n):
        ffi.oracle().ffo_h264_idct8_add(C.cast(wp.ctypes.data + int(offs[i]), ffi.u8p), ffi.ptr(wc[i], ffi.i16p), stride)
    d_plane = torch.from_numpy(plane).cuda()
    d_c = torch.from_numpy(coefs).cuda()
    h264.idct_add_batch(h264.IDCT8, d_plane, stride, torch.from_numpy(offs).cuda(), d_c)
    torch.cuda.synchronize()
    assert np.array_equal(d_plane.cpu().numpy(), wp) and not d_c.cpu().numpy().any(), "smoke: idct8 differs"
    print("smoke ok: nv12 %dx%d->%dx%d bicubic and %d idct8 blocks bit-exact vs oracle on %s" %
          (sw, sh, dw, dh, n, torch.cuda.get_device_name(0)))
