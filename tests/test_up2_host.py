"""CPU check of the host side of the exact-2x kernel (sws_up2.hip): the virtual banks (ffhip_sws_up2_virtual_bank_host —
four coefficients per output on the REGULAR window of the edge-replicated row) reproduce the reference's scaler when the
kernel's schedule is emulated in numpy: replicate the rows' edges, hScale8To15_c on the regular windows, after source row r
emit output rows 2r-3 and 2r-2 from rows r-3..r (clamped), yuv2planeX_8_c arithmetic."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import PIX
from ffmpeg_amd import _lib, swscale as S


def vbank(f, p, n_dst, n_src):
    f = np.ascontiguousarray(f, np.int16); p = np.ascontiguousarray(p, np.int32)
    out = np.zeros(n_dst * 2, np.uint32)
    ok = _lib.lib().ffhip_sws_up2_virtual_bank_host(f.ctypes.data, p.ctypes.data, n_dst, n_src, out.ctypes.data)
    return ok, out.view(np.int16).reshape(n_dst, 4).astype(np.int64)


def hpass(plane, cv):
    """[H, W] u8 -> [H, 2W] 15-bit samples on the regular windows of the replicated rows"""
    h, w = plane.shape
    x = np.arange(2 * w)
    s0 = (x >> 1) - 2 + (x & 1)
    acc = np.zeros((h, 2 * w), np.int64)
    for k in range(4):
        idx = np.clip(s0 + k, 0, w - 1)
        acc += plane[:, idx].astype(np.int64) * cv[:, k][None, :]
    return np.minimum(acc >> 7, 32767)


def vpass(hs, cv):
    """[H, W2] samples -> [2H, W2] u8, the kernel's static schedule: step r emits rows 2r-3 and 2r-2 from rows r-3..r"""
    h, w2 = hs.shape
    out = np.zeros((2 * h, w2), np.uint8)
    for r in range(1, h + 2):
        rows = [min(max(r - 3 + k, 0), h - 1) for k in range(4)]
        for y in (2 * r - 3, 2 * r - 2):
            if 0 <= y < 2 * h:
                assert (y >> 1) - 2 + (y & 1) == r - 3
                acc = np.full(w2, 64 << 12, np.int64)
                for k in range(4):
                    acc += hs[rows[k]] * cv[y, k]
                out[y] = np.clip(acc >> 19, 0, 255)
    return out


@pytest.mark.parametrize("flags", [ffi.SWS_BICUBIC, ffi.SWS_BILINEAR, ffi.SWS_POINT], ids=["bicubic", "bilinear", "point"])
@pytest.mark.parametrize("sw,sh", [(64, 36), (16, 8), (200, 50)])
def test_virtual_banks_reproduce_the_scaler(sw, sh, flags):
    ht = S.HostTables(sw, sh, PIX["yuv420p"], 2 * sw, 2 * sh, PIX["yuv420p"], flags)
    banks = ht.banks()
    t = ffi.make_otables(sw, sh, PIX["yuv420p"], 2 * sw, 2 * sh, PIX["yuv420p"], flags, banks, ht.coeffs())
    rng = np.random.default_rng(sw + flags)
    src = ffi.alloc_frame(PIX["yuv420p"], sw, sh, rng)
    src[0][::3] = np.where(rng.integers(0, 2, src[0][::3].shape) > 0, 255, 0)   # rows that saturate the 15-bit intermediate
    want = ffi.alloc_frame(PIX["yuv420p"], 2 * sw, 2 * sh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == 2 * sh
    for pl in range(3):
        hb, vb = ("hLum", "vLum") if pl == 0 else ("hChr", "vChr")
        w, h = src[pl].shape[1], src[pl].shape[0]
        views = []
        for name, nsrc in ((hb, w), (vb, h)):
            f, p, fs, n = banks[name]
            f4 = np.zeros((n, 4), np.int16)
            f4[:, :fs] = np.asarray(f).reshape(n, fs)                # zero taps behind the real ones: same sums
            if fs == 1 and name == vb:
                f4[:, 0] = 4096                                     # yuv2plane1_8_c as a 1-tap yuv2planeX (sws_api.hip)
            p4 = np.asarray(p).copy()
            over = np.maximum(p4 + 4 - nsrc, 0)                     # keep the padded window inside the plane
            for i in np.nonzero(over)[0]:
                f4[i] = np.roll(f4[i], over[i]); p4[i] -= over[i]
            ok, cv = vbank(f4.reshape(-1), p4, n, nsrc)
            assert ok, name
            views.append(cv)
        got = vpass(hpass(src[pl], views[0]), views[1])
        assert np.array_equal(got, want[pl]), "plane %d: %d mismatches" % (pl, (got != want[pl]).sum())


def test_virtual_bank_rejects_irregular_banks():
    sw = 64
    ht = S.HostTables(sw, 36, PIX["yuv420p"], 2 * sw, 72, PIX["yuv420p"], ffi.SWS_BICUBIC)
    f, p, fs, n = ht.banks()["hLum"]
    assert vbank(f, p, n, sw)[0] == 1
    p2 = np.asarray(p).copy(); p2[40] += 2                          # a window off its regular place
    assert vbank(f, p2, n, sw)[0] == 0
    assert vbank(f, p, n, sw + 1)[0] == 0                           # not an exact 2x
    f3, p3, _, n3 = S.HostTables(sw, 36, PIX["yuv420p"], 3 * sw, 72, PIX["yuv420p"], ffi.SWS_BICUBIC).banks()["hLum"]
    assert vbank(f3, p3, n3, sw)[0] == 0
