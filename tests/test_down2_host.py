"""CPU check of the host side of the exact-2:1 kernel (sws_down2.hip): the virtual banks (ffhip_sws_down2_virtual_bank_host —
eight coefficients per output on the REGULAR window 2x - 3 .. 2x + 4 of the edge-replicated row) reproduce the reference's
scaler when the kernel's schedule is emulated in numpy: replicate the rows' edges, hScale8To15_c on the regular windows, output
row y from the row pairs T(y-1) .. T(y+2), T(t) = (row 2t-1, row 2t) clamped, yuv2planeX_8_c arithmetic."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import PIX
from ffmpeg_amd import _lib, swscale as S


def vbank(f, p, fs, n_dst, n_src):
    f = np.ascontiguousarray(f, np.int16); p = np.ascontiguousarray(p, np.int32)
    out = np.zeros(n_dst * 4, np.uint32)
    ok = _lib.lib().ffhip_sws_down2_virtual_bank_host(f.ctypes.data, p.ctypes.data, fs, n_dst, n_src, out.ctypes.data)
    return ok, out.view(np.int16).reshape(n_dst, 8).astype(np.int64)


def hpass(plane, cv):
    h, w = plane.shape
    x = np.arange(w // 2)
    acc = np.zeros((h, w // 2), np.int64)
    for k in range(8):
        idx = np.clip(2 * x - 3 + k, 0, w - 1)
        acc += plane[:, idx].astype(np.int64) * cv[:, k][None, :]
    return np.clip(acc >> 7, -32768, 32767)          # the kernel packs with saturation; below -32768 is host-checked away


def vpass(hs, cv):
    h, w2 = hs.shape
    out = np.zeros((h // 2, w2), np.uint8)
    for y in range(h // 2):
        acc = np.full(w2, 64 << 12, np.int64)
        for k in range(8):
            acc += hs[min(max(2 * y - 3 + k, 0), h - 1)] * cv[y, k]
        out[y] = np.clip(acc >> 19, 0, 255)
    return out


@pytest.mark.parametrize("flags", [ffi.SWS_BICUBIC, ffi.SWS_BILINEAR], ids=["bicubic", "bilinear"])
@pytest.mark.parametrize("sw,sh", [(64, 36), (48, 16), (200, 52), (1032, 8)])
def test_virtual_banks_reproduce_the_scaler(sw, sh, flags):
    ht = S.HostTables(sw, sh, PIX["yuv420p"], sw // 2, sh // 2, PIX["yuv420p"], flags)
    banks = ht.banks()
    t = ffi.make_otables(sw, sh, PIX["yuv420p"], sw // 2, sh // 2, PIX["yuv420p"], flags, banks, ht.coeffs())
    rng = np.random.default_rng(sw + flags)
    src = ffi.alloc_frame(PIX["yuv420p"], sw, sh, rng)
    src[0][::3] = np.where(rng.integers(0, 2, src[0][::3].shape) > 0, 255, 0)
    want = ffi.alloc_frame(PIX["yuv420p"], sw // 2, sh // 2)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == sh // 2
    for pl in range(3):
        hb, vb = ("hLum", "vLum") if pl == 0 else ("hChr", "vChr")
        w, h = src[pl].shape[1], src[pl].shape[0]
        views = []
        for name, nsrc in ((hb, w), (vb, h)):
            f, p, fs, n = banks[name]
            ok, cv = vbank(f, p, fs, n, nsrc)
            assert ok, (name, fs, np.asarray(p)[:4])
            views.append(cv)
        got = vpass(hpass(src[pl], views[0]), views[1])
        assert np.array_equal(got, want[pl]), "plane %d: %d mismatches" % (pl, (got != want[pl]).sum())


def test_virtual_bank_rejects_irregular_banks():
    sw = 128
    ht = S.HostTables(sw, 36, PIX["yuv420p"], sw // 2, 18, PIX["yuv420p"], ffi.SWS_BICUBIC)
    f, p, fs, n = ht.banks()["hLum"]
    assert vbank(f, p, fs, n, sw)[0] == 1
    p2 = np.asarray(p).copy(); p2[20] += 2                          # a window off its regular place
    assert vbank(f, p2, fs, n, sw)[0] == 0
    assert vbank(f, p, fs, n, sw + 2)[0] == 0                       # not an exact 2:1
    f4, p4, fs4, n4 = S.HostTables(sw, 36, PIX["yuv420p"], sw // 4, 18, PIX["yuv420p"], ffi.SWS_BICUBIC).banks()["hLum"]
    assert vbank(f4, p4, fs4, n4, sw // 2)[0] == 0                  # a 16-tap bank does not fit the 8-sample windows
