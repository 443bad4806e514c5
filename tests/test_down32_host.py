"""CPU check of the host side of the exact-3:2 kernel (sws_down32.hip): the virtual banks (ffhip_sws_d32_virtual_bank_host — six
coefficients per output on the REGULAR window 3 (x >> 1) - 2 + (x & 1) .. + 5 of the edge-replicated row) reproduce the reference's
scaler when the kernel's schedule is emulated in numpy: replicate the rows' edges, hScale8To15_c on the regular windows, output row y
from the pairs P(q) = (row q, row q + 1) at q = 3 (y >> 1) - 2 + (y & 1), + 2, + 4 (clamped), yuv2planeX_8_c arithmetic."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import PIX
from ffmpeg_amd import _lib, swscale as S


def vbank(f, p, fs, n_dst, n_src):
    f = np.ascontiguousarray(f, np.int16); p = np.ascontiguousarray(p, np.int32)
    out = np.zeros(n_dst * 3, np.uint32)
    ok = _lib.lib().ffhip_sws_d32_virtual_bank_host(f.ctypes.data, p.ctypes.data, fs, n_dst, n_src, out.ctypes.data)
    return ok, out.view(np.int16).reshape(n_dst, 6).astype(np.int64)


def start(x):
    return 3 * (x >> 1) - 2 + (x & 1)


def hpass(plane, cv):
    h, w = plane.shape
    n = cv.shape[0]
    x = np.arange(n)
    acc = np.zeros((h, n), np.int64)
    for k in range(6):
        idx = np.clip(start(x) + k, 0, w - 1)
        acc += plane[:, idx].astype(np.int64) * cv[:, k][None, :]
    return np.clip(acc >> 7, -32768, 32767)          # the kernel packs with saturation; below -32768 is host-checked away


def vpass(hs, cv):
    h, n = hs.shape
    out = np.zeros((cv.shape[0], n), np.uint8)
    for y in range(cv.shape[0]):
        acc = np.full(n, 64 << 12, np.int64)
        for k in range(6):
            acc += hs[min(max(start(y) + k, 0), h - 1)] * cv[y, k]
        out[y] = np.clip(acc >> 19, 0, 255)
    return out


@pytest.mark.parametrize("flags", [ffi.SWS_BICUBIC, ffi.SWS_BILINEAR, ffi.SWS_POINT], ids=["bicubic", "bilinear", "point"])
@pytest.mark.parametrize("sw,sh", [(96, 36), (48, 24), (384, 54), (1560, 12)])
def test_virtual_banks_reproduce_the_scaler(sw, sh, flags):
    dw, dh = sw * 2 // 3, sh * 2 // 3
    ht = S.HostTables(sw, sh, PIX["yuv420p"], dw, dh, PIX["yuv420p"], flags)
    banks = ht.banks()
    t = ffi.make_otables(sw, sh, PIX["yuv420p"], dw, dh, PIX["yuv420p"], flags, banks, ht.coeffs())
    rng = np.random.default_rng(sw + flags)
    src = ffi.alloc_frame(PIX["yuv420p"], sw, sh, rng)
    src[0][::3] = np.where(rng.integers(0, 2, src[0][::3].shape) > 0, 255, 0)
    want = ffi.alloc_frame(PIX["yuv420p"], dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    for pl in range(3):
        hb, vb = ("hLum", "vLum") if pl == 0 else ("hChr", "vChr")
        w, h = src[pl].shape[1], src[pl].shape[0]
        views = []
        for name, nsrc in ((hb, w), (vb, h)):
            f, p, fs, n = banks[name]
            ok, cv = vbank(f, p, fs, n, nsrc)
            assert ok, (name, pl)
            views.append(cv)
        got = vpass(hpass(src[pl], views[0]), views[1])
        assert np.array_equal(got, want[pl][:, :got.shape[1]]), "plane %d: %d mismatches" % (pl, (got != want[pl][:, :got.shape[1]]).sum())


def test_banks_of_another_shape_are_refused():
    ht = S.HostTables(96, 36, PIX["yuv420p"], 48, 18, PIX["yuv420p"], ffi.SWS_BICUBIC)      # 2:1
    f, p, fs, n = ht.banks()["hLum"]
    assert not vbank(f, p, fs, n, 96)[0]
    ht = S.HostTables(96, 36, PIX["yuv420p"], 64, 24, PIX["yuv420p"], ffi.SWS_BICUBIC)
    f, p, fs, n = ht.banks()["hLum"]
    ok, _ = vbank(f, p + 1, fs, n, 96)                                                      # a bank shifted by a sample
    assert not ok
