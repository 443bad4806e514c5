"""CPU check of the host side of the exact-2x packed-RGB kernel (sws_up2rgb.hip) and of the equal-size one (sws_eqrgb.hip): the virtual
banks at ratio 2 and 4 (ffhip_sws_upn_virtual_bank_host) give, on the regular windows of the edge-replicated row, exactly the sums the
reference's banks give (initFilter()'s folded end taps included); the horizontal banks fold into the kernel's 32 scalar dwords
(ffhip_sws_up2rgb_hco_host: two coefficient rows + the three columns at either end); and the static schedule's window arithmetic — luma
row r completes output rows 2r-3, 2r-2, chroma row c completes 4c-6 .. 4c-3 — is what the banks' positions say."""
import numpy as np
import pytest

import ffi
from ffi import PIX
from ffmpeg_amd import _lib, swscale as S


def vbank(f, p, n_dst, n_src, ratio):
    f = np.ascontiguousarray(f, np.int16); p = np.ascontiguousarray(p, np.int32)
    out = np.zeros(n_dst * 2, np.uint32)
    ok = _lib.lib().ffhip_sws_upn_virtual_bank_host(f.ctypes.data, p.ctypes.data, n_dst, n_src, ratio, out.ctypes.data)
    return ok, out


def s0_of(x, ratio):
    return (x >> 1) - 2 + (x & 1) if ratio == 2 else ((x + 2) >> 2) - 2


def pad4(f, p, fs, n, n_src):
    """a bank of <= 4 taps as 4 taps, windows kept inside the row (what build_fast_view() hands the kernels)"""
    f4 = np.zeros((n, 4), np.int16)
    f = np.asarray(f).reshape(n, fs)
    p4 = np.asarray(p).copy()
    for x in range(n):
        q = min(p4[x], n_src - 4)
        f4[x, p4[x] - q: p4[x] - q + fs] = f[x]
        p4[x] = q
    return f4, p4


@pytest.mark.parametrize("flags", [ffi.SWS_BICUBIC, ffi.SWS_BILINEAR], ids=["bicubic", "bilinear"])
@pytest.mark.parametrize("sw,sh", [(16, 8), (64, 36), (200, 50), (1920, 1080)])
def test_virtual_banks_of_an_rgb_target(sw, sh, flags):
    ht = S.HostTables(sw, sh, PIX["yuv420p"], 2 * sw, 2 * sh, PIX["rgb24"], flags)
    banks = ht.banks()
    rng = np.random.default_rng(sw * 3 + flags)
    views = {}
    for name, n_src, ratio in (("hLum", sw, 2), ("hChr", sw // 2, 2), ("vLum", sh, 2), ("vChr", sh // 2, 4)):
        f, p, fs, n = banks[name]
        assert n == ratio * n_src and fs <= 4
        f4, p4 = pad4(f, p, fs, n, n_src)
        ok, vb = vbank(f4, p4, n, n_src, ratio)
        assert ok, name
        cv = vb.view(np.int16).reshape(n, 4).astype(np.int64)
        views[name] = vb
        # the same sums on any row: taps on clamp(s0 + k) against the bank's own window
        row = rng.integers(0, 32768, n_src).astype(np.int64)
        x = np.arange(n)
        s0 = np.array([s0_of(int(i), ratio) for i in x])
        got = sum(cv[:, k] * row[np.clip(s0 + k, 0, n_src - 1)] for k in range(4))
        want = sum(f4[:, k].astype(np.int64) * row[p4 + k] for k in range(4))
        assert np.array_equal(got, want), name
        # the schedule: source row r is the LAST row of output rows 2r-3, 2r-2 (ratio 2) / 4r-6 .. 4r-3 (ratio 4)
        for r in (3, n_src // 2, n_src - 1):
            ys = (2 * r - 3, 2 * r - 2) if ratio == 2 else tuple(range(4 * r - 6, 4 * r - 2))
            assert all(s0_of(y, ratio) + 3 == r for y in ys if 0 <= y < n)
    out = np.zeros(32, np.uint32)
    hl, hc = views["hLum"], views["hChr"]
    assert _lib.lib().ffhip_sws_up2rgb_hco_host(hl.ctypes.data, hl.size // 2, hc.ctypes.data, hc.size // 2, out.ctypes.data) == 1
    for b, v in enumerate((hl, hc)):
        v2 = v.reshape(-1, 2)
        n = v2.shape[0]
        assert np.array_equal(out[4 * b: 4 * b + 2], v2[4]) and np.array_equal(out[4 * b + 2: 4 * b + 4], v2[5])
        for x in range(3, n - 3):                       # between the ends the bank is two coefficient rows
            assert np.array_equal(v2[x], out[4 * b + 2 * (x & 1): 4 * b + 2 * (x & 1) + 2]), (b, x)
        assert np.array_equal(out[8 + 12 * b: 14 + 12 * b], v2[:3].reshape(-1))
        assert np.array_equal(out[14 + 12 * b: 20 + 12 * b], v2[n - 3:].reshape(-1))


def test_a_bank_that_does_not_repeat_is_refused():
    ht = S.HostTables(64, 36, PIX["yuv420p"], 128, 72, PIX["rgb24"], ffi.SWS_BICUBIC)
    f, p, fs, n = ht.bank("hLum")
    f4, p4 = pad4(f, p, fs, n, 64)
    f4[40, 1] += 1
    f4[40, 2] -= 1
    ok, hl = vbank(f4, p4, n, 64, 2)
    assert ok
    f, p, fs, n2 = ht.bank("hChr")
    ok, hc = vbank(*pad4(f, p, fs, n2, 32), n2, 32, 2)
    assert ok
    out = np.zeros(32, np.uint32)
    assert _lib.lib().ffhip_sws_up2rgb_hco_host(hl.ctypes.data, n, hc.ctypes.data, n2, out.ctypes.data) == 0


def test_a_tap_off_its_regular_window_is_refused():
    ht = S.HostTables(64, 36, PIX["yuv420p"], 128, 72, PIX["rgb24"], ffi.SWS_BICUBIC)
    f, p, fs, n = ht.bank("vChr")
    f4, p4 = pad4(f, p, fs, n, 18)
    p4[30] += 1          # the window of output row 30 one chroma row down: its last tap leaves the regular window
    ok, _ = vbank(f4, p4, n, 18, 4)
    assert not ok
    ok, _ = vbank(f4, p4, n, 17, 4)   # not a 4x bank
    assert not ok
