"""oracle == reference above 8 bits: hevcdsp at 10 and 12 bits (libavcodec/hevc/dsp.c:133-196 instantiates dsp_template.c /
h26x templates per BIT_DEPTH; pixels are uint16_t, strides stay in bytes).  The reference's HEVCDSPContext is re-initialised at
the depth under test through ffref_hevc_set_bit_depth(); the oracle's *_bd functions take the depth as an argument."""
import ctypes as C
import os

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p, i16p, i32p
from test_oracle_vs_ref import HEVC_WIDTHS, hevc_coeffs, hevc_restore_case, hevc_weight_case

pytestmark = pytest.mark.skipif(not os.path.exists(ffi.REF_SO), reason="oracle/_ref not built")
DEPTHS = [10, 12]


@pytest.fixture
def depth(request):
    R = ffi.ref()
    R.ffref_hevc_set_bit_depth(request.param)
    yield request.param
    R.ffref_hevc_set_bit_depth(8)


def pix(rng, shape, bd, extremes=False):
    a = rng.integers(0, 1 << bd, shape).astype(np.uint16)
    if extremes:
        a[: shape[0] // 2] = rng.choice(np.array([0, (1 << bd) - 1], np.uint16), (shape[0] // 2, shape[1]))
    return a


def at(a, row, col):
    return C.cast(a.ctypes.data + (row * a.shape[1] + col) * a.itemsize, u8p)


@pytest.mark.parametrize("depth", DEPTHS, indirect=True)
def test_hevc_transforms_hbd(depth):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(700 + depth)
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        for col_limit in list(range(0, 2 * n + 6, 3)) + [1000]:
            for kind in range(4):
                c = hevc_coeffs(rng, n, kind)
                a, b = c.copy(), c.copy()
                R.ffref_hevc_idct(lg - 2, ptr(a, i16p), col_limit)
                O.ffo_hevc_idct_bd(depth, lg, ptr(b, i16p), col_limit)
                assert np.array_equal(a, b), (n, col_limit, kind)
        for rep in range(12):
            c = hevc_coeffs(rng, n, rep % 4)
            a, b = c.copy(), c.copy()
            R.ffref_hevc_idct_dc(lg - 2, ptr(a, i16p))
            O.ffo_hevc_idct_dc_bd(depth, lg, ptr(b, i16p))
            assert np.array_equal(a, b)
            res = hevc_coeffs(rng, n, rep % 4)
            d0 = pix(rng, (n + 2, 48), depth, rep % 2 == 0)
            a, b = d0.copy(), d0.copy()
            R.ffref_hevc_add_residual(lg - 2, at(a, 1, 3), ptr(res, i16p), 96)
            O.ffo_hevc_add_residual_bd(depth, lg, at(b, 1, 3), ptr(res, i16p), 96)
            assert np.array_equal(a, b)
            c = hevc_coeffs(rng, n, rep % 4).ravel().copy()
            a, b = c.copy(), c.copy()
            R.ffref_hevc_dequant(ptr(a, i16p), lg)
            O.ffo_hevc_dequant_bd(depth, ptr(b, i16p), lg)
            assert np.array_equal(a, b), ("dequant", lg)
    for rep in range(100):
        c = hevc_coeffs(rng, 4, rep % 4)
        a, b = c.copy(), c.copy()
        R.ffref_hevc_transform_4x4_luma(ptr(a, i16p))
        O.ffo_hevc_transform_4x4_luma_bd(depth, ptr(b, i16p))
        assert np.array_equal(a, b)


def lf_case(rng, smooth, bd):
    sc = 1 << (bd - 8)
    if smooth:
        base = int(rng.integers(20, 230)) * sc
        buf = np.clip(base + rng.integers(-3 * sc, 3 * sc + 1, (16, 16)) + np.where(np.arange(16)[None, :] >= 8, int(rng.integers(-12, 13)) * sc, 0),
                      0, (1 << bd) - 1)
        if rng.random() < .5:
            buf = buf.T
    else:
        buf = rng.integers(0, 1 << bd, (16, 16))
    beta = int(rng.integers(0, 65))
    tc = rng.integers(0, 25, 2).astype(np.int32)
    no_p, no_q = rng.integers(0, 2, 2) * (rng.random() < .3), rng.integers(0, 2, 2) * (rng.random() < .3)
    return np.ascontiguousarray(buf.astype(np.uint16)), beta, tc, no_p.astype(np.uint8), no_q.astype(np.uint8)


@pytest.mark.parametrize("depth", DEPTHS, indirect=True)
def test_hevc_loop_filter_hbd(depth):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(750 + depth)
    changed = 0
    for rep in range(2000):
        buf, beta, tc, no_p, no_q = lf_case(rng, rep % 4 != 0, depth)
        which = rep % 4
        chroma, vertical = (which >> 1) & 1, which & 1
        a, b = buf.copy(), buf.copy()
        r, c = (4, 8) if vertical else (8, 4)
        R.ffref_hevc_loop_filter(which, at(a, r, c), 32, beta, ptr(tc, i32p), ptr(no_p), ptr(no_q))
        O.ffo_hevc_loop_filter_bd(depth, chroma, vertical, at(b, r, c), 32, beta, ptr(tc, i32p), ptr(no_p), ptr(no_q))
        assert np.array_equal(a, b), (rep, which, beta, tc, no_p, no_q)
        changed += int((a != buf).any())
    assert changed > 500


@pytest.mark.parametrize("depth", DEPTHS, indirect=True)
def test_hevc_sao_hbd(depth):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(760 + depth)
    sc = 1 << (depth - 8)
    for rep in range(120):
        w, h = int(rng.choice([8, 16, 24, 32, 48, 64])), int(rng.choice([4, 8, 16, 33, 64]))
        idx = [0, 1, 2, 2, 3, 3, 4, 4][((w + 7) >> 3) - 1]
        off = (rng.integers(-31, 32, 5) * sc).astype(np.int16)
        off[0] = 0
        # the reference's edge filter reads a padded copy with a fixed stride of 192 BYTES (96 samples here)
        src = pix(rng, (h + 2, 96), depth) if rep % 3 else (np.clip(rng.integers(-2, 3, (h + 2, 96)) + 500 * sc // 4, 0, (1 << depth) - 1)).astype(np.uint16)
        d0 = pix(rng, (h, 80), depth)
        a, b = d0.copy(), d0.copy()
        lc = int(rng.integers(0, 32))
        R.ffref_hevc_sao_band(idx, ptr(a), at(src, 1, 1), 160, 192, ptr(off, i16p), lc, w, h)
        O.ffo_hevc_sao_band_bd(depth, ptr(b), at(src, 1, 1), 160, 192, ptr(off, i16p), lc, w, h)
        assert np.array_equal(a, b), ("band", rep)
        for eo in range(4):
            a, b = d0.copy(), d0.copy()
            R.ffref_hevc_sao_edge(idx, ptr(a), at(src, 1, 1), 160, ptr(off, i16p), eo, w, h)
            O.ffo_hevc_sao_edge_bd(depth, ptr(b), at(src, 1, 1), 160, 192, ptr(off, i16p), eo, w, h)
            assert np.array_equal(a, b), ("edge", rep, eo)
    for rep in range(300):
        var, eo, off0, borders, w, h, ve, he, de = hevc_restore_case(rng, rep)
        src = pix(rng, (h, 80), depth)
        d0 = pix(rng, (h, 72), depth)
        a, b = d0.copy(), d0.copy()
        R.ffref_hevc_sao_edge_restore(var, ptr(a), ptr(src), 144, 160, eo, off0 * sc, ptr(borders, i32p), w, h, ptr(ve), ptr(he), ptr(de))
        O.ffo_hevc_sao_edge_restore_bd(depth, var, ptr(b), ptr(src), 144, 160, eo, off0 * sc, ptr(borders, i32p), w, h, ptr(ve), ptr(he), ptr(de))
        assert np.array_equal(a, b), ("restore", rep)


@pytest.mark.parametrize("depth", DEPTHS, indirect=True)
def test_hevc_mc_hbd(depth):
    """put_hevc_{qpel,epel}{,_uni,_uni_w,_bi,_bi_w}: every fractional position x the 10 width classes, 16-bit samples"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(780 + depth)
    src = pix(rng, (80, 96), depth, extremes=True)
    rep = 0
    for chroma in (0, 1):
        nfrac = 8 if chroma else 4
        for w in HEVC_WIDTHS:
            for mx in range(nfrac):
                for my in range(nfrac):
                    h = int(rng.choice([2, 4, 8, 16, 64])) if w > 2 else 2
                    y0, x0 = int(rng.integers(4, 80 - h - 5)), int(rng.integers(4, 96 - w - 5))
                    sp = at(src, y0, x0)
                    a16, b16 = np.zeros((64, 64), np.int16), np.zeros((64, 64), np.int16)
                    R.ffref_hevc_mc(chroma, 0, a16.ctypes.data, 0, sp, 192, h, mx, my, w)
                    O.ffo_hevc_mc_bd(depth, chroma, 0, b16.ctypes.data, 0, sp, 192, h, mx, my, w)
                    assert np.array_equal(a16, b16), (chroma, w, mx, my)
                    a, b = np.full((64, 80), 7, np.uint16), np.full((64, 80), 7, np.uint16)
                    R.ffref_hevc_mc(chroma, 1, a.ctypes.data, 160, sp, 192, h, mx, my, w)
                    O.ffo_hevc_mc_bd(depth, chroma, 1, b.ctypes.data, 160, sp, 192, h, mx, my, w)
                    assert np.array_equal(a, b), (chroma, w, mx, my, "uni")
                    src2 = rng.integers(-8192, 16384, (64, 64)).astype(np.int16) if rep % 4 else np.full((64, 64), 16383 if rep % 8 else -8192, np.int16)
                    for mode in (2, 3, 4):
                        rep += 1
                        d, wx0, wx1, ox = hevc_weight_case(rng, rep)
                        a, b = np.full((64, 80), 7, np.uint16), np.full((64, 80), 7, np.uint16)
                        R.ffref_hevc_mc_w(chroma, mode, ptr(a), 160, sp, 192, ptr(src2, i16p), h, d, wx0, wx1, ox, mx, my, w)
                        O.ffo_hevc_mc_w_bd(depth, chroma, mode, ptr(b), 160, sp, 192, ptr(src2, i16p), h, d, wx0, wx1, ox, mx, my, w)
                        assert np.array_equal(a, b), (chroma, mode, w, mx, my, d, wx0, wx1, ox)


# ---------------------------------------------------------------------------------------------------------------------------
# vp9dsp above 8 bits (ff_vp9dsp_init(dsp, 10 / 12, bitexact): vp9dsp_10bpp.c / vp9dsp_12bpp.c instantiate vp9dsp_template.c;
# pixels uint16_t, itxfm_add's block holds int32 coefficients, dctint = int64_t)
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture
def vdepth(request):
    R = ffi.ref()
    R.ffref_vp9_set_bit_depth(request.param)
    yield request.param
    R.ffref_vp9_set_bit_depth(8)


def vp9_block32(rng, n, kind, bd):
    """int32 coefficient blocks: the range a bd-bit stream can produce is about 2^(bd + 8); sparse, dense, dc-only, extreme"""
    lim = 1 << (bd + 7)
    blk = np.zeros(n * n, np.int32)
    if kind == 0:
        blk[:] = rng.integers(-lim // 32, lim // 32 + 1, n * n)
    elif kind == 1:
        blk[:] = rng.integers(-lim // 64, lim // 64 + 1, n * n) * (rng.random(n * n) < .2)
    elif kind == 2:
        blk[0] = rng.integers(-lim // 8, lim // 8 + 1)
    elif kind == 3:
        blk[:] = rng.integers(-lim // 4, lim // 4 + 1, n * n) * (rng.random(n * n) < .05)
    else:
        blk[:] = rng.integers(-lim, lim, n * n)
    return blk


@pytest.mark.parametrize("vdepth", DEPTHS, indirect=True)
def test_vp9_itxfm_add_hbd(vdepth):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(950 + vdepth)
    for tx in range(5):
        n = 4 if tx == 4 else 4 << tx
        for txtp in range(4):
            for rep in range(30):
                kind = rep % 5
                blk = vp9_block32(rng, n, kind, vdepth)
                eob = 1 if kind == 2 else int(rng.integers(2, n * n + 1))
                dst0 = pix(rng, (n, n + 5), vdepth, rep % 2 == 0)
                a, b, ba, bb = dst0.copy(), dst0.copy(), blk.copy(), blk.copy()
                R.ffref_vp9_itxfm_add(tx, txtp, ptr(a), 2 * (n + 5), C.cast(ba.ctypes.data, i16p), eob)
                O.ffo_vp9_itxfm_add_bd(vdepth, tx, txtp, ptr(b), 2 * (n + 5), ptr(bb, i32p), eob)
                assert np.array_equal(a, b) and np.array_equal(ba, bb), (tx, txtp, rep)


@pytest.mark.parametrize("vdepth", DEPTHS, indirect=True)
def test_vp9_mc_hbd(vdepth):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(960 + vdepth)
    src = pix(rng, (150, 170), vdepth, extremes=False)
    src[:40] = rng.choice(np.array([0, (1 << vdepth) - 1], np.uint16), (40, 170))
    for rep in range(1500):
        f, avg = int(rng.integers(0, 4)), rep & 1
        w = int(rng.choice([4, 8, 16, 32, 64])); h = int(rng.choice([1, 2, 4, 8, 16, 33, 64]))
        mx, my = (int(v) for v in rng.integers(0, 16, 2))
        if rep % 5 == 0:
            mx = 0
        if rep % 7 == 0:
            my = 0
        y0, x0 = int(rng.integers(4, 150 - h - 5)), int(rng.integers(4, 170 - w - 5))
        sp = at(src, y0, x0)
        d0 = pix(rng, (64, 72), vdepth)
        a, b = d0.copy(), d0.copy()
        R.ffref_vp9_mc(f, avg, ptr(a), 144, sp, 340, w, h, mx, my)
        O.ffo_vp9_mc_bd(vdepth, f, avg, ptr(b), 144, sp, 340, w, h, mx, my)
        assert np.array_equal(a, b), (f, avg, w, h, mx, my)
    for rep in range(300):                               # scaled: steps 1..32 sixteenths (16x up to 2x down)
        f, avg = int(rng.integers(0, 4)), rep & 1
        w = int(rng.choice([4, 8, 16, 32, 64])); h = int(rng.choice([2, 4, 8, 16, 33, 64]))
        mx, my = (int(v) for v in rng.integers(0, 16, 2))
        dx, dy = (int(v) for v in rng.integers(1, 33, 2))
        sp = at(src, 4, 4)
        d0 = pix(rng, (64, 72), vdepth)
        a, b = d0.copy(), d0.copy()
        R.ffref_vp9_smc(f, avg, ptr(a), 144, sp, 340, w, h, mx, my, dx, dy)
        O.ffo_vp9_smc_bd(vdepth, f, avg, ptr(b), 144, sp, 340, w, h, mx, my, dx, dy)
        assert np.array_equal(a, b), ("scaled", f, avg, w, h, mx, my, dx, dy)


def vp9_lf_plane16(rng, bd, n=48):
    sc = 1 << (bd - 8)
    kind = int(rng.integers(0, 5))
    a, b = int(rng.integers(0, 256)) * sc, int(rng.integers(0, 256)) * sc
    if kind < 3:
        b = int(np.clip(a + rng.integers(-6 * sc, 6 * sc + 1), 0, (1 << bd) - 1))
    p = np.empty((n, n), np.int64)
    p[:, :n // 2] = a
    p[:, n // 2:] = b
    p = p.T.copy() if rng.random() < .5 else p
    noise = [0, 1, sc, 3 * sc, 40 * sc][kind]
    p = p + rng.integers(-noise, noise + 1, (n, n))
    return np.clip(p, 0, (1 << bd) - 1).astype(np.uint16)


@pytest.mark.parametrize("vdepth", DEPTHS, indirect=True)
def test_vp9_loop_filter_hbd(vdepth):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(970 + vdepth)
    WD = [4, 8, 16]
    for rep in range(2400):
        pl = vp9_lf_plane16(rng, vdepth)
        E, I, H = int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 16))
        if rep % 3 == 0:
            E, I = 255, 63
        a, b = pl.copy(), pl.copy()
        dirn = rep & 1
        which = rep % 3
        seg2 = 8 * (48 if not dirn else 1)
        if which == 0:
            w = int(rng.integers(0, 3))
            R.ffref_vp9_loop_filter(0, w, 0, dirn, at(a, 24, 24), 96, E, I, H)
            O.ffo_vp9_loop_filter_bd(vdepth, WD[w], dirn, at(b, 24, 24), 96, E, I, H)
        elif which == 1:
            R.ffref_vp9_loop_filter(1, 0, 0, dirn, at(a, 24, 24), 96, E, I, H)
            O.ffo_vp9_loop_filter_bd(vdepth, 16, dirn, at(b, 24, 24), 96, E, I, H)
            O.ffo_vp9_loop_filter_bd(vdepth, 16, dirn, C.cast(b.ctypes.data + 2 * (24 * 48 + 24 + seg2), u8p), 96, E, I, H)
        else:
            w1, w2 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            E2, I2, H2 = int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 16))
            R.ffref_vp9_loop_filter(2, w1, w2, dirn, at(a, 24, 24), 96, E | E2 << 8, I | I2 << 8, H | H2 << 8)
            O.ffo_vp9_loop_filter_bd(vdepth, WD[w1], dirn, at(b, 24, 24), 96, E, I, H)
            O.ffo_vp9_loop_filter_bd(vdepth, WD[w2], dirn, C.cast(b.ctypes.data + 2 * (24 * 48 + 24 + seg2), u8p), 96, E2, I2, H2)
        assert np.array_equal(a, b), (rep, which, dirn, E, I, H)


@pytest.mark.parametrize("vdepth", DEPTHS, indirect=True)
def test_vp9_intra_pred_hbd(vdepth):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(980 + vdepth)
    mx = (1 << vdepth) - 1
    for tx in range(4):
        n = 4 << tx
        for mode in range(15):
            for rep in range(8):
                left = rng.integers(0, mx + 1, n + 16).astype(np.uint16)
                topbuf = rng.integers(0, mx + 1, 16 + 2 * n + 32).astype(np.uint16)
                if rep % 3 == 0:
                    left[:] = rng.choice(np.array([0, mx], np.uint16), left.size)
                    topbuf[:] = rng.choice(np.array([0, mx], np.uint16), topbuf.size)
                tp = C.cast(topbuf.ctypes.data + 32, u8p)
                a = pix(rng, (n, n + 3), vdepth)
                b = a.copy()
                R.ffref_vp9_intra_pred(tx, mode, ptr(a), 2 * (n + 3), ptr(left), tp)
                O.ffo_vp9_intra_pred_bd(vdepth, tx, mode, ptr(b), 2 * (n + 3), ptr(left), tp)
                assert np.array_equal(a, b), (tx, mode, rep)
