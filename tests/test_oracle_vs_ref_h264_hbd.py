"""oracle == reference for the H.264 tables at every depth the reference instantiates (8 / 9 / 10 / 12 / 14 bits:
libavcodec/h264dsp.c:135-147, h264qpel.c:87-103, h264chroma.c:38-52) and for the 4:2:2 / MBAFF members.  The reference's
H264DSPContext / H264QpelContext / H264ChromaContext are re-initialised at the depth under test (ffref_h264_set_bit_depth, as the
decoder does per SPS); the oracle's *_bd functions (oracle/ffo_h264_hbd.c) take the depth as an argument.  Shapes follow
tests/checkasm/h264dsp.c:175-440 and h264qpel.c:51-82: random blocks scaled into the depth's range, extreme samples, every tc0 /
alpha / beta ladder step, all 16 quarter-pel positions x 3 sizes x put / avg."""
import ctypes as C
import os

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p, i16p

pytestmark = pytest.mark.skipif(not os.path.exists(ffi.REF_SO), reason="oracle/_ref not built")
DEPTHS = [8, 9, 10, 12, 14]
SCAN8 = [4 + 1 * 8, 5 + 1 * 8, 4 + 2 * 8, 5 + 2 * 8, 6 + 1 * 8, 7 + 1 * 8, 6 + 2 * 8, 7 + 2 * 8, 4 + 3 * 8, 5 + 3 * 8, 4 + 4 * 8, 5 + 4 * 8,
         6 + 3 * 8, 7 + 3 * 8, 6 + 4 * 8, 7 + 4 * 8, 4 + 6 * 8, 5 + 6 * 8, 4 + 7 * 8, 5 + 7 * 8, 6 + 6 * 8, 7 + 6 * 8, 6 + 7 * 8, 7 + 7 * 8,
         4 + 8 * 8, 5 + 8 * 8, 4 + 9 * 8, 5 + 9 * 8, 6 + 8 * 8, 7 + 8 * 8, 6 + 9 * 8, 7 + 9 * 8, 4 + 11 * 8, 5 + 11 * 8, 4 + 12 * 8, 5 + 12 * 8,
         6 + 11 * 8, 7 + 11 * 8, 6 + 12 * 8, 7 + 12 * 8, 4 + 13 * 8, 5 + 13 * 8, 4 + 14 * 8, 5 + 14 * 8, 6 + 13 * 8, 7 + 13 * 8, 6 + 14 * 8, 7 + 14 * 8]


@pytest.fixture
def depth(request):
    R = ffi.ref()
    R.ffref_h264_set_bit_depth.argtypes = [C.c_int]
    R.ffref_h264_set_bit_depth(request.param)
    yield request.param
    R.ffref_h264_set_bit_depth(8)


def pixels(rng, shape, bd, extremes=False):
    a = rng.integers(0, 1 << bd, shape).astype(np.uint16 if bd > 8 else np.uint8)
    if extremes:
        a[::2] = rng.choice(np.array([0, (1 << bd) - 1], a.dtype), a[::2].shape)
    return a


def coefs(rng, n, bd, big=False):
    lim = (1 << (bd + 7)) if big else (1 << (bd + 2))
    return rng.integers(-lim, lim, n).astype(np.int32 if bd > 8 else np.int16)


def at(a, row, col):
    return C.cast(a.ctypes.data + (row * a.shape[1] + col) * a.itemsize, u8p)


def bptr(b):
    return C.cast(b.ctypes.data, i16p)


def _sigs(R, O):
    R.ffref_h264_idct.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
    R.ffref_h264_add_pixels_clear.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
    R.ffref_h264_idct_multi.argtypes = [C.c_int, u8p, C.POINTER(C.c_int), i16p, C.c_ssize_t, u8p]
    R.ffref_h264_idct_add8.argtypes = [C.POINTER(u8p), C.POINTER(C.c_int), i16p, C.c_ssize_t, u8p, C.c_int]
    R.ffref_h264_luma_dc_dequant_idct.argtypes = [i16p, i16p, C.c_int]
    R.ffref_h264_chroma_dc_dequant_idct.argtypes = [i16p, C.c_int]
    R.ffref_h264_chroma_dc_dequant_idct_422.argtypes = [i16p, C.c_int]
    R.ffref_h264_loop_filter_variant.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.POINTER(C.c_int8)]
    R.ffref_h264_qpel.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t]
    R.ffref_h264_chroma.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
    R.ffref_h264_weight.argtypes = [C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
    R.ffref_h264_biweight.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    O.ffo_h264_idct_bd.argtypes = [C.c_int, C.c_int, u8p, i16p, C.c_ssize_t]
    O.ffo_h264_idct_mb_bd.argtypes = [C.c_int, C.c_int, u8p, C.POINTER(C.c_int), i16p, C.c_ssize_t, u8p]
    O.ffo_h264_idct_add8_bd.argtypes = [C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(C.c_int), i16p, C.c_ssize_t, u8p]
    O.ffo_h264_luma_dc_dequant_bd.argtypes = [C.c_int, i16p, i16p, C.c_int]
    O.ffo_h264_chroma_dc_dequant_bd.argtypes = [C.c_int, C.c_int, i16p, C.c_int]
    O.ffo_h264_loop_filter_bd.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.POINTER(C.c_int8)]
    O.ffo_h264_qpel_bd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t]
    O.ffo_h264_chroma_mc_bd.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
    O.ffo_h264_weight_bd.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
    O.ffo_h264_biweight_bd.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]


@pytest.mark.parametrize("depth", DEPTHS, indirect=True)
def test_h264_idct_bd(depth):
    R, O = ffi.ref(), ffi.oracle()
    _sigs(R, O)
    rng = np.random.default_rng(900 + depth)
    for kind in range(6):
        n = 4 if kind in (0, 2, 4) else 8
        for rep in range(40):
            c = coefs(rng, n * n, depth, big=rep % 5 == 4)
            if kind in (2, 3):
                c[1:] = 0
            d0 = pixels(rng, (n + 4, 40), depth, rep % 3 == 0)
            a, b, ca, cb = d0.copy(), d0.copy(), c.copy(), c.copy()
            stride = d0.strides[0]
            if kind < 4:
                R.ffref_h264_idct(kind, at(a, 2, 5), bptr(ca), stride)
            else:
                R.ffref_h264_add_pixels_clear(n, at(a, 2, 5), bptr(ca), stride)
            O.ffo_h264_idct_bd(depth, kind, at(b, 2, 5), bptr(cb), stride)
            assert np.array_equal(a, b) and np.array_equal(ca, cb), (kind, rep)


@pytest.mark.parametrize("depth", DEPTHS, indirect=True)
def test_h264_idct_dispatchers_and_dc_bd(depth):
    R, O = ffi.ref(), ffi.oracle()
    _sigs(R, O)
    rng = np.random.default_rng(950 + depth)
    px = 2 if depth > 8 else 1
    for rep in range(30):
        stride_px = 48
        d0 = pixels(rng, (40, stride_px), depth, rep % 4 == 0)
        stride = d0.strides[0]
        bo = np.zeros(48, np.int32)
        for i in range(16):
            bx, by = (i & 1) + 2 * ((i >> 2) & 1), ((i >> 1) & 1) + 2 * (i >> 3)
            bo[i] = (4 * by * stride_px + 4 * bx) * px
        for which in range(3):
            blk = coefs(rng, 16 * 16, depth)
            nn = np.zeros(15 * 8, np.uint8)
            for i in range(16):
                mode = rng.integers(0, 4)
                nn[SCAN8[i]] = [0, 1, 1, 5][mode]
                if mode == 0 and rng.random() < .5:
                    blk[16 * i:16 * i + 16] = 0
                if mode == 1:
                    blk[16 * i + 1:16 * i + 16] = 0
                if mode == 2:
                    blk[16 * i] = 0
            bo8 = bo.copy()
            if which == 1:
                for i in range(0, 16, 4):
                    bo8[i] = ((i >> 3) * 8 * stride_px + ((i >> 2) & 1) * 8) * px
                blk = coefs(rng, 4 * 64, depth)
            a, b, ca, cb = d0.copy(), d0.copy(), blk.copy(), blk.copy()
            R.ffref_h264_idct_multi(which, at(a, 2, 4), bo8.ctypes.data_as(C.POINTER(C.c_int)), bptr(ca), stride, ptr(nn))
            O.ffo_h264_idct_mb_bd(depth, which, at(b, 2, 4), bo8.ctypes.data_as(C.POINTER(C.c_int)), bptr(cb), stride, ptr(nn))
            assert np.array_equal(a, b) and np.array_equal(ca, cb), (which, rep)
        for is422 in (0, 1):
            blk = coefs(rng, 48 * 16, depth)
            nn = rng.choice(np.array([0, 0, 1, 3], np.uint8), 15 * 8)
            zero = rng.random(48) < .3
            for i in range(48):
                if zero[i]:
                    blk[16 * i] = 0
            bo2 = np.zeros(48, np.int32)
            for j in (1, 2):
                for k in range(8):
                    idx = 16 * j + k + (4 if k >= 4 else 0) if is422 else 16 * j + k
                    if idx < 48:
                        bo2[idx] = ((k >> 1) * 4 * stride_px + (k & 1) * 4) * px
            pl = [pixels(rng, (40, stride_px), depth), pixels(rng, (40, stride_px), depth)]
            pa, pb = [p.copy() for p in pl], [p.copy() for p in pl]
            ca, cb = blk.copy(), blk.copy()
            da = (u8p * 2)(at(pa[0], 1, 2), at(pa[1], 1, 2))
            db = (u8p * 2)(at(pb[0], 1, 2), at(pb[1], 1, 2))
            R.ffref_h264_idct_add8(da, bo2.ctypes.data_as(C.POINTER(C.c_int)), bptr(ca), stride, ptr(nn), 2 if is422 else 1)
            O.ffo_h264_idct_add8_bd(depth, is422, db, bo2.ctypes.data_as(C.POINTER(C.c_int)), bptr(cb), stride, ptr(nn))
            assert all(np.array_equal(x, y) for x, y in zip(pa, pb)) and np.array_equal(ca, cb), (is422, rep)
        qmul = int(rng.integers(1, 1 << 12))
        inp = coefs(rng, 16, depth, big=True)
        out0 = coefs(rng, 256, depth)
        oa, ob = out0.copy(), out0.copy()
        R.ffref_h264_luma_dc_dequant_idct(bptr(oa), bptr(inp.copy()), qmul)
        O.ffo_h264_luma_dc_dequant_bd(depth, bptr(ob), bptr(inp.copy()), qmul)
        assert np.array_equal(oa, ob)
        for is422 in (0, 1):
            blk = coefs(rng, 256, depth, big=True)
            ca, cb = blk.copy(), blk.copy()
            (R.ffref_h264_chroma_dc_dequant_idct_422 if is422 else R.ffref_h264_chroma_dc_dequant_idct)(bptr(ca), qmul)
            O.ffo_h264_chroma_dc_dequant_bd(depth, is422, bptr(cb), qmul)
            assert np.array_equal(ca, cb), is422


ALPHA = [0, 4, 7, 17, 40, 90, 182, 255]
BETA = [0, 2, 3, 6, 9, 13, 16, 18]


@pytest.mark.parametrize("depth", DEPTHS, indirect=True)
def test_h264_loop_filter_bd(depth):
    """every member of the loop-filter family: v/h x luma/chroma x normal/intra, the MBAFF forms and the 4:2:2 chroma forms"""
    R, O = ffi.ref(), ffi.oracle()
    _sigs(R, O)
    rng = np.random.default_rng(1000 + depth)
    # (kind, variant, inner): variant 0 plain, 1 MBAFF, 2 4:2:2, 3 4:2:2 MBAFF
    fam = [(0, 0, 4), (1, 0, 4), (1, 1, 2), (4, 0, 4), (5, 0, 4), (5, 1, 2), (2, 0, 2), (3, 0, 2), (3, 1, 1), (3, 2, 4), (3, 3, 2),
           (6, 0, 2), (7, 0, 2), (7, 1, 1), (7, 2, 4), (7, 3, 2)]
    for kind, variant, inner in fam:
        for rep in range(40):
            base = int(rng.integers(0, 1 << depth))
            amp = [1, 3, 12, 60][rep % 4] << (depth - 8)
            d0 = np.clip(base + rng.integers(-amp, amp + 1, (40, 40)), 0, (1 << depth) - 1).astype(np.uint16 if depth > 8 else np.uint8)
            if rep % 7 == 0:
                d0 = pixels(rng, (40, 40), depth, True)
            tc = rng.integers(-1, 6, 4).astype(np.int8)
            alpha, beta = ALPHA[rep % 8], BETA[(rep // 2) % 8]
            a, b = d0.copy(), d0.copy()
            stride = d0.strides[0]
            R.ffref_h264_loop_filter_variant(kind, variant, at(a, 12, 12), stride, alpha, beta, tc.ctypes.data_as(C.POINTER(C.c_int8)))
            O.ffo_h264_loop_filter_bd(depth, kind, inner, at(b, 12, 12), stride, alpha, beta, tc.ctypes.data_as(C.POINTER(C.c_int8)))
            assert np.array_equal(a, b), (kind, variant, rep)


@pytest.mark.parametrize("depth", DEPTHS, indirect=True)
def test_h264_qpel_chroma_weight_bd(depth):
    R, O = ffi.ref(), ffi.oracle()
    _sigs(R, O)
    rng = np.random.default_rng(1100 + depth)
    for avg in (0, 1):
        for size_idx in range(3):
            for mcxy in range(16):
                src = pixels(rng, (32, 40), depth, mcxy % 5 == 0)
                d0 = pixels(rng, (32, 40), depth)
                a, b = d0.copy(), d0.copy()
                stride = d0.strides[0]
                R.ffref_h264_qpel(avg, size_idx, mcxy, at(a, 6, 8), at(src, 6, 8), stride)
                O.ffo_h264_qpel_bd(depth, avg, size_idx, mcxy, at(b, 6, 8), at(src, 6, 8), stride)
                assert np.array_equal(a, b), (avg, size_idx, mcxy)
        for idx, w in ((0, 8), (1, 4), (2, 2)):
            for rep in range(24):
                x, y = int(rng.integers(0, 8)), int(rng.integers(0, 8))
                if rep == 0:
                    x = y = 0
                h = [2, 4, 8, 16][rep % 4]
                src = pixels(rng, (24, 24), depth, rep % 6 == 0)
                d0 = pixels(rng, (24, 24), depth)
                a, b = d0.copy(), d0.copy()
                stride = d0.strides[0]
                R.ffref_h264_chroma(avg, idx, at(a, 2, 4), at(src, 2, 4), stride, h, x, y)
                O.ffo_h264_chroma_mc_bd(depth, avg, w, at(b, 2, 4), at(src, 2, 4), stride, h, x, y)
                assert np.array_equal(a, b), (avg, w, rep)
    for idx, w in ((0, 16), (1, 8), (2, 4), (3, 2)):
        for rep in range(30):
            h = [2, 4, 8, 16][rep % 4]
            ld = int(rng.integers(0, 8))
            wt, ws, off = int(rng.integers(-128, 128)), int(rng.integers(-128, 128)), int(rng.integers(-128, 128))
            d0 = pixels(rng, (20, 24), depth, rep % 5 == 0)
            src = pixels(rng, (20, 24), depth)
            a, b = d0.copy(), d0.copy()
            stride = d0.strides[0]
            R.ffref_h264_weight(idx, at(a, 1, 4), stride, h, ld, wt, off)
            O.ffo_h264_weight_bd(depth, w, at(b, 1, 4), stride, h, ld, wt, off)
            assert np.array_equal(a, b), ("weight", w, rep)
            a, b = d0.copy(), d0.copy()
            R.ffref_h264_biweight(idx, at(a, 1, 4), at(src, 1, 4), stride, h, ld, wt, ws, off)
            O.ffo_h264_biweight_bd(depth, w, at(b, 1, 4), at(src, 1, 4), stride, h, ld, wt, ws, off)
            assert np.array_equal(a, b), ("biweight", w, rep)
