"""The legacy scaler above 8 bits: oracle (oracle/ffo_sws_hbd.c) == the reference's sws_scale() on the real pixel formats
(yuv4xxp9/10/12/14/16le, p010le / p012le / p016le, mixed with yuv420p / nv12), scaled contexts, several scalers and ratios; and the
banks libffhip's host side derives (ffhip_sws_tables_create) == the banks of the reference's context for the same formats.
Frames hold extreme samples as well as noise (the clipping and the wrap-around of the horizontal sums are part of the result)."""
import ctypes as C
import os

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p

pytestmark = pytest.mark.skipif(not os.path.exists(ffi.REF_SO), reason="oracle/_ref not built")

# name -> (AVPixelFormat, depth, layout (0 planar / 1 semi-planar msb / 2 nv12), hsub, vsub)
FMT = {
    "yuv420p": (0, 8, 0, 1, 1), "nv12": (23, 8, 2, 1, 1), "yuv422p": (4, 8, 0, 1, 0), "yuv444p": (5, 8, 0, 0, 0),
    "yuv420p9le": (60, 9, 0, 1, 1), "yuv420p10le": (62, 10, 0, 1, 1), "yuv420p12le": (123, 12, 0, 1, 1), "yuv420p14le": (125, 14, 0, 1, 1),
    "yuv420p16le": (45, 16, 0, 1, 1), "yuv422p10le": (64, 10, 0, 1, 0), "yuv444p10le": (68, 10, 0, 0, 0), "yuv444p16le": (49, 16, 0, 0, 0),
    "yuv422p12le": (127, 12, 0, 1, 0), "p010le": (158, 10, 1, 1, 1), "p012le": (209, 12, 1, 1, 1), "p016le": (169, 16, 1, 1, 1),
}


def make_frame(name, w, h, rng, pad=0):
    """planes of one frame as 2-D arrays (uint8 or uint16), noise with runs of extreme samples; None rng: zeros"""
    _, depth, layout, hs, vs = FMT[name]
    cw, ch = -((-w) >> hs), -((-h) >> vs)
    dt = np.uint16 if depth > 8 else np.uint8

    def mk(r, c):
        a = np.zeros((r, c + pad), dt)
        if rng is not None:
            a[:] = rng.integers(0, 1 << depth, a.shape)
            a[::5, : c // 2] = (1 << depth) - 1
            a[3::7, c // 3:] = 0
            if layout == 1:
                a[:] = a << (16 - depth)
        return a
    if layout == 0:
        return [mk(h, w), mk(ch, cw), mk(ch, cw)]
    return [mk(h, w), mk(ch, 2 * cw)]


def planes_of(arrs):
    p = (u8p * 4)()
    s = (C.c_int * 4)()
    for i, a in enumerate(arrs):
        p[i] = C.cast(a.ctypes.data, u8p)
        s[i] = a.strides[0]
    return p, s


CASES = [
    ("yuv420p10le", 64, 36, "yuv420p10le", 128, 72, ffi.SWS_BICUBIC),
    ("yuv420p10le", 96, 54, "p010le", 192, 108, ffi.SWS_BICUBIC),
    ("p010le", 96, 54, "yuv420p10le", 64, 36, ffi.SWS_BILINEAR),
    ("p010le", 128, 72, "p010le", 200, 90, ffi.SWS_BICUBIC),
    ("yuv420p", 64, 36, "yuv420p10le", 128, 72, ffi.SWS_BICUBIC),
    ("nv12", 64, 36, "p010le", 96, 54, ffi.SWS_BILINEAR),
    ("yuv420p10le", 128, 72, "yuv420p", 64, 36, ffi.SWS_BICUBIC),        # 8-bit target from a deeper source: dithered
    ("p010le", 128, 72, "nv12", 96, 40, ffi.SWS_BICUBIC),
    ("yuv420p12le", 64, 36, "yuv420p12le", 80, 44, ffi.SWS_BICUBIC),
    ("yuv420p9le", 64, 36, "yuv420p14le", 96, 40, ffi.SWS_BILINEAR),
    ("yuv420p16le", 64, 36, "yuv420p16le", 128, 72, ffi.SWS_BICUBIC),     # 19-bit intermediates
    ("yuv420p10le", 64, 36, "yuv420p16le", 96, 54, ffi.SWS_BICUBIC),
    ("yuv420p", 64, 36, "p016le", 96, 54, ffi.SWS_BICUBIC),
    ("p016le", 96, 54, "yuv420p10le", 64, 36, ffi.SWS_BICUBIC),
    ("yuv422p10le", 64, 36, "yuv444p10le", 96, 54, ffi.SWS_BICUBIC),
    ("yuv444p16le", 48, 30, "yuv422p12le", 96, 60, ffi.SWS_BILINEAR),
    ("yuv420p10le", 64, 36, "yuv420p10le", 64, 72, ffi.SWS_POINT),        # one-tap vertical banks
    ("p012le", 64, 36, "yuv420p12le", 128, 36, ffi.SWS_AREA),
    ("yuv420p10le", 200, 120, "yuv420p10le", 50, 30, ffi.SWS_BICUBIC),   # wide banks (down-scaling)
]


def oracle_tables(sname, sw, sh, dname, dw, dh, flags):
    from ffmpeg_amd import swscale as S
    ht = S.HostTables(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags)
    return ht, ffi.make_otables(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags, ht.banks(), ht.coeffs())


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_scaler_above_8_bits(case):
    sname, sw, sh, dname, dw, dh, flags = case
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    src = make_frame(sname, sw, sh, rng, pad=6)
    want, got = make_frame(dname, dw, dh, None, pad=4), make_frame(dname, dw, dh, None, pad=4)
    ctx = R.ffref_sws_create(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags, 1)
    assert ctx
    sp, ss = planes_of(src)
    wp, ws = planes_of(want)
    assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, wp, ws) == dh
    # the banks libffhip's host side derives for these formats are the reference's
    ht, t = oracle_tables(sname, sw, sh, dname, dw, dh, flags)
    rb, ob = ffi.ref_tables(ctx), ht.banks()
    for k in ("hLum", "hChr", "vLum", "vChr"):
        assert rb[k][2] == ob[k][2] and rb[k][3] == ob[k][3], k
        assert np.array_equal(rb[k][0], ob[k][0]) and np.array_equal(rb[k][1], ob[k][1]), k
    R.ffref_sws_free(ctx)
    O.ffo_sws_scale_frame_hbd.argtypes = [C.POINTER(ffi.OSwsTables), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(C.c_int),
                                          C.POINTER(u8p), C.POINTER(C.c_int)]
    gp, gs = planes_of(got)
    _, sd, sl, _, _ = FMT[sname]
    _, dd, dl, _, _ = FMT[dname]
    assert O.ffo_sws_scale_frame_hbd(C.byref(t), sd, sl, dd, dl, sp, ss, gp, gs) == 0
    for i, (a, b) in enumerate(zip(want, got)):
        assert np.array_equal(a, b), "plane %d: %d of %d samples differ (max %d)" % (i, (a != b).sum(), a.size, np.abs(a.astype(int) - b.astype(int)).max())


RANGE_CASES = [("yuv420p10le", 64, 36, "yuv420p10le", 128, 72, ffi.SWS_BICUBIC, 1, 0), ("yuv420p10le", 64, 36, "yuv420p10le", 128, 72, ffi.SWS_BICUBIC, 0, 1),
               ("p010le", 96, 54, "p010le", 96, 54, ffi.SWS_BICUBIC, 0, 1),          # equal sizes with a range change: the scaler, not a copy
               ("yuv420p16le", 64, 36, "yuv420p16le", 96, 54, ffi.SWS_BICUBIC, 1, 0),  # 19-bit intermediates: lumRangeFromJpeg16_c
               ("yuv420p16le", 64, 36, "p016le", 96, 54, ffi.SWS_BILINEAR, 0, 1),
               ("yuv420p", 64, 36, "yuv420p10le", 96, 54, ffi.SWS_BICUBIC, 1, 0), ("yuv420p12le", 96, 54, "yuv420p", 64, 36, ffi.SWS_BICUBIC, 0, 1)]


@pytest.mark.parametrize("case", RANGE_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x_%d%d" % c)
def test_range_conversion_above_8_bits(case):
    """c->lumConvertRange / chrConvertRange at 9..16 bits (lumRangeToJpeg_c ... for targets up to 14 bits, the ...16_c forms with 64-bit
    products above; libswscale/swscale.c:160-255, 591-660), ranges set the way sws_setColorspaceDetails() sets them"""
    sname, sw, sh, dname, dw, dh, flags, sr, dr = case
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    src = make_frame(sname, sw, sh, rng, pad=6)
    want, got = make_frame(dname, dw, dh, None, pad=4), make_frame(dname, dw, dh, None, pad=4)
    ctx = R.ffref_sws_create_ranges(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags, 1, sr, dr)
    assert ctx and not R.ffref_sws_is_unscaled(ctx)
    sp, ss = planes_of(src)
    wp, ws = planes_of(want)
    assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, wp, ws) == dh
    banks = ffi.ref_tables(ctx)
    R.ffref_sws_free(ctx)
    _, sd, sl, _, _ = FMT[sname]
    _, dd, dl, _, _ = FMT[dname]
    t = ffi.make_otables(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags, banks, ranges=(sr, dr), dst_depth=dd)
    O.ffo_sws_scale_frame_hbd.argtypes = [C.POINTER(ffi.OSwsTables), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(C.c_int),
                                          C.POINTER(u8p), C.POINTER(C.c_int)]
    gp, gs = planes_of(got)
    assert O.ffo_sws_scale_frame_hbd(C.byref(t), sd, sl, dd, dl, sp, ss, gp, gs) == 0
    for i, (a, b) in enumerate(zip(want, got)):
        assert np.array_equal(a, b), "plane %d: %d of %d samples differ (max %d)" % (i, (a != b).sum(), a.size, np.abs(a.astype(int) - b.astype(int)).max())


# round 6: sources above 8 bits into packed 8-bit RGB (HDR / 10-bit video for a display): name -> (AVPixelFormat, bytes per pixel)
RGBT = {"rgb24": (2, 3), "bgr24": (3, 3), "argb": (25, 4), "rgba": (26, 4), "abgr": (27, 4), "bgra": (28, 4)}
RGB_CASES = [("yuv420p10le", 64, 36, "rgb24", 64, 36, ffi.SWS_BICUBIC), ("p010le", 96, 54, "bgra", 192, 108, ffi.SWS_BICUBIC),
             ("yuv422p10le", 128, 72, "rgba", 64, 36, ffi.SWS_BICUBIC), ("yuv420p12le", 64, 36, "argb", 96, 54, ffi.SWS_BILINEAR),
             ("yuv420p10le", 64, 36, "bgr24", 128, 72, ffi.SWS_BILINEAR),     # two-tap banks on both: yuv2rgb_2 (no rounding term)
             ("p010le", 128, 72, "abgr", 100, 40, ffi.SWS_BILINEAR), ("yuv420p9le", 64, 36, "rgb24", 64, 72, ffi.SWS_POINT),
             ("yuv420p10le", 64, 36, "rgb24", 64, 36, ffi.SWS_BILINEAR),      # one luma tap, a blending chroma pair: yuv2rgb_1 with uvalpha
             ("yuv420p14le", 62, 34, "bgra", 124, 34, ffi.SWS_BICUBIC)]


def oracle_hbd_rgb(sname, src, sw, sh, dname, dw, dh, flags):
    O = ffi.oracle()
    dfmt, bpp = RGBT[dname]
    ht, t = oracle_tables_fmt(sname, sw, sh, dfmt, dw, dh, flags)
    O.ffo_sws_scale_frame_hbd.argtypes = [C.POINTER(ffi.OSwsTables), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(C.c_int),
                                          C.POINTER(u8p), C.POINTER(C.c_int)]
    got = np.zeros((dh, dw * bpp + 5), np.uint8)
    sp, ss = planes_of(src)
    gp, gs = planes_of([got])
    _, sd, sl, _, _ = FMT[sname]
    assert O.ffo_sws_scale_frame_hbd(C.byref(t), sd, sl, 8, 3, sp, ss, gp, gs) == 0
    return got


def oracle_tables_fmt(sname, sw, sh, dfmt, dw, dh, flags):
    from ffmpeg_amd import swscale as S
    ht = S.HostTables(sw, sh, FMT[sname][0], dw, dh, dfmt, flags)
    return ht, ffi.make_otables(sw, sh, FMT[sname][0], dw, dh, dfmt, flags, ht.banks(), ht.coeffs())


@pytest.mark.parametrize("case", RGB_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_deeper_sources_into_packed_rgb(case):
    """hScale16To15_c lines through yuv2rgb_X / _2 / _1 (libswscale/vscale.c:126-170, output.c:1789-1939): the oracle's RGB branch of
    ffo_sws_scale_frame_hbd == the reference's sws_scale(), the banks of libffhip's host side == the reference's"""
    sname, sw, sh, dname, dw, dh, flags = case
    R = ffi.ref()
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    src = make_frame(sname, sw, sh, rng, pad=6)
    dfmt, bpp = RGBT[dname]
    want = np.zeros((dh, dw * bpp + 5), np.uint8)
    ctx = R.ffref_sws_create(sw, sh, FMT[sname][0], dw, dh, dfmt, flags, 1)
    assert ctx and not R.ffref_sws_is_unscaled(ctx)
    sp, ss = planes_of(src)
    wp, ws = planes_of([want])
    assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, wp, ws) == dh
    ht, _ = oracle_tables_fmt(sname, sw, sh, dfmt, dw, dh, flags)
    rb, ob = ffi.ref_tables(ctx), ht.banks()
    for k in ("hLum", "hChr", "vLum", "vChr"):
        assert rb[k][2] == ob[k][2] and rb[k][3] == ob[k][3], k
        assert np.array_equal(rb[k][0], ob[k][0]) and np.array_equal(rb[k][1], ob[k][1]), k
    R.ffref_sws_free(ctx)
    got = oracle_hbd_rgb(sname, src, sw, sh, dname, dw, dh, flags)
    assert np.array_equal(want, got), "%d of %d bytes differ (max %d)" % ((want != got).sum(), want.size, np.abs(want.astype(int) - got.astype(int)).max())
