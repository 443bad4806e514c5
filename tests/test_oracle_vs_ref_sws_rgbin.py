"""Packed 8-bit RGB sources of the legacy scaler (rgb24 / bgr24 / rgba / bgra / argb / abgr -> a YUV target) on the CPU tier:
the reference's sws_scale() on the real formats == oracle/ffo_sws_rgbin.c (the input converters' int16 lines) followed by
oracle/ffo_sws_hbd.c on those lines as a 14-bit planar source without dither — the equivalence libffhip's RGB-source contexts rest on
(kernels/sws_rgbin.hip) — at equal sizes and scaled, half and full chroma input, odd widths, every component order."""
import ctypes as C
import os

import numpy as np
import pytest

import ffi
from ffi import PIX, u8p
from test_oracle_vs_ref_sws_hbd import planes_of

# name -> (bytes per pixel, (R, G, B byte))
RGB = {"rgb24": (3, (0, 1, 2)), "bgr24": (3, (2, 1, 0)), "rgba": (4, (0, 1, 2)), "bgra": (4, (2, 1, 0)), "argb": (4, (1, 2, 3)), "abgr": (4, (3, 2, 1))}
# target -> (AVPixelFormat, layout of ffo_sws_scale_frame_hbd (0 planar, 2 nv12), hsub, vsub)
DST = {"yuv420p": (0, 0, 1, 1), "nv12": (23, 2, 1, 1), "yuv422p": (4, 0, 1, 0), "yuv444p": (5, 0, 0, 0)}
P14 = {1: 129, 0: 133}   # chroma at half width -> yuv422p14le, else yuv444p14le
FULL_CHR_H_INP = 0x4000

CASES = [("rgb24", 64, 36, "yuv420p", 64, 36, ffi.SWS_BICUBIC), ("rgb24", 64, 36, "yuv420p", 128, 72, ffi.SWS_BICUBIC),
         ("bgra", 96, 54, "nv12", 64, 36, ffi.SWS_BILINEAR), ("rgba", 64, 36, "yuv444p", 64, 36, ffi.SWS_BICUBIC),
         ("argb", 65, 37, "yuv420p", 65, 37, ffi.SWS_BICUBIC), ("abgr", 64, 36, "yuv420p", 200, 36, ffi.SWS_BICUBIC),
         ("bgr24", 64, 36, "nv12", 64, 36, ffi.SWS_BICUBIC), ("rgb24", 64, 36, "yuv420p", 64, 36, ffi.SWS_BICUBIC | FULL_CHR_H_INP),
         ("bgra", 256, 144, "yuv420p", 256, 144, ffi.SWS_BICUBIC), ("bgr24", 96, 54, "yuv422p", 48, 30, ffi.SWS_BICUBIC),
         ("rgba", 200, 120, "nv12", 50, 30, ffi.SWS_BICUBIC), ("bgr24", 64, 36, "yuv420p", 66, 36, ffi.SWS_BICUBIC),
         ("rgb24", 62, 34, "yuv420p", 62, 34, ffi.SWS_POINT), ("bgra", 64, 36, "yuv420p", 64, 36, ffi.SWS_BICUBIC | 0x40000)]


def make_rgb(name, w, h, rng, pad=5):
    bpp, _ = RGB[name]
    a = rng.integers(0, 256, (h, w * bpp + pad), dtype=np.uint8)
    a[::5, : (w // 2) * bpp] = 255
    a[3::7, (w // 3) * bpp:] = 0
    return a


def alloc_dst(name, w, h, pad=4):
    _, layout, hs, vs = DST[name]
    cw, ch = -((-w) >> hs), -((-h) >> vs)
    if layout == 2:
        return [np.zeros((h, w + pad), np.uint8), np.zeros((ch, 2 * cw + pad), np.uint8)]
    return [np.zeros((h, w + pad), np.uint8), np.zeros((ch, cw + pad), np.uint8), np.zeros((ch, cw + pad), np.uint8)]


def oracle_rgb_scale(sname, rgb, sw, sh, dname, dw, dh, flags):
    """the oracle's composite: converter lines, then the 14-bit planar scaler with the dither off; returns the target planes"""
    from ffmpeg_amd import swscale as S
    O = ffi.oracle()
    bpp, (ro, go, bo) = RGB[sname]
    dfmt, dlayout, hs, _ = DST[dname]
    O.ffo_sws_rgb_half.argtypes = [C.c_int] * 4
    O.ffo_sws_rgb2yuv_default.argtypes = [C.POINTER(C.c_int32)]
    O.ffo_sws_rgb_in.argtypes = [C.c_void_p, C.c_ssize_t] + [C.c_int] * 7 + [C.POINTER(C.c_int32), C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p, C.c_ssize_t]
    O.ffo_sws_rgb_in.restype = None
    half = O.ffo_sws_rgb_half(sw, dw, hs, flags)
    tab = (C.c_int32 * 9)()
    O.ffo_sws_rgb2yuv_default(tab)
    cw = sw // 2 if half else sw
    Y, U, V = np.zeros((sh, sw + 3), np.uint16), np.zeros((sh, cw + 3), np.uint16), np.zeros((sh, cw + 3), np.uint16)
    O.ffo_sws_rgb_in(rgb.ctypes.data, rgb.strides[0], sw, sh, bpp, ro, go, bo, half, tab, Y.ctypes.data, Y.strides[0], U.ctypes.data, V.ctypes.data,
                     U.strides[0])
    inner = P14[half]
    ht = S.HostTables(sw, sh, inner, dw, dh, dfmt, flags)
    t = ffi.make_otables(sw, sh, inner, dw, dh, dfmt, flags, ht.banks(), ht.coeffs())
    O.ffo_sws_scale_frame_hbd.argtypes = [C.POINTER(ffi.OSwsTables), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(C.c_int),
                                          C.POINTER(u8p), C.POINTER(C.c_int)]
    got = alloc_dst(dname, dw, dh)
    sp, ss = planes_of([Y, U, V])
    gp, gs = planes_of(got)
    assert O.ffo_sws_scale_frame_hbd(C.byref(t), 14 | 0x100, 0, 8, dlayout, sp, ss, gp, gs) == 0
    return got, half


@pytest.mark.skipif(not os.path.exists(ffi.REF_SO), reason="oracle/_ref not built")
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_rgb_source_is_its_converter_lines_as_a_14_bit_planar_source(case):
    sname, sw, sh, dname, dw, dh, flags = case
    R = ffi.ref()
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    rgb = make_rgb(sname, sw, sh, rng)
    want = alloc_dst(dname, dw, dh)
    ctx = R.ffref_sws_create(sw, sh, PIX[sname], dw, dh, DST[dname][0], flags, 1)
    assert ctx and not R.ffref_sws_is_unscaled(ctx)
    sp, ss = planes_of([rgb])
    wp, ws = planes_of(want)
    assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, wp, ws) == dh
    R.ffref_sws_free(ctx)
    got, _ = oracle_rgb_scale(sname, rgb, sw, sh, dname, dw, dh, flags)
    for i, (a, b) in enumerate(zip(want, got)):
        assert np.array_equal(a, b), "plane %d: %d of %d samples differ (max %d)" % (i, (a != b).sum(), a.size, np.abs(a.astype(int) - b.astype(int)).max())


@pytest.mark.skipif(not os.path.exists(ffi.REF_SO), reason="oracle/_ref not built")
def test_bgr24_to_yuv420p_at_equal_size_is_the_special_converter():
    """the one RGB-source conversion this path must NOT take: bgr24ToYv12Wrapper (swscale_unscaled.c:2483-2491)"""
    R = ffi.ref()
    ctx = R.ffref_sws_create(64, 36, PIX["bgr24"], 64, 36, 0, ffi.SWS_BICUBIC, 1)
    assert ctx and R.ffref_sws_is_unscaled(ctx)
    R.ffref_sws_free(ctx)
    ctx = R.ffref_sws_create(64, 36, PIX["rgb24"], 64, 36, 0, ffi.SWS_BICUBIC, 1)
    assert ctx and not R.ffref_sws_is_unscaled(ctx)
    R.ffref_sws_free(ctx)


def test_golden_vectors():
    """the committed reference outputs (tests/golden/sws_rgbin.npz, tools/make_golden.py sws_rgbin) against the oracle's composite: what
    pins it where /root/reference does not exist"""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "sws_rgbin.npz"))
    names = {v: k for k, v in PIX.items()}
    dn = {v[0]: k for k, v in DST.items()}
    for i in range(int(d["ncases"][0])):
        sf, sw, sh, df, dw, dh, fl = (int(v) for v in d["c%d_meta" % i])
        got, _ = oracle_rgb_scale(names[sf], np.ascontiguousarray(d["c%d_src" % i]), sw, sh, dn[df], dw, dh, fl)
        for p, g in enumerate(got):
            a = d["c%d_dst%d" % (i, p)]
            assert np.array_equal(a, g[:, :a.shape[1]]), (i, p)
