"""CPU tier: what the frame-level SwsFunc hook (integration/swscale_unscaled_hip.c) rests on, pinned against the reference compiled in place.

1. ffhip_sws_yuv2rgb_coeffs() — the coefficient arithmetic of ff_yuv2rgb_c_init_tables() (libswscale/yuv2rgb.c:750-797) from the values a
   context stores (matrix row, range, brightness / contrast / saturation): the six int16 context fields the reference keeps are equal,
   and the oracle's LUT converter fed with our seven coefficients reproduces the reference's sws_scale() after
   sws_setColorspaceDetails() byte for byte.
2. ffo_yuv2rgb_unscaled() — the oracle's restatement of the converter's 4:2:2, source-alpha and planar-gbrp forms (the checker of the
   GPU tests) == the reference's own converter on the same frames."""
import ctypes as C

import numpy as np
import pytest

from tests import ffi
from tests.ffi import PIX, ptr

pytestmark = pytest.mark.skipif(not ffi.have_ref(), reason="oracle/_ref/libffref.so not built (needs /root/reference)")

CS = {"bt709": 1, "fcc": 4, "bt601": 5, "smpte240m": 7, "bt2020": 9}
BCS = [(0, 1 << 16, 1 << 16), (3 << 11, (1 << 16) + 5000, (1 << 16) - 9000), (-(5 << 10), 52000, 90000)]


def our_coeffs(cs, full_range, b, c, s):
    from ffmpeg_amd import _lib
    R = ffi.ref()
    inv = (C.c_int * 4)()
    R.ffref_sws_coefficients(cs, inv)
    t = _lib.SwsTables()
    _lib.check(_lib.lib().ffhip_sws_yuv2rgb_coeffs(C.byref(t), inv, full_range, b, c, s))
    return t


def luts_of(t):
    O = ffi.oracle()
    luts = ffi.OLuts()
    k = ffi.OYuv2RgbCoeffs(t.yuv2rgb_cy, t.yuv2rgb_oy, t.yuv2rgb_crv, t.yuv2rgb_cbu, t.yuv2rgb_cgu, t.yuv2rgb_cgv, t.yuv2rgb_yoffs)
    O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
    return luts


@pytest.mark.parametrize("bcs", BCS)
@pytest.mark.parametrize("full_range", [0, 1])
@pytest.mark.parametrize("cs", sorted(CS))
def test_coefficients_follow_the_context(cs, full_range, bcs):
    R, O = ffi.ref(), ffi.oracle()
    w, h = 96, 34
    rng = np.random.default_rng(CS[cs] * 7 + full_range + abs(bcs[0]))
    src = ffi.alloc_frame(PIX["yuv420p"], w, h, rng, pad=3)
    # all 256 values of every plane are met
    src[0][:8, :64].flat[:256] = np.arange(256, dtype=np.uint8)
    src[1][:8, :32].flat[:256] = np.arange(256, dtype=np.uint8)
    src[2][8:16, :32].flat[:256] = np.arange(256, dtype=np.uint8)
    ctx = R.ffref_sws_create(w, h, PIX["yuv420p"], w, h, PIX["rgb24"], ffi.SWS_BICUBIC, 1)
    assert ctx and R.ffref_sws_is_unscaled(ctx)
    assert R.ffref_sws_set_colorspace(ctx, CS[cs], full_range, *bcs) >= 0
    rk = (C.c_int * 6)()
    R.ffref_sws_full_coeffs(ctx, rk)
    want = ffi.alloc_frame(PIX["rgb24"], w, h)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert R.ffref_sws_scale(ctx, sp, ss, 0, h, dp, ds) == h
    R.ffref_sws_free(ctx)
    t = our_coeffs(CS[cs], full_range, *bcs)
    assert [int(v) for v in t.yuv2rgb_full] == list(rk)
    luts = luts_of(t)
    got = np.zeros_like(want[0])
    O.ffo_yuv420p_to_rgb24(C.byref(luts), w, sp, ss, 0, h, ptr(got), got.strides[0], 0)
    assert np.array_equal(got, want[0]), "%d bytes differ" % (got != want[0]).sum()


FORMS = [(s, d) for s in ("yuv420p", "yuv422p", "yuva420p") for d in ("rgb24", "bgr24", "argb", "rgba", "abgr", "bgra", "gbrp")]


@pytest.mark.parametrize("w,h", [(64, 16), (354, 10), (30, 4)])
@pytest.mark.parametrize("sf,df", FORMS)
def test_oracle_forms_equal_the_reference_converter(sf, df, w, h):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(w * 3 + h + len(sf) * 11 + len(df))
    src = ffi.alloc_frame(PIX[sf], w, h, rng, pad=5)
    ctx = R.ffref_sws_create(w, h, PIX[sf], w, h, PIX[df], ffi.SWS_BICUBIC, 1)
    assert ctx and R.ffref_sws_is_unscaled(ctx), "the reference picks the table converter"
    want = ffi.alloc_frame(PIX[df], w, h)
    got = ffi.alloc_frame(PIX[df], w, h)
    for a in want + got:
        a[:] = 0xA5
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert R.ffref_sws_scale(ctx, sp, ss, 0, h, dp, ds) == h
    R.ffref_sws_free(ctx)
    luts = luts_of(our_coeffs(5, 0, 0, 1 << 16, 1 << 16))
    gp, gs = ffi.planes(got)
    alpha = sf == "yuva420p" and df in ("argb", "rgba", "abgr", "bgra")
    assert O.ffo_yuv2rgb_unscaled(C.byref(luts), w, sp, ss, 0, h, gp, gs, ffi.RGB_LAYOUT[PIX[df]], int(sf == "yuv422p"), int(alpha)) == h
    for p, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), "plane %d: %d bytes differ" % (p, (a != b).sum())


def test_sliced_calls_equal_the_frame():
    """the SwsFunc contract: 2-line aligned slices, src[] at the slice's first rows, dst[] at the picture's"""
    O = ffi.oracle()
    w, h = 66, 12
    rng = np.random.default_rng(3)
    src = ffi.alloc_frame(PIX["yuva420p"], w, h, rng)
    luts = luts_of(our_coeffs(5, 0, 0, 1 << 16, 1 << 16))
    whole = ffi.alloc_frame(PIX["bgra"], w, h)
    parts = ffi.alloc_frame(PIX["bgra"], w, h)
    sp, ss = ffi.planes(src)
    O.ffo_yuv2rgb_unscaled(C.byref(luts), w, sp, ss, 0, h, ffi.planes(whole)[0], ffi.planes(whole)[1], 5, 0, 1)
    for y, sh in ((0, 4), (4, 2), (6, 6)):
        sl = [src[0][y:], src[1][y >> 1:], src[2][y >> 1:], src[3][y:]]
        p, s = ffi.planes(sl)
        O.ffo_yuv2rgb_unscaled(C.byref(luts), w, p, s, y, sh, ffi.planes(parts)[0], ffi.planes(parts)[1], 5, 0, 1)
    assert np.array_equal(whole[0], parts[0])


# (source, sw, sh, target, dw, dh, flags): bicubic -> yuv2rgb_X; an exact-2x bilinear has 2-tap 4096-sum rows -> yuv2rgb_2 on the rows where
# both banks are; equal sizes + ACCURATE_RND -> yuv2rgb_1 rows; 0x2000 / 4:4:4 / odd widths -> the _full writers
RGBA_ALPHA_CASES = [("yuva420p", 64, 36, "rgba", 128, 72, ffi.SWS_BICUBIC), ("yuva420p", 64, 36, "argb", 100, 50, ffi.SWS_BICUBIC),
                    ("yuva420p", 96, 54, "bgra", 64, 36, ffi.SWS_BILINEAR), ("yuva420p", 64, 36, "abgr", 128, 72, ffi.SWS_BILINEAR),
                    ("yuva420p", 64, 36, "rgba", 64, 36, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND),
                    ("yuva420p", 64, 36, "bgra", 64, 72, ffi.SWS_POINT), ("yuva422p", 64, 36, "rgba", 96, 54, ffi.SWS_BICUBIC),
                    ("yuva444p", 64, 36, "argb", 96, 54, ffi.SWS_BICUBIC), ("yuva420p", 64, 36, "rgba", 97, 55, ffi.SWS_BICUBIC),
                    ("yuva420p", 64, 36, "bgra", 128, 72, ffi.SWS_BILINEAR | 0x2000), ("yuva444p", 64, 36, "abgr", 64, 36, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND),
                    ("yuva420p", 64, 36, "rgba", 64, 72, ffi.SWS_POINT | 0x2000),
                    ("yuva420p", 64, 36, "rgba", 64, 36, ffi.SWS_BILINEAR | ffi.SWS_ACCURATE_RND),          # yuv2rgb_1 with uvalpha != 0
                    ("yuva420p", 64, 36, "argb", 64, 36, ffi.SWS_BILINEAR | ffi.SWS_ACCURATE_RND | 0x2000), # yuv2rgb_full_1
                    ("yuva420p", 640, 360, "bgra", 1280, 720, ffi.SWS_BICUBIC)]


def rgba_alpha_oracle(sf, sw, sh, df, dw, dh, flags, src):
    """the oracle's frame with the source's alpha plane in the alpha byte; returns (picture, host tables)"""
    from ffmpeg_amd import swscale as S
    O = ffi.oracle()
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    assert ht.t.dst_alpha_fill == 2 and not ht.unscaled_yuv2rgb
    t = ffi.make_otables(sw, sh, ht.t.srcFormat, dw, dh, PIX[df], ht.t.flags, ht.banks(), ht.coeffs(), full=ht.full())
    got = ffi.alloc_frame(PIX[df], dw, dh)
    sp, ss = ffi.planes(src[:3])
    gp, gs = ffi.planes(got)
    assert O.ffo_sws_scale_frame(C.byref(t), sp, ss, gp, gs) == dh
    assert O.ffo_sws_rgba_alpha(C.byref(t), ptr(src[3]), src[3].strides[0], ptr(got[0]), got[0].strides[0]) == dh
    return got, ht


@pytest.mark.parametrize("sf,sw,sh,df,dw,dh,flags", RGBA_ALPHA_CASES)
def test_scaled_source_alpha_into_packed_rgba(sf, sw, sh, df, dw, dh, flags):
    """yuv2rgba32_{1,2,X}_c and the _full twins (libswscale/output.c:1789-1939, 2160-2310): the oracle's picture + ffo_sws_rgba_alpha ==
    the reference's sws_scale() of a YUVA source to the four 32-bit orders"""
    R = ffi.ref()
    rng = np.random.default_rng(sw + dw + len(df) + flags % 97)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=2)
    src[3][:4] = 255   # saturated rows: where the bicubic overshoot makes the writers clip
    src[3][4:8] = 0
    ctx = R.ffref_sws_create(sw, sh, PIX[sf], dw, dh, PIX[df], flags, 1)
    assert ctx and not R.ffref_sws_is_unscaled(ctx)
    want = ffi.alloc_frame(PIX[df], dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, dp, ds) == dh
    R.ffref_sws_free(ctx)
    got, _ = rgba_alpha_oracle(sf, sw, sh, df, dw, dh, flags, src)
    assert np.array_equal(got[0], want[0]), "%d bytes differ" % (got[0] != want[0]).sum()
