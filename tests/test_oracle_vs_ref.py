"""Pin oracle/ (our CPU restatement) bit-exact against the REAL reference compiled from
/root/reference (oracle/_ref/libffref.so).  Skipped where that library is absent; the committed
fixtures in tests/golden/ (generated from it by tools/make_golden.py) cover that case."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import PIX, ptr, u8p, i8p, i16p, i32p, f32p

pytestmark = pytest.mark.skipif(not ffi.have_ref(), reason="oracle/_ref/libffref.so not built")


def _ref_convert(srcfmt, sw, sh, dstfmt, dw, dh, flags, src):
    R = ffi.ref()
    ctx = R.ffref_sws_create(sw, sh, srcfmt, dw, dh, dstfmt, flags, 1)
    assert ctx
    dst = ffi.alloc_frame(dstfmt, dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(dst)
    assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, dp, ds) == dh
    banks = None if R.ffref_sws_is_unscaled(ctx) else ffi.ref_tables(ctx)
    R.ffref_sws_free(ctx)
    return dst, banks


@pytest.mark.parametrize("w,h", [(64, 16), (1920, 1080), (1078, 6), (1076, 4), (30, 2)])
@pytest.mark.parametrize("dst", ["rgb24", "bgr24", "argb", "rgba", "abgr", "bgra"])
def test_yuv420p_rgb24_table_path(w, h, dst):
    rng = np.random.default_rng(w * 31 + h)
    src = ffi.alloc_frame(PIX["yuv420p"], w, h, rng, pad=5)
    want, banks = _ref_convert(PIX["yuv420p"], w, h, PIX[dst], w, h, ffi.SWS_BICUBIC, src)
    assert banks is None, "reference should pick the unscaled table converter"
    O = ffi.oracle()
    luts = ffi.OLuts()
    k = ffi.OYuv2RgbCoeffs(*[ffi.DEFAULT_COEFFS[n] for n in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
    O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
    got = np.zeros_like(want[0])
    sp, ss = ffi.planes(src)
    O.ffo_yuv420p_to_rgb24(C.byref(luts), w, sp, ss, 0, h, ptr(got), got.strides[0], ffi.RGB_LAYOUT[PIX[dst]])
    assert np.array_equal(got, want[0])


def test_yuv420p_rgb24_exhaustive_yuv():
    """every (Y,U,V) triple once: 4096x4096 frame would be 16M px; use all U,V with a Y sweep"""
    w, h = 512, 256
    y = np.tile(np.arange(256, dtype=np.uint8).repeat(2), (h, 1))[:, :w].copy()
    y = ((np.arange(w)[None, :] + np.arange(h)[:, None] * 7) & 255).astype(np.uint8)
    u = np.tile(np.arange(256, dtype=np.uint8), (h // 2, 1)).copy()
    v = np.tile(np.arange(128, dtype=np.uint8)[:, None] * 2, (1, w // 2)).astype(np.uint8).copy()
    src = [y, u, v]
    want, _ = _ref_convert(0, w, h, 2, w, h, 4, src)
    O = ffi.oracle()
    luts = ffi.OLuts()
    k = ffi.OYuv2RgbCoeffs(*[ffi.DEFAULT_COEFFS[n] for n in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
    O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
    got = np.zeros_like(want[0])
    sp, ss = ffi.planes(src)
    O.ffo_yuv420p_to_rgb24(C.byref(luts), w, sp, ss, 0, h, ptr(got), got.strides[0], 0)
    assert np.array_equal(got, want[0])


SCALE_CASES = [
    ("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_BICUBIC),
    ("nv12", 160, 90, "nv12", 100, 62, ffi.SWS_BICUBIC),
    ("nv21", 96, 64, "nv21", 200, 130, ffi.SWS_BILINEAR),
    ("yuv420p", 128, 72, "yuv420p", 256, 144, ffi.SWS_BICUBIC),
    ("yuv420p", 101, 77, "yuv420p", 333, 191, ffi.SWS_BICUBIC),
    ("nv12", 128, 72, "yuv420p", 64, 36, ffi.SWS_AREA),
    ("yuv420p", 128, 72, "nv12", 128, 90, ffi.SWS_POINT | ffi.SWS_ACCURATE_RND),
    ("yuv420p", 176, 144, "rgb24", 352, 288, ffi.SWS_BICUBIC),
    ("yuv420p", 176, 144, "bgr24", 176, 144, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND | ffi.SWS_BITEXACT),
    ("yuv420p", 176, 144, "rgb24", 176, 288, ffi.SWS_BILINEAR),
    ("nv12", 176, 144, "rgb24", 352, 288, ffi.SWS_BILINEAR),
    ("yuv420p", 352, 288, "rgb24", 120, 90, ffi.SWS_BICUBIC),
    ("yuv420p", 176, 144, "rgba", 352, 288, ffi.SWS_BICUBIC),
    ("nv12", 176, 144, "bgra", 352, 288, ffi.SWS_BILINEAR),
    ("yuv420p", 176, 144, "argb", 176, 144, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND | ffi.SWS_BITEXACT),
    ("nv21", 176, 144, "abgr", 120, 200, ffi.SWS_BICUBIC),
    ("nv12", 192, 108, "nv12", 384, 216, 0x200),
    # 4:2:2 / 4:4:4 planar (chroma subsampled horizontally only / not at all), to themselves and across subsamplings
    ("yuv422p", 128, 72, "yuv422p", 256, 144, ffi.SWS_BICUBIC),
    ("yuv444p", 128, 72, "yuv444p", 256, 144, ffi.SWS_BICUBIC),
    ("yuv422p", 192, 108, "yuv422p", 96, 54, ffi.SWS_BICUBIC),
    ("yuv444p", 192, 108, "yuv444p", 96, 54, ffi.SWS_BICUBIC),
    ("yuv444p", 101, 77, "yuv420p", 333, 191, ffi.SWS_BICUBIC),
    ("yuv420p", 128, 72, "yuv444p", 200, 100, ffi.SWS_BILINEAR),
    ("yuv422p", 161, 90, "nv12", 100, 62, ffi.SWS_BICUBIC),
    ("nv21", 96, 64, "yuv422p", 200, 130, ffi.SWS_BICUBIC),
    ("yuv422p", 128, 72, "yuv444p", 128, 72, ffi.SWS_BICUBIC),
    ("yuv444p", 128, 72, "yuv420p", 128, 72, ffi.SWS_BICUBIC),
]


@pytest.mark.parametrize("case", [c for c in SCALE_CASES if "yuv422p" in c or "yuv444p" in c], ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_host_tables_422_444(case):
    """our initFilter() restatement on the 4:2:2 / 4:4:4 formats: chroma plane sizes and banks equal the reference's"""
    from ffmpeg_amd import swscale as S
    sf, sw, sh, df, dw, dh, flags = case
    R = ffi.ref()
    ctx = R.ffref_sws_create(sw, sh, PIX[sf], dw, dh, PIX[df], flags, 1)
    assert ctx and not R.ffref_sws_is_unscaled(ctx)
    banks = ffi.ref_tables(ctx)
    R.ffref_sws_free(ctx)
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    for name in ("hLum", "hChr", "vLum", "vChr"):
        f, p, fs, n = ht.bank(name)
        rf, rp, rfs, rn = banks[name]
        assert (fs, n) == (rfs, rn) and np.array_equal(p, rp) and np.array_equal(f, rf), name


@pytest.mark.parametrize("base,j", [(0, 12), (4, 13), (5, 14)], ids=["yuvj420p", "yuvj422p", "yuvj444p"])
def test_full_range_twins_are_their_base_formats(base, j):
    """yuvjXXXp on BOTH sides: equal ranges, no range conversion — the reference's frames and banks equal the base formats'
    (handle_jpeg, libswscale/utils.c:1019-1050), and our host tables take the pair as the base pair; one J side alone is refused"""
    from ffmpeg_amd import swscale as S
    R = ffi.ref()
    rng = np.random.default_rng(j)
    sw, sh, dw, dh = 64, 40, 160, 88
    src = ffi.alloc_frame(base, sw, sh, rng)
    outs = []
    for fmt in (base, j):
        ctx = R.ffref_sws_create(sw, sh, fmt, dw, dh, fmt, ffi.SWS_BICUBIC, 1)
        assert ctx
        dst = ffi.alloc_frame(base, dw, dh)
        sp, ss = ffi.planes(src)
        dp, ds = ffi.planes(dst)
        assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, dp, ds) == dh
        outs.append((dst, ffi.ref_tables(ctx)))
        R.ffref_sws_free(ctx)
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)
    ht = S.HostTables(sw, sh, j, dw, dh, j, ffi.SWS_BICUBIC)
    assert (ht.t.srcFormat, ht.t.dstFormat) == (base, base)
    for name in ("hLum", "hChr", "vLum", "vChr"):
        f, p, fs, n = ht.bank(name)
        rf, rp, rfs, rn = outs[1][1][name]
        assert (fs, n) == (rfs, rn) and np.array_equal(p, rp) and np.array_equal(f, rf), name
    # a J format on one side only: a range conversion between YUV formats (round 3), refused for packed RGB targets
    ht = S.HostTables(sw, sh, j, dw, dh, base, ffi.SWS_BICUBIC)
    assert (ht.t.srcFormat, ht.t.dstFormat, ht.t.src_range, ht.t.dst_range) == (base, base, 1, 0)


@pytest.mark.parametrize("dst", ["rgb24", "bgra"])
@pytest.mark.parametrize("sw,sh,dw,dh,flags", [(64, 16, 64, 16, ffi.SWS_BICUBIC), (1078, 6, 1078, 6, ffi.SWS_BICUBIC), (64, 40, 160, 88, ffi.SWS_BICUBIC),
                                               (96, 54, 48, 28, ffi.SWS_BILINEAR), (64, 40, 64, 40, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND)])
def test_full_range_yuv_to_rgb(dst, sw, sh, dw, dh, flags):
    """yuvj420p -> packed RGB: the source's range goes into the yuv2rgb tables (ff_yuv2rgb_c_init_tables' fullRange branch,
    libswscale/yuv2rgb.c:749-768), not into a range stage: the oracle with the coefficients libffhip's host side derives for a
    full-range source == the reference, on the unscaled table path and through the scaler"""
    from ffmpeg_amd import swscale as S
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(sw + dw + len(dst))
    src = ffi.alloc_frame(PIX["yuv420p"], sw, sh, rng, pad=3)
    for pl in src:
        pl[::5, : pl.shape[1] // 2] = 255
        pl[3::7, pl.shape[1] // 3:] = 0
    want, banks = _ref_convert(PIX["yuvj420p"], sw, sh, PIX[dst], dw, dh, flags, src)
    ht = S.HostTables(sw, sh, PIX["yuvj420p"], dw, dh, PIX[dst], flags)
    co = ht.coeffs()
    assert co != ffi.DEFAULT_COEFFS and ht.t.src_range == 1
    sp, ss = ffi.planes(src)
    if banks is None:   # the unscaled table converter
        assert ht.unscaled_yuv2rgb
        luts = ffi.OLuts()
        k = ffi.OYuv2RgbCoeffs(*[co[n] for n in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
        O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
        got = np.zeros_like(want[0])
        O.ffo_yuv420p_to_rgb24(C.byref(luts), sw, sp, ss, 0, sh, ptr(got), got.strides[0], ffi.RGB_LAYOUT[PIX[dst]])
        assert np.array_equal(got, want[0])
    else:
        t = ffi.make_otables(sw, sh, PIX["yuv420p"], dw, dh, PIX[dst], flags, banks, co)
        got = ffi.alloc_frame(PIX[dst], dw, dh)
        gp, gs = ffi.planes(got)
        assert O.ffo_sws_scale_frame(C.byref(t), sp, ss, gp, gs) == dh
        assert np.array_equal(got[0], want[0])


RANGE_CASES = [("yuvj420p", 64, 40, "yuv420p", 160, 88, ffi.SWS_BICUBIC), ("yuv420p", 64, 40, "yuvj420p", 160, 88, ffi.SWS_BICUBIC),
               ("yuvj420p", 96, 54, "yuv420p", 96, 54, ffi.SWS_BICUBIC),      # equal sizes: the scaler with one-tap banks, not a copy
               ("yuv420p", 96, 54, "yuvj420p", 96, 54, ffi.SWS_BILINEAR),
               ("yuvj444p", 64, 40, "yuv422p", 48, 30, ffi.SWS_BICUBIC), ("yuv422p", 80, 40, "yuvj420p", 120, 90, ffi.SWS_BILINEAR),
               ("yuvj420p", 64, 40, "nv12", 128, 80, ffi.SWS_BICUBIC), ("nv12", 64, 40, "yuvj420p", 100, 60, ffi.SWS_BICUBIC),
               ("yuvj420p", 200, 120, "yuv420p", 50, 30, ffi.SWS_BICUBIC)]


@pytest.mark.parametrize("case", RANGE_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_range_conversion(case):
    """lumRangeToJpeg_c / lumRangeFromJpeg_c / chrRange*_c between the horizontal and the vertical pass (libswscale/swscale.c:160-207,
    591-660): oracle == reference on extreme samples (the ToJpeg clip, the int16 wrap of FromJpeg) as well as noise; the constants
    libffhip's host side derives == the reference's"""
    from ffmpeg_amd import swscale as S
    sf, sw, sh, df, dw, dh, flags = case
    base = {"yuvj420p": "yuv420p", "yuvj422p": "yuv422p", "yuvj444p": "yuv444p"}
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    bs, bd = base.get(sf, sf), base.get(df, df)
    src = ffi.alloc_frame(PIX[bs], sw, sh, rng, pad=3)
    for pl in src:
        pl[::5, : pl.shape[1] // 2] = 255
        pl[3::7, pl.shape[1] // 3:] = 0
    ctx = R.ffref_sws_create(sw, sh, PIX[sf], dw, dh, PIX[df], flags, 1)
    assert ctx and not R.ffref_sws_is_unscaled(ctx)
    want = ffi.alloc_frame(PIX[bd], dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, dp, ds) == dh
    banks = ffi.ref_tables(ctx)
    R.ffref_sws_free(ctx)
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    ranges = (int(sf in base), int(df in base))
    assert (ht.t.src_range, ht.t.dst_range) == ranges
    t = ffi.make_otables(sw, sh, PIX[bs], dw, dh, PIX[bd], flags, banks, ranges=ranges)
    assert (ht.t.lumConvertRange_coeff, ht.t.lumConvertRange_offset, ht.t.chrConvertRange_coeff, ht.t.chrConvertRange_offset) == \
           (t.lum_rc_coeff, t.lum_rc_offset, t.chr_rc_coeff, t.chr_rc_offset)
    got = ffi.alloc_frame(PIX[bd], dw, dh)
    gp, gs = ffi.planes(got)
    assert O.ffo_sws_scale_frame(C.byref(t), sp, ss, gp, gs) == dh
    for i, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), "plane %d: %d samples differ" % (i, (a != b).sum())


@pytest.mark.parametrize("case", SCALE_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_scaled_frame(case):
    sf, sw, sh, df, dw, dh, flags = case
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=3)
    want, banks = _ref_convert(PIX[sf], sw, sh, PIX[df], dw, dh, flags, src)
    assert banks is not None
    t = ffi.make_otables(sw, sh, PIX[sf], dw, dh, PIX[df], flags, banks)
    got = ffi.alloc_frame(PIX[df], dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(got)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("fs", [1, 4, 8, 12, 16, 32, 40])
def test_hscale_adversarial(fs):
    """checkasm's coefficient torture (tests/checkasm/sw_scale.c:356-458)"""
    R, O = ffi.ref(), ffi.oracle()
    ctx = R.ffref_sws_create(64, 16, 0, 128, 32, 0, 4, 1)
    dstW, srcW = 512, 560
    rng = np.random.default_rng(fs)
    src = rng.integers(0, 256, srcW, dtype=np.uint8)
    filt = rng.integers(-(1 << 14), 1 << 14, (dstW, fs)).astype(np.int16)
    filt[::3] = -((1 << 14) // max(fs - 1, 1))
    filt[::3, 0] = (1 << 15) - 1
    pos = np.sort(rng.integers(0, srcW - fs, dstW)).astype(np.int32)
    a = np.zeros(dstW, np.int16)
    b = np.zeros(dstW, np.int16)
    R.ffref_sws_hyscale(ctx, ptr(a, i16p), dstW, ptr(src), ptr(filt, i16p), ptr(pos, i32p), fs)
    O.ffo_hscale8to15(ptr(b, i16p), dstW, ptr(src), ptr(filt, i16p), ptr(pos, i32p), fs)
    R.ffref_sws_free(ctx)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("fs", [1, 2, 4, 8, 16])
def test_vscale_lines(fs):
    R, O = ffi.ref(), ffi.oracle()
    ctx = R.ffref_sws_create(64, 16, 0, 128, 32, PIX["nv12"], 4, 1)
    dstW = 333
    rng = np.random.default_rng(fs + 100)
    lines = rng.integers(-32768, 32768, (fs, dstW + 8)).astype(np.int16)
    lines2 = rng.integers(-32768, 32768, (fs, dstW + 8)).astype(np.int16)
    filt = rng.integers(-4096, 8192, fs).astype(np.int16)
    dither = rng.integers(0, 128, 8, dtype=np.uint8)
    rows = (i16p * fs)(*[ptr(lines[j], i16p) for j in range(fs)])
    rows2 = (i16p * fs)(*[ptr(lines2[j], i16p) for j in range(fs)])
    for off in (0, 3):
        a = np.zeros(dstW, np.uint8); b = np.zeros(dstW, np.uint8)
        if fs == 1:
            R.ffref_sws_yuv2plane1(ctx, rows[0], ptr(a), dstW, ptr(dither), off)
            O.ffo_yuv2plane1_8(rows[0], ptr(b), dstW, ptr(dither), off)
        else:
            R.ffref_sws_yuv2planeX(ctx, ptr(filt, i16p), fs, rows, ptr(a), dstW, ptr(dither), off)
            O.ffo_yuv2planeX8(ptr(filt, i16p), fs, rows, ptr(b), dstW, ptr(dither), off)
        assert np.array_equal(a, b)
    for fmt, swap in ((PIX["nv12"], 0), (PIX["nv21"], 1)):
        a = np.zeros(2 * dstW, np.uint8); b = np.zeros(2 * dstW, np.uint8)
        R.ffref_sws_yuv2nv12cX(ctx, fmt, ptr(dither), ptr(filt, i16p), fs, rows, rows2, ptr(a), dstW)
        O.ffo_yuv2nv12cX(swap, ptr(dither), ptr(filt, i16p), fs, rows, rows2, ptr(b), dstW)
        assert np.array_equal(a, b)
    R.ffref_sws_free(ctx)


def packed_line_inputs(rng, dstW, lfs, cfs):
    """int16 luma / chroma lines in the 15-bit range a horizontal pass produces, unit-gain vertical banks with negative lobes
    (the reference indexes its 2048-entry luma ramp with the unclamped sample: gains far above 1 leave the table)"""
    lum = rng.integers(0, 32768, (max(lfs, 2), dstW + 8)).astype(np.int16)
    cu = rng.integers(0, 32768, (max(cfs, 2), dstW // 2 + 8)).astype(np.int16)
    cv = rng.integers(0, 32768, (max(cfs, 2), dstW // 2 + 8)).astype(np.int16)

    def bank(n):
        g = rng.dirichlet(np.ones(n)) * 1.16 - 0.16 / n
        f = np.round(g * 4096).astype(np.int32)
        f[-1] += 4096 - f.sum()
        return f.astype(np.int16)
    return lum, cu, cv, bank(lfs), bank(cfs)


@pytest.mark.parametrize("dst", ["rgb24", "bgr24", "argb", "rgba", "abgr", "bgra"])
def test_packed_lines(dst):
    """yuv2packedX / yuv2packed2 / yuv2packed1 of a packed-RGB context (yuv2rgb_{X,2,1}_c_template): oracle == reference"""
    R, O = ffi.ref(), ffi.oracle()
    ctx = R.ffref_sws_create(64, 16, PIX["yuv420p"], 128, 32, PIX[dst], 4, 1)
    luts = ffi.OLuts()
    k = ffi.OYuv2RgbCoeffs(*[ffi.DEFAULT_COEFFS[n] for n in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
    O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
    lay, bpp, dstW = ffi.RGB_LAYOUT[PIX[dst]], (3 if dst in ("rgb24", "bgr24") else 4), 330
    rng = np.random.default_rng(ffi.RGB_LAYOUT[PIX[dst]] + 40)
    for lfs, cfs in ((4, 4), (1, 4), (2, 2), (8, 3)):
        lum, cu, cv, lf, cf = packed_line_inputs(rng, dstW, lfs, cfs)
        rl = (i16p * lfs)(*[ptr(lum[j], i16p) for j in range(lfs)])
        ru = (i16p * cfs)(*[ptr(cu[j], i16p) for j in range(cfs)])
        rv = (i16p * cfs)(*[ptr(cv[j], i16p) for j in range(cfs)])
        a, b = np.zeros(dstW * bpp + 8, np.uint8), np.zeros(dstW * bpp + 8, np.uint8)
        R.ffref_sws_yuv2packedX(ctx, ptr(lf, i16p), rl, lfs, ptr(cf, i16p), ru, rv, cfs, ptr(a), dstW, 5)
        O.ffo_yuv2rgb_X(C.byref(luts), ptr(lf, i16p), rl, lfs, ptr(cf, i16p), ru, rv, cfs, ptr(b), dstW, lay)
        assert np.array_equal(a, b) and a.any()
    lum, cu, cv, _, _ = packed_line_inputs(rng, dstW, 2, 2)
    r2 = [(i16p * 2)(ptr(x[0], i16p), ptr(x[1], i16p)) for x in (lum, cu, cv)]
    for ya, uva in ((0, 0), (1234, 4000), (4096, 2048)):
        a, b = np.zeros(dstW * bpp + 8, np.uint8), np.zeros(dstW * bpp + 8, np.uint8)
        R.ffref_sws_yuv2packed2(ctx, r2[0], r2[1], r2[2], ptr(a), dstW, ya, uva, 5)
        O.ffo_yuv2rgb_2(C.byref(luts), r2[0], r2[1], r2[2], ptr(b), dstW, ya, uva, lay)
        assert np.array_equal(a, b)
    for uva in (0, 1000, 4096):
        a, b = np.zeros(dstW * bpp + 8, np.uint8), np.zeros(dstW * bpp + 8, np.uint8)
        R.ffref_sws_yuv2packed1(ctx, ptr(lum[0], i16p), r2[1], r2[2], ptr(a), dstW, uva, 5)
        O.ffo_yuv2rgb_1(C.byref(luts), ptr(lum[0], i16p), r2[1], r2[2], ptr(b), dstW, uva, lay)
        assert np.array_equal(a, b)
    R.ffref_sws_free(ctx)


# ---------------------------------------------------------------------------------------------
def _coef_blocks(rng, n, size):
    """int16 coefficient blocks spanning the decoder's legal range plus adversarial extremes"""
    c = rng.integers(-2048, 2048, (n, size * size)).astype(np.int16)
    c[::7] = rng.integers(-32768, 32768, c[::7].shape).astype(np.int16)
    c[1::5, 1:] = 0
    return c


@pytest.mark.parametrize("which,size", [(0, 4), (1, 8), (2, 4), (3, 8)])
def test_h264_idct(which, size):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(which)
    ofn = [O.ffo_h264_idct_add, O.ffo_h264_idct8_add, O.ffo_h264_idct_dc_add, O.ffo_h264_idct8_dc_add][which]
    coefs = _coef_blocks(rng, 400, size)
    for c in coefs:
        dst = rng.integers(0, 256, (size, 32), dtype=np.uint8)
        d1, d2, c1, c2 = dst.copy(), dst.copy(), c.copy(), c.copy()
        R.ffref_h264_idct(which, ptr(d1), ptr(c1, i16p), 32)
        ofn(ptr(d2), ptr(c2, i16p), 32)
        assert np.array_equal(d1, d2) and np.array_equal(c1, c2)


@pytest.mark.parametrize("which", [0, 1, 2])
def test_h264_idct_multi(which):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(10 + which)
    ofn = [O.ffo_h264_idct_add16, O.ffo_h264_idct8_add4, O.ffo_h264_idct_add16intra][which]
    bo = np.array([(i & 3) * 4 + (i >> 2) * 4 * 48 for i in range(16)], np.int32)
    if which == 1:
        bo = np.array([((i >> 2) & 1) * 8 + (i >> 3) * 8 * 48 for i in range(16)], np.int32)
    for _ in range(100):
        blocks = rng.integers(-512, 512, 256).astype(np.int16)
        nnzc = rng.integers(0, 3, 40, dtype=np.uint8)
        for i in range(16):
            if rng.random() < .3:
                blocks[i * 16 + 1:(i + 1) * 16] = 0
            if rng.random() < .2:
                blocks[i * 16] = 0
        dst = rng.integers(0, 256, (16, 48), dtype=np.uint8)
        d1, d2, b1, b2 = dst.copy(), dst.copy(), blocks.copy(), blocks.copy()
        R.ffref_h264_idct_multi(which, ptr(d1), ptr(bo, i32p), ptr(b1, i16p), 48, ptr(nnzc))
        ofn(ptr(d2), ptr(bo, i32p), ptr(b2, i16p), 48, ptr(nnzc))
        assert np.array_equal(d1, d2) and np.array_equal(b1, b2)


def h264_misc_cases(rng, n=60):
    """inputs of idct_add8 / luma + chroma dc_dequant_idct / add_pixels{4,8}_clear (shared with tools/make_golden.py)"""
    stride = 32
    bo = np.array([(i & 1) * 4 + ((i >> 1) & 1) * 4 * stride + ((i >> 2) & 1) * 8 for i in range(48)], np.int32)
    cases = []
    for _ in range(n):
        blocks = rng.integers(-512, 512, 768).astype(np.int16)
        for i in list(range(16, 20)) + list(range(32, 36)):
            if rng.random() < .4:
                blocks[i * 16 + 1:(i + 1) * 16] = 0
            if rng.random() < .2:
                blocks[i * 16] = 0
        cases.append(dict(blocks=blocks, nnzc=rng.integers(0, 2, 120, dtype=np.uint8),
                          cb=rng.integers(0, 256, (12, stride), dtype=np.uint8), cr=rng.integers(0, 256, (12, stride), dtype=np.uint8),
                          dc=rng.integers(-4000, 4000, 16).astype(np.int16), out=rng.integers(-50, 50, 256).astype(np.int16),
                          cdc=rng.integers(-4000, 4000, 64).astype(np.int16), qmul=int(rng.choice([16, 208, 1024, 14000, 65535, -9])),
                          px=rng.integers(0, 256, (8, stride), dtype=np.uint8), res=rng.integers(-400, 400, 64).astype(np.int16)))
    return bo, stride, cases


def h264_misc_apply(L, prefix, bo, stride, k):
    """runs the five members on copies of case k through library L (`ffo` oracle / `ffref` reference); returns the outputs"""
    cb, cr, blocks = k["cb"].copy(), k["cr"].copy(), k["blocks"].copy()
    dp = (ffi.u8p * 2)(ptr(cb), ptr(cr))
    if prefix == "ffo":
        L.ffo_h264_idct_add8(dp, ptr(bo, i32p), ptr(blocks, i16p), stride, ptr(k["nnzc"]))
    else:
        L.ffref_h264_idct_add8(dp, ptr(bo, i32p), ptr(blocks, i16p), stride, ptr(k["nnzc"]), 1)
    out, dc, cdc = k["out"].copy(), k["dc"].copy(), k["cdc"].copy()
    getattr(L, prefix + "_h264_luma_dc_dequant_idct")(ptr(out, i16p), ptr(dc, i16p), k["qmul"])
    getattr(L, prefix + "_h264_chroma_dc_dequant_idct")(ptr(cdc, i16p), k["qmul"])
    res = []
    for n in (4, 8):
        px, r = k["px"].copy(), k["res"][:n * n].copy()
        getattr(L, prefix + "_h264_add_pixels_clear")(n, ptr(px), ptr(r, i16p), stride)
        res += [px, r]
    return [cb, cr, blocks, out, dc, cdc] + res


def test_h264_idct_add8_dc_dequant_add_pixels():
    """the h264dsp members a macroblock needs beside the IDCTs: oracle == reference, bit for bit"""
    R, O = ffi.ref(), ffi.oracle()
    bo, stride, cases = h264_misc_cases(np.random.default_rng(77))
    changed = 0
    for k in cases:
        a, b = h264_misc_apply(R, "ffref", bo, stride, k), h264_misc_apply(O, "ffo", bo, stride, k)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        changed += int((a[0] != k["cb"]).any()) + int((a[3] != k["out"]).any())
    assert changed > len(cases)


# (alpha, beta, tc0) ladder in the spirit of tests/checkasm/h264dsp.c:394-402
LF_PARAMS = [(a, b, t) for a, b, t in [(255, 18, 13), (226, 17, 11), (127, 16, 6), (80, 13, 4), (45, 10, 3),
                                       (28, 8, 1), (17, 6, 1), (9, 3, 0), (4, 2, 0), (0, 0, 0), (255, 18, 0)]]


@pytest.mark.parametrize("which", range(8))
def test_h264_loop_filter(which):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(20 + which)
    for alpha, beta, t in LF_PARAMS:
        for rep in range(12):
            base = rng.integers(0, 256, 1, dtype=np.uint8)[0]
            spread = [2, 6, 20, 255][rep % 4]
            img = np.clip(base.astype(int) + rng.integers(-spread, spread + 1, (32, 32)), 0, 255).astype(np.uint8)
            tc0 = np.array([t, -1 if rep & 1 else t, 0, max(t - 1, 0)], np.int8)
            a, b = img.copy(), img.copy()
            off = 8 * 32 + 8
            pa = C.cast(C.addressof(ptr(a).contents) + off, u8p)
            pb = C.cast(C.addressof(ptr(b).contents) + off, u8p)
            R.ffref_h264_loop_filter(which, pa, 32, alpha, beta, ptr(tc0.copy(), i8p))
            O.ffo_h264_loop_filter(which, pb, 32, alpha, beta, ptr(tc0.copy(), i8p))
            assert np.array_equal(a, b)


@pytest.mark.parametrize("size_idx", [0, 1, 2])
@pytest.mark.parametrize("avg", [0, 1])
def test_h264_qpel(size_idx, avg):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(30 + size_idx * 2 + avg)
    for mcxy in range(16):
        for rep in range(3):
            src = rng.integers(0, 256, (32, 64), dtype=np.uint8)
            if rep == 2:
                src = (rng.integers(0, 2, (32, 64)) * 255).astype(np.uint8)
            dst = rng.integers(0, 256, (32, 64), dtype=np.uint8)
            a, b = dst.copy(), dst.copy()
            so = 6 * 64 + 8
            ps = C.cast(C.addressof(ptr(src).contents) + so, u8p)
            pa = C.cast(C.addressof(ptr(a).contents) + so, u8p)
            pb = C.cast(C.addressof(ptr(b).contents) + so, u8p)
            R.ffref_h264_qpel(avg, size_idx, mcxy, pa, ps, 64)
            O.ffo_h264_qpel(avg, size_idx, mcxy, pb, ps, 64)
            assert np.array_equal(a, b), (mcxy, rep)


@pytest.mark.parametrize("avg", [0, 1])
def test_h264_chroma_mc(avg):
    """tests/checkasm/h264chroma.c shape: put/avg x widths 8/4/2 x all 64 (x, y) eighth-pel positions"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(60 + avg)
    for idx, w in enumerate((8, 4, 2)):
        for h in (2, 4, 8, 16)[:4 if w == 8 else 3]:
            for y in range(8):
                for x in range(8):
                    src = rng.integers(0, 256, (24, 32), dtype=np.uint8)
                    dst = rng.integers(0, 256, (24, 32), dtype=np.uint8)
                    a, b = dst.copy(), dst.copy()
                    off = 2 * 32 + 8
                    R.ffref_h264_chroma(avg, idx, C.cast(a.ctypes.data + off, u8p), C.cast(src.ctypes.data + off, u8p), 32, h, x, y)
                    O.ffo_h264_chroma_mc(avg, w, C.cast(b.ctypes.data + off, u8p), C.cast(src.ctypes.data + off, u8p), 32, h, x, y)
                    assert np.array_equal(a, b), (w, h, x, y)


def test_h264_weight_biweight():
    """tests/checkasm/h264dsp.c has no weight test; the parameter ranges are the slice header's
    (log2_denom 0..7, weights -128..127, offsets -128..127: libavcodec/h264_parse.c ff_h264_pred_weight_table)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(61)
    for idx, w in enumerate((16, 8, 4, 2)):
        for rep in range(60):
            height = int(rng.choice([2, 4, 8, 16]))
            ld = int(rng.integers(0, 8))
            wt, ws, off = [int(v) for v in rng.integers(-128, 128, 3)]
            blk = rng.integers(0, 256, (20, 32), dtype=np.uint8)
            src = rng.integers(0, 256, (20, 32), dtype=np.uint8)
            a, b = blk.copy(), blk.copy()
            R.ffref_h264_weight(idx, C.cast(a.ctypes.data + 40, u8p), 32, height, ld, wt, off)
            O.ffo_h264_weight(w, C.cast(b.ctypes.data + 40, u8p), 32, height, ld, wt, off)
            assert np.array_equal(a, b), ("weight", w, height, ld, wt, off)
            a, b = blk.copy(), blk.copy()
            R.ffref_h264_biweight(idx, C.cast(a.ctypes.data + 40, u8p), C.cast(src.ctypes.data + 40, u8p), 32, height, ld, wt, ws, off)
            O.ffo_h264_biweight(w, C.cast(b.ctypes.data + 40, u8p), C.cast(src.ctypes.data + 40, u8p), 32, height, ld, wt, ws, off)
            assert np.array_equal(a, b), ("biweight", w, height, ld, wt, ws, off)


def fdsp_operands(rng, op, n):
    """(dst, src0, src1, src2, mul) for FFO_FDSP_* op with len n: values across magnitudes, incl. denormals and signed zeros"""
    def vec(k):
        v = (rng.standard_normal(k) * 10.0 ** rng.integers(-6, 7, k)).astype(np.float32)
        v[::17] = 0.0
        v[3::29] = -0.0
        v[5::31] = np.float32(1e-41)
        return v
    dst = vec(2 * n if op == 3 else n)
    src2 = vec(2 * n if op == 3 else n)
    return dst, vec(n), vec(n), src2, float(np.float32(rng.standard_normal() * 3))


def test_float_dsp():
    """libavutil/float_dsp.c vector ops (tests/checkasm/float_dsp.c shapes: len a multiple of 16 plus ragged lengths)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(90)
    for op in range(7):
        for n in (16, 256, 1024, 1000, 4, 1, 37):
            dst, s0, s1, s2, mul = fdsp_operands(rng, op, n)
            a, b = dst.copy(), dst.copy()
            a0, b0 = s0.copy(), s0.copy()
            R.ffref_fdsp(op, ptr(a, f32p), ptr(a0, f32p), ptr(s1, f32p), ptr(s2, f32p), mul, n)
            O.ffo_fdsp(op, ptr(b, f32p), ptr(b0, f32p), ptr(s1, f32p), ptr(s2, f32p), mul, n)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (op, n)
            assert np.array_equal(a0.view(np.uint32), b0.view(np.uint32)), (op, n)


def hevc_coeffs(rng, n, kind):
    """coefficient blocks the way tests/checkasm/hevc_idct.c makes them (random int16 in the decoder's range), plus
    sparse low-frequency blocks (what col_limit is for) and saturating extremes"""
    if kind == 0:
        c = rng.integers(-32768, 32768, (n, n))
    elif kind == 1:
        c = rng.integers(-512, 512, (n, n))
    elif kind == 2:
        c = np.zeros((n, n), np.int64)
        k = int(rng.integers(1, n + 1))
        c[:k, :k] = rng.integers(-2048, 2048, (k, k))
    else:
        c = rng.choice(np.array([-32768, 32767, 0, 1, -1]), (n, n))
    return np.ascontiguousarray(c.astype(np.int16))


def test_hevc_idct():
    """every size x every col_limit the decoder can pass (hevc/cabac.c: 4..2n incl. odd values) + out-of-range limits;
    the reference only reads the coefficients its limits keep, so the blocks are NOT restricted to them"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(70)
    for k in range(32):                                  # our generated matrix == the reference's table, via idct probes
        for i in range(32):
            assert abs(O.ffo_hevc_coef(k, i)) <= 90
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        for col_limit in list(range(0, 2 * n + 6)) + [1000]:
            for kind in range(4):
                c = hevc_coeffs(rng, n, kind)
                a, b = c.copy(), c.copy()
                R.ffref_hevc_idct(lg - 2, ptr(a, ffi.i16p), col_limit)
                O.ffo_hevc_idct(lg, ptr(b, ffi.i16p), col_limit)
                assert np.array_equal(a, b), (n, col_limit, kind)
        for rep in range(20):
            c = hevc_coeffs(rng, n, rep % 4)
            a, b = c.copy(), c.copy()
            R.ffref_hevc_idct_dc(lg - 2, ptr(a, ffi.i16p))
            O.ffo_hevc_idct_dc(lg, ptr(b, ffi.i16p))
            assert np.array_equal(a, b)
            res = hevc_coeffs(rng, n, rep % 4)
            d0 = rng.integers(0, 256, (n + 2, 48), dtype=np.uint8)
            a, b = d0.copy(), d0.copy()
            R.ffref_hevc_add_residual(lg - 2, C.cast(a.ctypes.data + 48 + 3, u8p), ptr(res, ffi.i16p), 48)
            O.ffo_hevc_add_residual(lg, C.cast(b.ctypes.data + 48 + 3, u8p), ptr(res, ffi.i16p), 48)
            assert np.array_equal(a, b)
    for rep in range(200):
        c = hevc_coeffs(rng, 4, rep % 4)
        a, b = c.copy(), c.copy()
        R.ffref_hevc_transform_4x4_luma(ptr(a, ffi.i16p))
        O.ffo_hevc_transform_4x4_luma(ptr(b, ffi.i16p))
        assert np.array_equal(a, b)


def hevc_lf_case(rng, smooth):
    """a 16x16 neighbourhood with an edge in the middle + the arguments of one hevc loop-filter call (ranges as the
    decoder's tables: beta 0..64, tc 0..24, cf. tests/checkasm/hevc_deblock.c)"""
    if smooth:
        base = int(rng.integers(20, 230))
        buf = np.clip(base + rng.integers(-3, 4, (16, 16)) + np.where(np.arange(16)[None, :] >= 8, int(rng.integers(-12, 13)), 0), 0, 255)
        if rng.random() < .5:
            buf = buf.T
    else:
        buf = rng.integers(0, 256, (16, 16))
    beta = int(rng.integers(0, 65))
    tc = rng.integers(0, 25, 2).astype(np.int32)
    no_p, no_q = rng.integers(0, 2, 2).astype(np.uint8) * (rng.random() < .3), rng.integers(0, 2, 2).astype(np.uint8) * (rng.random() < .3)
    return np.ascontiguousarray(buf.astype(np.uint8)), beta, tc, no_p.astype(np.uint8), no_q.astype(np.uint8)


def test_hevc_loop_filter():
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(75)
    changed = 0
    for rep in range(3000):
        buf, beta, tc, no_p, no_q = hevc_lf_case(rng, rep % 4 != 0)
        which = rep % 8
        chroma, vertical = (which >> 1) & 1, which & 1
        a, b = buf.copy(), buf.copy()
        off = 4 * 16 + 8 if vertical else 8 * 16 + 4       # vertical edge: 8 lines down from row 4; horizontal: 8 columns from col 4
        R.ffref_hevc_loop_filter(which, C.cast(a.ctypes.data + off, u8p), 16, beta, ptr(tc, i32p), ptr(no_p), ptr(no_q))
        O.ffo_hevc_loop_filter(chroma, vertical, C.cast(b.ctypes.data + off, u8p), 16, beta, ptr(tc, i32p), ptr(no_p), ptr(no_q))
        assert np.array_equal(a, b), (rep, which, beta, tc, no_p, no_q)
        changed += int((a != buf).any())
    assert changed > 800


HEVC_WIDTHS = [2, 4, 6, 8, 12, 16, 24, 32, 48, 64]


def test_hevc_mc():
    """put_hevc_{qpel,epel}{,_uni}: every fractional position x the 10 width classes (tests/checkasm/hevc_pel.c shapes)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(78)
    src = rng.integers(0, 256, (80, 96), dtype=np.uint8)
    src[:40] = rng.choice(np.array([0, 255], np.uint8), (40, 96))          # extremes: the 14-bit intermediates at their limits
    for chroma in (0, 1):
        nfrac = 8 if chroma else 4
        for w in HEVC_WIDTHS:
            for mx in range(nfrac):
                for my in range(nfrac):
                    h = int(rng.choice([2, 4, 8, 16, 64])) if w > 2 else 2
                    y0 = int(rng.integers(4, 80 - h - 5)); x0 = int(rng.integers(4, 96 - w - 5))
                    sp = C.cast(src.ctypes.data + y0 * 96 + x0, u8p)
                    a16, b16 = np.zeros((64, 64), np.int16), np.zeros((64, 64), np.int16)
                    R.ffref_hevc_mc(chroma, 0, a16.ctypes.data, 0, sp, 96, h, mx, my, w)
                    O.ffo_hevc_mc(chroma, 0, b16.ctypes.data, 0, sp, 96, h, mx, my, w)
                    assert np.array_equal(a16, b16), (chroma, w, mx, my)
                    a8, b8 = np.full((64, 80), 7, np.uint8), np.full((64, 80), 7, np.uint8)
                    R.ffref_hevc_mc(chroma, 1, a8.ctypes.data, 80, sp, 96, h, mx, my, w)
                    O.ffo_hevc_mc(chroma, 1, b8.ctypes.data, 80, sp, 96, h, mx, my, w)
                    assert np.array_equal(a8, b8), (chroma, w, mx, my, "uni")


def hevc_weight_case(rng, rep):
    """(denom, wx0, wx1, ox): the ranges the slice header allows (denom 0..7, w = 2^denom + [-128,127], o in [-128,127]; bi: o0 + o1)
    mixed with tests/checkasm/hevc_pel.c's ladders (denoms 0/7/12, weights 0/128/255, offsets 0/255)"""
    if rep % 3 == 0:
        return int(rng.choice([0, 7, 12])), int(rng.choice([0, 128, 255])), int(rng.choice([0, 128, 255])), int(rng.choice([0, 255]))
    d = int(rng.integers(0, 8))
    return d, (1 << d) + int(rng.integers(-128, 128)), (1 << d) + int(rng.integers(-128, 128)), int(rng.integers(-256, 255))


def test_hevc_mc_weighted():
    """put_hevc_{qpel,epel}_{uni_w,bi,bi_w}: every fractional position x the 10 width classes (tests/checkasm/hevc_pel.c shapes)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(79)
    src = rng.integers(0, 256, (80, 96), dtype=np.uint8)
    src[:40] = rng.choice(np.array([0, 255], np.uint8), (40, 96))
    rep = 0
    for chroma in (0, 1):
        nfrac = 8 if chroma else 4
        for w in HEVC_WIDTHS:
            for mx in range(nfrac):
                for my in range(nfrac):
                    h = int(rng.choice([2, 4, 8, 16, 64])) if w > 2 else 2
                    y0 = int(rng.integers(4, 80 - h - 5)); x0 = int(rng.integers(4, 96 - w - 5))
                    sp = C.cast(src.ctypes.data + y0 * 96 + x0, u8p)
                    # the other list's prediction: a put_hevc_* output, i.e. anything in the 14-bit intermediate range
                    src2 = rng.integers(-8192, 16384, (64, 64)).astype(np.int16) if rep % 4 else np.full((64, 64), 16383 if rep % 8 else -8192, np.int16)
                    for mode in (2, 3, 4):
                        rep += 1
                        d, wx0, wx1, ox = hevc_weight_case(rng, rep)
                        a8, b8 = np.full((64, 80), 7, np.uint8), np.full((64, 80), 7, np.uint8)
                        R.ffref_hevc_mc_w(chroma, mode, ptr(a8), 80, sp, 96, ptr(src2, i16p), h, d, wx0, wx1, ox, mx, my, w)
                        O.ffo_hevc_mc_w(chroma, mode, ptr(b8), 80, sp, 96, ptr(src2, i16p), h, d, wx0, wx1, ox, mx, my, w)
                        assert np.array_equal(a8, b8), (chroma, mode, w, mx, my, d, wx0, wx1, ox)


def vp9_block(rng, n, kind):
    """coefficient blocks of tests/checkasm/vp9dsp.c's spirit (sparse low-frequency content) plus dense, dc-only and
    wrap-around ones: the reference's arithmetic is unsigned 32-bit, so every input is defined"""
    blk = np.zeros(n * n, np.int16)
    if kind == 0:
        blk[:] = rng.integers(-1024, 1025, n * n)
    elif kind == 1:
        blk[:] = rng.integers(-300, 301, n * n) * (rng.random(n * n) < .2)
    elif kind == 2:
        blk[0] = rng.integers(-2000, 2001)
    elif kind == 3:
        blk[:] = rng.integers(-8000, 8001, n * n) * (rng.random(n * n) < .05)
    else:
        blk[:] = rng.integers(-32768, 32768, n * n)
    return blk


def test_vp9_itxfm_add():
    """VP9DSPContext.itxfm_add[5][4]: every size x type, dc-only shortcut, block consumption"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(95)
    for tx in range(5):
        n = 4 if tx == 4 else 4 << tx
        for txtp in range(4):
            for rep in range(40):
                kind = rep % 5
                blk = vp9_block(rng, n, kind)
                eob = 1 if kind == 2 else int(rng.integers(2, n * n + 1))
                dst0 = rng.integers(0, 256, (n, n + 5), dtype=np.uint8)
                a, b, ba, bb = dst0.copy(), dst0.copy(), blk.copy(), blk.copy()
                R.ffref_vp9_itxfm_add(tx, txtp, ptr(a), n + 5, ptr(ba, i16p), eob)
                O.ffo_vp9_itxfm_add(tx, txtp, ptr(b), n + 5, ptr(bb, i16p), eob)
                assert np.array_equal(a, b) and np.array_equal(ba, bb), (tx, txtp, rep)


def test_vp9_mc():
    """VP9DSPContext.mc: 4 filters x put/avg x every (mx, my) class x the 5 widths (tests/checkasm/vp9dsp.c shapes)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(96)
    src = rng.integers(0, 256, (90, 100), dtype=np.uint8)
    src[:30] = rng.choice(np.array([0, 255], np.uint8), (30, 100))          # extremes: both clips fire
    for rep in range(2400):
        f, avg = int(rng.integers(0, 4)), rep & 1
        w = int(rng.choice([4, 8, 16, 32, 64])); h = int(rng.choice([1, 2, 4, 8, 16, 33, 64]))
        mx, my = (int(v) for v in rng.integers(0, 16, 2))
        if rep % 5 == 0:
            mx = 0
        if rep % 7 == 0:
            my = 0
        y0, x0 = int(rng.integers(4, 90 - h - 5)), int(rng.integers(4, 100 - w - 5))
        sp = C.cast(src.ctypes.data + y0 * 100 + x0, u8p)
        d0 = rng.integers(0, 256, (64, 72), dtype=np.uint8)
        a, b = d0.copy(), d0.copy()
        R.ffref_vp9_mc(f, avg, ptr(a), 72, sp, 100, w, h, mx, my)
        O.ffo_vp9_mc(f, avg, ptr(b), 72, sp, 100, w, h, mx, my)
        assert np.array_equal(a, b), (f, avg, w, h, mx, my)


def vp9_lf_plane(rng, n=48):
    """a 48x48 patch around an edge at (24, 24): flat, nearly flat (the flat8 / flat16 decisions sit at |d| <= 1), stepped or noisy on
    either side — tests/checkasm/vp9dsp.c's randomize_loopfilter_buffers in spirit"""
    kind = int(rng.integers(0, 5))
    a, b = int(rng.integers(0, 256)), int(rng.integers(0, 256))
    if kind < 3:
        b = int(np.clip(a + rng.integers(-6, 7), 0, 255))
    p = np.empty((n, n), np.int32)
    p[:, :n // 2] = a
    p[:, n // 2:] = b
    p = p.T.copy() if rng.random() < .5 else p
    noise = [0, 1, 1, 3, 40][kind]
    p = p + rng.integers(-noise, noise + 1, (n, n))
    return np.clip(p, 0, 255).astype(np.uint8)


def test_vp9_loop_filter():
    """loop_filter_8[3][2], loop_filter_16[2] and loop_filter_mix2[2][2][2] against the one-segment oracle"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(97)
    WD = [4, 8, 16]
    for rep in range(3000):
        pl = vp9_lf_plane(rng)
        E, I, H = int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 16))
        if rep % 3 == 0:
            E, I = 255, 63                       # wide open: the flat branches decide
        a, b = pl.copy(), pl.copy()
        dirn = rep & 1
        which = rep % 3
        at_ = lambda arr: C.cast(arr.ctypes.data + 24 * 48 + 24 - (8 if not dirn else 8 * 48) * 0, u8p)
        seg2 = 8 * (48 if not dirn else 1)       # the second segment of the 16-sample forms
        if which == 0:
            w = int(rng.integers(0, 3))
            R.ffref_vp9_loop_filter(0, w, 0, dirn, at_(a), 48, E, I, H)
            O.ffo_vp9_loop_filter(WD[w], dirn, at_(b), 48, E, I, H)
        elif which == 1:
            R.ffref_vp9_loop_filter(1, 0, 0, dirn, at_(a), 48, E, I, H)
            O.ffo_vp9_loop_filter(16, dirn, at_(b), 48, E, I, H)
            O.ffo_vp9_loop_filter(16, dirn, C.cast(b.ctypes.data + 24 * 48 + 24 + seg2, u8p), 48, E, I, H)
        else:
            w1, w2 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            E2, I2, H2 = int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 16))
            R.ffref_vp9_loop_filter(2, w1, w2, dirn, at_(a), 48, E | E2 << 8, I | I2 << 8, H | H2 << 8)
            O.ffo_vp9_loop_filter(WD[w1], dirn, at_(b), 48, E, I, H)
            O.ffo_vp9_loop_filter(WD[w2], dirn, C.cast(b.ctypes.data + 24 * 48 + 24 + seg2, u8p), 48, E2, I2, H2)
        assert np.array_equal(a, b), (rep, which, dirn, E, I, H)


def test_vp9_intra_pred():
    """VP9DSPContext.intra_pred[4][15]: every size x mode (tests/checkasm/vp9dsp.c check_ipred shapes: aligned top with a corner
    in front and top-right samples behind)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(98)
    for tx in range(4):
        n = 4 << tx
        for mode in range(15):
            for rep in range(12):
                left = rng.integers(0, 256, n + 16, dtype=np.uint8)
                topbuf = rng.integers(0, 256, 16 + 2 * n + 32, dtype=np.uint8)
                if rep % 3 == 0:
                    left[:] = rng.choice(np.array([0, 255], np.uint8), left.size); topbuf[:] = rng.choice(np.array([0, 255], np.uint8), topbuf.size)
                tp = C.cast(topbuf.ctypes.data + 16, u8p)
                a = rng.integers(0, 256, (n, n + 3), dtype=np.uint8)
                b = a.copy()
                R.ffref_vp9_intra_pred(tx, mode, ptr(a), n + 3, ptr(left), tp)
                O.ffo_vp9_intra_pred(tx, mode, ptr(b), n + 3, ptr(left), tp)
                assert np.array_equal(a, b), (tx, mode, rep)


def h264_pred_plane(rng, rep):
    """a 48x48 patch with the block at (16, 16): random, saturated, or smooth (the plane predictor's clip on both sides)"""
    if rep % 4 == 1:
        return rng.choice(np.array([0, 255], np.uint8), (48, 48))
    if rep % 4 == 2:
        g = np.add.outer(np.arange(48) * int(rng.integers(-9, 10)), np.arange(48) * int(rng.integers(-9, 10))) + int(rng.integers(0, 256))
        return np.clip(g + rng.integers(-3, 4, (48, 48)), 0, 255).astype(np.uint8)
    return rng.integers(0, 256, (48, 48), dtype=np.uint8)


#: H264PredContext members by size: (name, number of modes filled for the H.264 codec at 8 bits, 4:2:0) - h264pred.c:448-538
H264_PRED_SETS = (("pred4x4", 12), ("pred8x8l", 12), ("pred8x8", 11), ("pred16x16", 7))


def h264_pred_call(L, pref, name, mode, buf, tr, tl_tr):
    at = C.cast(buf.ctypes.data + 16 * 48 + 16, u8p)
    if name == "pred4x4":
        getattr(L, pref + "_h264_pred4x4")(mode, at, ptr(tr), 48)
    elif name == "pred8x8l":
        getattr(L, pref + "_h264_pred8x8l")(mode, at, tl_tr[0], tl_tr[1], 48)
    else:
        getattr(L, pref + "_h264_" + name)(mode, at, 48)


def test_h264_pred():
    """H264PredContext.pred4x4 / pred8x8l / pred8x8 / pred16x16, every member the H.264 decoder gets (tests/checkasm/h264pred.c
    walks the same tables); pred4x4's topright is a separate pointer as in the decoder (h264_mb_template.c passes either the row
    above or a replicated sample)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(2640)
    for name, nmodes in H264_PRED_SETS:
        for mode in range(nmodes):
            for rep in range(16):
                a = h264_pred_plane(rng, rep)
                b = a.copy()
                tr = rng.integers(0, 256, 4, dtype=np.uint8) if rep & 1 else a[15, 20:24].copy()
                # the diagonal modes that read the corner are only called with it available
                tl_tr = (1 if mode in (4, 5, 6) else int(rng.integers(0, 2)), int(rng.integers(0, 2)))
                h264_pred_call(R, "ffref", name, mode, a, tr, tl_tr)
                h264_pred_call(O, "ffo", name, mode, b, tr, tl_tr)
                assert np.array_equal(a, b), (name, mode, rep, tl_tr)


def test_h264_pred_add():
    """the lossless members: pred4x4_add / pred8x8l_add / pred8x8l_filter_add [VERT, HOR], pred8x8_add / pred16x16_add
    [VERT_PRED8x8, HOR_PRED8x8] with the decoder's block_offset tables (h264_slice.c init_scan_tables / h264dec block_offset)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(2641)
    scan = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (3, 0), (2, 1), (3, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 2), (3, 2), (2, 3), (3, 3)]
    for rep in range(24):
        lim = 300 if rep % 3 else 32767
        for name, n in (("pred4x4_add", 4), ("pred8x8l_add", 8), ("pred8x8l_filter_add", 8)):
            for mode in (0, 1):
                a = h264_pred_plane(rng, rep); b = a.copy()
                ca = rng.integers(-lim, lim + 1, n * n).astype(np.int16); cb = ca.copy()
                at = lambda x: C.cast(x.ctypes.data + 16 * 48 + 16, u8p)
                if name == "pred8x8l_filter_add":
                    tl, tr = int(rng.integers(0, 2)), int(rng.integers(0, 2))
                    R.ffref_h264_pred8x8l_filter_add(mode, at(a), ptr(ca, i16p), tl, tr, 48)
                    O.ffo_h264_pred8x8l_filter_add(mode, at(b), ptr(cb, i16p), tl, tr, 48)
                else:
                    getattr(R, "ffref_h264_" + name)(mode, at(a), ptr(ca, i16p), 48)
                    getattr(O, "ffo_h264_" + name)(mode, at(b), ptr(cb, i16p), 48)
                assert np.array_equal(a, b) and np.array_equal(ca, cb) and not ca.any(), (name, mode, rep)
        for name, nb in (("pred8x8_add", 4), ("pred16x16_add", 16)):
            for mode in (2, 1):
                a = h264_pred_plane(rng, rep); b = a.copy()
                ca = rng.integers(-lim, lim + 1, nb * 16).astype(np.int16); cb = ca.copy()
                offs = np.array([4 * x + 4 * y * 48 for x, y in scan[:nb]], np.int32)
                at = lambda x: C.cast(x.ctypes.data + 16 * 48 + 16, u8p)
                getattr(R, "ffref_h264_" + name)(mode, at(a), ptr(offs, i32p), ptr(ca, i16p), 48)
                getattr(O, "ffo_h264_" + name)(mode, at(b), ptr(offs, i32p), ptr(cb, i16p), 48)
                assert np.array_equal(a, b) and np.array_equal(ca, cb), (name, mode, rep)


def test_h264_pred_422():
    """chroma_format_idc 2: pred8x8[] / pred8x8_add[] are the 8 wide x 16 tall forms (h264pred.c:478-512, :534-535;
    h264pred_template.c:477-817, :1302-1330) - the restatement's ffo_h264_pred8x16 / _add == the reference's context
    initialised for 4:2:2, every mode"""
    R, O = ffi.ref(), ffi.oracle()
    R.ffref_h264_pred_set_format.argtypes = [C.c_int, C.c_int]
    O.ffo_h264_pred8x16.argtypes = [C.c_int, u8p, C.c_ssize_t]
    O.ffo_h264_pred8x16_add.argtypes = [C.c_int, u8p, i32p, i16p, C.c_ssize_t]
    rng = np.random.default_rng(2642)
    at = lambda x: C.cast(x.ctypes.data + 16 * 48 + 16, u8p)
    R.ffref_h264_pred_set_format(8, 2)
    try:
        for mode in range(11):
            for rep in range(16):
                a = h264_pred_plane(rng, rep); b = a.copy(); p0 = a.copy()
                R.ffref_h264_pred8x8(mode, at(a), 48)
                O.ffo_h264_pred8x16(mode, at(b), 48)
                assert np.array_equal(a, b), (mode, rep)
                keep = np.ones((48, 48), bool); keep[16:32, 16:24] = False
                assert np.array_equal(a[keep], p0[keep]), "nothing but the 8 x 16 block is written"
        offs = np.zeros(16, np.int32)
        for i in range(4):
            offs[i] = 4 * (i & 1) + 4 * (i >> 1) * 48
            offs[8 + i] = 4 * (i & 1) + 4 * (2 + (i >> 1)) * 48
        for rep in range(24):
            lim = 300 if rep % 3 else 32767
            for mode in (2, 1):
                a = h264_pred_plane(rng, rep); b = a.copy()
                ca = rng.integers(-lim, lim + 1, 8 * 16).astype(np.int16); cb = ca.copy()
                R.ffref_h264_pred8x8_add(mode, at(a), ptr(offs, i32p), ptr(ca, i16p), 48)
                O.ffo_h264_pred8x16_add(mode, at(b), ptr(offs, i32p), ptr(cb, i16p), 48)
                assert np.array_equal(a, b) and np.array_equal(ca, cb), (mode, rep)
    finally:
        R.ffref_h264_pred_set_format(8, 1)


#: the batch kinds of include/ffhip.h FFHIP_H264_PRED*: (block size, number of modes, member name)
H264_PRED_KINDS = ((4, 12, "pred4x4"), (8, 12, "pred8x8l"), (8, 11, "pred8x8"), (16, 7, "pred16x16"),
                   (4, 2, "pred4x4_add"), (8, 2, "pred8x8l_add"), (8, 2, "pred8x8l_filter_add"))
#: FFHIP_H264_PRED8x16 (kind 7): pred8x8[] at chroma_format_idc 2 - 8 wide, 16 tall; kept out of the tuple above, which the
#: committed h264pred.npz enumerates
H264_PRED_KIND_422 = (8, 11, "pred8x16")


def h264_pred_kind(kind):
    """(width, height, number of modes, member name) of a batch kind"""
    n, nmodes, name = H264_PRED_KINDS[kind] if kind < 7 else H264_PRED_KIND_422
    return n, 16 if kind == 7 else n, nmodes, name


def h264_pred_grid(rng, kind, height, width, count=None):
    """Independent blocks of one kind on a grid with gaps (3N across for the top-right run, 2N down), so that no block's
    neighbours are another block's output: rows of (x, y, mode, flags, aux).  flags / aux as FFHipH264Pred: pred8x8l's
    has_topleft (1) / has_topright (2); pred4x4's topright either at aux (bytes into the plane, stride = width) or replicated
    (flag 4); the _add kinds' aux = coefficient index."""
    n, nh, nmodes, _ = h264_pred_kind(kind)
    recs = []
    i = 0
    for y in range(n, height - nh + 1, nh + n):
        for x in range(n, width - 2 * n + 1, 3 * n):
            mode = i % nmodes
            flags, aux = 0, 0
            if kind in (1, 6):
                flags = (i // nmodes) & 3
                if kind == 1 and mode in (4, 5, 6):
                    flags |= 1                          # the corner modes are only called with it available
            elif kind == 0:
                sel = (i // nmodes) % 3
                if sel == 0:
                    aux = (y - 1) * width + x + 4       # the row above, as the decoder passes it
                elif sel == 1:
                    flags = 4
                else:
                    aux = int(rng.integers(0, n - 1)) * width + int(rng.integers(0, width - 4))  # anywhere (the top gap rows)
            if kind >= 4:
                aux = i * n * n
            recs.append((x, y, mode, flags, aux))
            i += 1
    return np.array(recs[:count] if count else recs, np.int32)


def h264_pred_apply(L, pref, kind, pic, recs, coeffs=None):
    """run the reference or the oracle over the records, in place on pic (and coeffs)"""
    n, _, _, name = h264_pred_kind(kind)
    width = pic.shape[1]
    if kind == 7 and pref == "ffref":       # the reference's pred8x8[] after ffref_h264_pred_set_format(8, 2)
        name = "pred8x8"
    fn = getattr(L, "%s_h264_%s" % (pref, name))
    for x, y, mode, flags, aux in recs.tolist():
        at = C.cast(pic.ctypes.data + y * width + x, u8p)
        if kind == 0:
            tr = np.full(4, pic[y - 1, x + 3], np.uint8) if flags & 4 else pic.reshape(-1)[aux:aux + 4].copy()
            fn(mode, at, ptr(tr), width)
        elif kind == 1:
            fn(mode, at, flags & 1, (flags >> 1) & 1, width)
        elif kind in (2, 3, 7):
            fn(mode, at, width)
        else:
            blk = C.cast(coeffs.ctypes.data + 2 * aux, i16p)
            if kind == 6:
                fn(mode, at, blk, flags & 1, (flags >> 1) & 1, width)
            else:
                fn(mode, at, blk, width)


def vp9_smc_case(rng):
    """(filter, avg, w, h, mx, my, dx, dy): steps from 16x up-scaling (1) to 2x down-scaling (32) of the reference"""
    w = int(rng.choice([4, 8, 16, 32, 64])); h = int(rng.choice([1, 2, 4, 8, 16, 32, 64]))
    return (int(rng.integers(0, 4)), int(rng.integers(0, 2)), w, h, int(rng.integers(0, 16)), int(rng.integers(0, 16)),
            int(rng.choice([1, 8, 11, 16, 20, 27, 32])), int(rng.choice([1, 8, 11, 16, 20, 27, 32])))


def test_vp9_smc():
    """VP9DSPContext.smc: scaled motion compensation, all filters, put / avg"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(99)
    src = rng.integers(0, 256, (160, 160), dtype=np.uint8)
    src[:40] = rng.choice(np.array([0, 255], np.uint8), (40, 160))
    for rep in range(800):
        f, avg, w, h, mx, my, dx, dy = vp9_smc_case(rng)
        y0, x0 = int(rng.integers(4, 16)), int(rng.integers(4, 16))
        sp = C.cast(src.ctypes.data + y0 * 160 + x0, u8p)
        d0 = rng.integers(0, 256, (64, 72), dtype=np.uint8)
        a, b = d0.copy(), d0.copy()
        R.ffref_vp9_smc(f, avg, ptr(a), 72, sp, 160, w, h, mx, my, dx, dy)
        O.ffo_vp9_smc(f, avg, ptr(b), 72, sp, 160, w, h, mx, my, dx, dy)
        assert np.array_equal(a, b), (f, avg, w, h, mx, my, dx, dy)


def hevc_restore_case(rng, rep):
    """(variant, eo, offset0, borders[4], width, height, vert_edge[2], horiz_edge[2], diag_edge[4]) — every flag on and off"""
    p = .5 if rep % 3 else .85
    return (rep & 1, int(rng.integers(0, 4)), int(rng.integers(-60, 61)), (rng.random(4) < p).astype(np.int32),
            int(rng.choice([2, 3, 8, 16, 33, 64])), int(rng.choice([2, 3, 8, 16, 33, 64])), (rng.random(2) < p).astype(np.uint8),
            (rng.random(2) < p).astype(np.uint8), (rng.random(4) < p).astype(np.uint8))


def test_hevc_small_members():
    """dequant, transform_rdpcm and sao_edge_restore[2] (hevc/dsp_template.c:85-143, h26x/h2656_sao_template.c:81-214)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(90)
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        for rep in range(6):
            c = rng.integers(-32768, 32768, n * n).astype(np.int16) if rep % 2 else rng.integers(-300, 301, n * n).astype(np.int16)
            a, b = c.copy(), c.copy()
            R.ffref_hevc_dequant(ptr(a, i16p), lg); O.ffo_hevc_dequant(ptr(b, i16p), lg)
            assert np.array_equal(a, b), ("dequant", lg)
            for mode in (0, 1):
                a, b = c.copy(), c.copy()
                R.ffref_hevc_transform_rdpcm(ptr(a, i16p), lg, mode); O.ffo_hevc_transform_rdpcm(ptr(b, i16p), lg, mode)
                assert np.array_equal(a, b), ("rdpcm", lg, mode)
    for rep in range(400):
        var, eo, off, borders, w, h, ve, he, de = hevc_restore_case(rng, rep)
        src = rng.integers(0, 256, (h, 80), dtype=np.uint8)
        dst0 = rng.integers(0, 256, (h, 72), dtype=np.uint8)
        a, b = dst0.copy(), dst0.copy()
        R.ffref_hevc_sao_edge_restore(var, ptr(a), ptr(src), 72, 80, eo, off, ptr(borders, i32p), w, h, ptr(ve), ptr(he), ptr(de))
        O.ffo_hevc_sao_edge_restore(var, ptr(b), ptr(src), 72, 80, eo, off, ptr(borders, i32p), w, h, ptr(ve), ptr(he), ptr(de))
        assert np.array_equal(a, b), ("restore", rep)


def test_hevc_sao():
    """band and edge offsets on CTB-sized blocks (tests/checkasm/hevc_sao.c shapes: widths 8..64, the padded 192-byte source)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(76)
    for rep in range(120):
        w = int(rng.choice([8, 16, 32, 48, 64])); h = int(rng.choice([8, 16, 32, 64]))
        idx = {8: 0, 16: 1, 32: 2, 48: 3, 64: 4}[w]
        src = rng.integers(0, 256, (h + 2, 192), dtype=np.uint8)
        if rep % 3 == 0:
            src[:] = np.clip(128 + rng.integers(-6, 7, src.shape), 0, 255)
        off = rng.integers(-7, 8, 5).astype(np.int16) * (1 if rep % 5 else 4)
        dst0 = rng.integers(0, 256, (h, 80), dtype=np.uint8)
        a, b = dst0.copy(), dst0.copy()
        left = int(rng.integers(0, 32))
        R.ffref_hevc_sao_band(idx, ptr(a), C.cast(src.ctypes.data + 192 + 1, u8p), 80, 192, ptr(off, i16p), left, w, h)
        O.ffo_hevc_sao_band(ptr(b), C.cast(src.ctypes.data + 192 + 1, u8p), 80, 192, ptr(off, i16p), left, w, h)
        assert np.array_equal(a, b), ("band", rep)
        off[0] = 0
        for eo in range(4):
            a, b = dst0.copy(), dst0.copy()
            R.ffref_hevc_sao_edge(idx, ptr(a), C.cast(src.ctypes.data + 192 + 1, u8p), 80, ptr(off, i16p), eo, w, h)
            O.ffo_hevc_sao_edge(ptr(b), C.cast(src.ctypes.data + 192 + 1, u8p), 80, 192, ptr(off, i16p), eo, w, h)
            assert np.array_equal(a, b), ("edge", rep, eo)


def test_me_cmp():
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(40)
    a = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    b = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    for _ in range(64):
        x, y = rng.integers(0, 40, 2)
        pa = C.cast(C.addressof(ptr(a).contents) + int(y) * 64 + int(x & ~15), u8p)
        pb = C.cast(C.addressof(ptr(b).contents) + int(y) * 64 + int(x), u8p)
        for h in (8, 16):
            assert R.ffref_me_cmp(0, 0, pa, pb, 64, h) == O.ffo_sad(16, pa, pb, 64, h)
            assert R.ffref_me_cmp(0, 1, pa, pb, 64, h) == O.ffo_sad(8, pa, pb, 64, h)
            assert R.ffref_me_cmp(1, 0, pa, pb, 64, h) == O.ffo_hadamard8_diff16(pa, pb, 64, h)
        assert R.ffref_me_cmp(1, 1, pa, pb, 64, 8) == O.ffo_hadamard8_diff8x8(pa, pb, 64)
        # the half-pel SADs, SSE and NSSE (ffref kinds 3 / 4 / 5 = pix_abs[w][1..3], 2 = sse, 6 = nsse; oracle kinds FFHIP_ME_*)
        O.ffo_me_cmp_other.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int]
        for h in (4, 8, 16):
            for idx, w in ((0, 16), (1, 8)):
                for rk, ok in ((3, 2), (4, 3), (5, 4), (2, 5), (6, 6)):
                    assert R.ffref_me_cmp(rk, idx, pa, pb, 64, h) == O.ffo_me_cmp_other(ok, w, pa, pb, 64, h), (rk, w, h)


@pytest.mark.parametrize("R_", [3, 7])
def test_me_search_esa(R_):
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(50 + R_)
    w, h = 96, 64
    ref_img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    cur = np.roll(ref_img, (2, -3), (0, 1)).copy()
    cur[:16, :16] = ref_img[:16, :16]               # a zero-cost zero-MV block (early return)
    cur[20:, 40:] = rng.integers(0, 4, cur[20:, 40:].shape, dtype=np.uint8)   # many ties
    for by in range(h // 16):
        for bx in range(w // 16):
            m1 = np.zeros(2, np.int32); m2 = np.array([bx * 16, by * 16], np.int32)
            c1 = R.ffref_me_search_esa(ptr(cur), ptr(ref_img), w, w, h, 16, R_, bx * 16, by * 16, ptr(m1, i32p))
            c2 = O.ffo_me_search_esa(ptr(cur), ptr(ref_img), w, w, h, 16, R_, 0, bx * 16, by * 16, ptr(m2, i32p))
            assert c1 == c2 and np.array_equal(m1, m2)


@pytest.mark.parametrize("len_", [16, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 120, 240, 480, 960, 1920])
@pytest.mark.parametrize("inv", [0, 1])
def test_mdct_float(len_, inv):
    """The restated split-radix recursion (and, for 2 * 15 * 2^k, the 15xM prime-factor codelet: Opus / AAC-960 sizes) reproduces
    the reference's float results exactly."""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(len_ + inv)
    scale = 1.0 / len_ if inv else 1.0
    rc = R.ffref_tx_create(1, inv, len_, scale, 0)
    oc = O.ffo_mdct_create(inv, len_, scale)
    assert rc and oc
    for _ in range(4):
        x = rng.uniform(-1, 1, 2 * len_ if not inv else len_).astype(np.float32)
        a = np.zeros(len_, np.float32); b = np.zeros(len_, np.float32)
        R.ffref_tx_run(rc, ptr(a, f32p), ptr(x.copy(), f32p), 4)
        O.ffo_mdct_run(oc, ptr(b, f32p), ptr(x, f32p), 4)
        assert np.array_equal(a, b), np.abs(a - b).max()
    R.ffref_tx_free(rc); O.ffo_mdct_free(oc)


PFA_LENS = [2 * f * m for f in (3, 5, 7, 9) for m in (4, 8, 16, 32, 64, 128, 256)]           # ff_tx_mdct_pfa_{3,5,7,9}xM


@pytest.mark.parametrize("len_", PFA_LENS + [2 * 3 * 2, 2 * 9 * 512, 2 * 5 * 1024])
@pytest.mark.parametrize("inv", [0, 1])
def test_mdct_float_pfa_3579(len_, inv):
    """DECL_COMP_MDCT(3 / 5 / 7 / 9) (libavutil/tx_template.c:1595-1598; fft7 / fft9 :250-461): the codelet av_tx_init picks for
    len / 2 = N * 2^k is the one the oracle restates (96 / 768-sample AAC frames = 3 x 32 / 3 x 256, Siren's 320 = 5 x 64, ...),
    negative scales included - bit-identical"""
    R, O = ffi.ref(), ffi.oracle()
    ffi_int = O.ffo_mdct_pfa_factor(len_)
    assert ffi_int in (3, 5, 7, 9)
    rng = np.random.default_rng(7 * len_ + inv)
    for scale in (1.0 / len_ if inv else 1.0, -0.013):
        rc = R.ffref_tx_create(1, inv, len_, scale, 0)
        oc = O.ffo_mdct_create(inv, len_, scale)
        assert rc and oc
        for _ in range(3):
            x = (rng.uniform(-1, 1, 2 * len_ if not inv else len_) * 10.0 ** float(rng.integers(-2, 3))).astype(np.float32)
            a = np.zeros(len_, np.float32); b = np.zeros(len_, np.float32)
            R.ffref_tx_run(rc, ptr(a, f32p), ptr(x.copy(), f32p), 4)
            O.ffo_mdct_run(oc, ptr(b, f32p), ptr(x, f32p), 4)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (scale, np.abs(a - b).max())
        R.ffref_tx_free(rc); O.ffo_mdct_free(oc)


@pytest.mark.parametrize("inv", [0, 1])
@pytest.mark.parametrize("len_", [4, 8, 16, 64, 256, 1024, 2048, 4096, 8192, 16384] +
                         [f * m for f in (3, 5, 7, 9) for m in (4, 16, 64, 128, 256)] + [15 * m for m in (4, 8, 16, 32, 64, 128)])
def test_fft_float(len_, inv):
    """AV_TX_FLOAT_FFT, both directions: bit-identical (tests/checkasm/av_tx.c compares to 5e-4 only).  Powers of two (the split-radix
    codelets) and F * 2^k, F = 3 / 5 / 7 / 9 / 15: the ff_tx_fft_pfa tree av_tx_init builds for them (120 / 960 / 1920 = fft15_ns x
    fft8 / 64 / 128_ns)"""
    R, O = ffi.ref(), ffi.oracle()
    rc = R.ffref_tx_create(0, inv, len_, 1.0, 0)
    assert rc
    rng = np.random.default_rng(len_ + inv)
    for rep in range(4):
        x = (rng.standard_normal(2 * len_) * 10.0 ** float(rng.integers(-3, 4))).astype(np.float32)
        a, b = np.zeros(2 * len_, np.float32), np.zeros(2 * len_, np.float32)
        xi = x.copy()
        R.ffref_tx_run(rc, ptr(a, f32p), ptr(xi, f32p), 8)
        O.ffo_fft_run(inv, len_, ptr(b, f32p), ptr(x, f32p))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    R.ffref_tx_free(rc)


@pytest.mark.parametrize("len_", [16, 256, 1024, 120, 960])
def test_imdct_full(len_):
    """AV_TX_FULL_IMDCT (flag 1 << 2): the half inverse mirrored to 2 * len outputs"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(len_)
    scale = 1.0 / len_
    rc = R.ffref_tx_create(1, 1, len_, scale, 4)
    oc = O.ffo_mdct_create(1, len_, scale)
    assert rc and oc
    for _ in range(3):
        x = rng.uniform(-1, 1, len_).astype(np.float32)
        a, b = np.zeros(2 * len_, np.float32), np.zeros(2 * len_, np.float32)
        R.ffref_tx_run(rc, ptr(a, f32p), ptr(x.copy(), f32p), 4)
        O.ffo_imdct_full_run(oc, ptr(b, f32p), ptr(x, f32p))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    R.ffref_tx_free(rc); O.ffo_mdct_free(oc)


@pytest.mark.parametrize("inv", [0, 1])
@pytest.mark.parametrize("len_", [8, 16, 64, 512, 1024, 4096])
def test_rdft_float(len_, inv):
    """AV_TX_FLOAT_RDFT (r2c / c2r), power-of-two: bit-identical, with and without a scale"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(3 * len_ + inv)
    for scale in (1.0, 1.0 / len_, -0.37):
        rc = R.ffref_tx_create(6, inv, len_, scale, 0)      # AV_TX_FLOAT_RDFT = 6 (libavutil/tx.h:90)
        assert rc
        for rep in range(3):
            x = (rng.standard_normal(len_ + 2 if inv else len_) * 10.0 ** float(rng.integers(-3, 4))).astype(np.float32)
            if inv:
                x[1] = x[-1] = 0                              # the imaginary parts of the DC and Nyquist bins
            a, b = np.zeros(len_ if inv else len_ + 2, np.float32), np.zeros(len_ if inv else len_ + 2, np.float32)
            R.ffref_tx_run(rc, ptr(a, f32p), ptr(x.copy(), f32p), 4)
            O.ffo_rdft_run(inv, len_, scale, ptr(b, f32p), ptr(x, f32p))
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (scale, rep)
        R.ffref_tx_free(rc)


@pytest.mark.parametrize("n", [8, 16, 64, 256, 1024, 4096])
@pytest.mark.parametrize("inv", [0, 1])
def test_dct_float(n, inv):
    """AV_TX_FLOAT_DCT (DCT-II forward / DCT-III inverse) of n real samples, power-of-two: bit-identical, including the forward
    transform's running sum.  av_tx_init is handed n resp. n / 2 (ff_tx_dct_init doubles the inverse's length); the DCT-III reads
    two samples of padding behind its input and both clobber it (libavutil/tx.h:95-103)"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(5 * n + inv)
    for scale in (1.0, 1.0 / n, -0.37):
        rc = R.ffref_tx_create(9, inv, n >> inv, scale, 0)      # AV_TX_FLOAT_DCT = 9 (libavutil/tx.h:104)
        assert rc
        for rep in range(3):
            x = np.zeros(n + 2, np.float32)
            x[:n] = (rng.standard_normal(n) * 10.0 ** float(rng.integers(-3, 4))).astype(np.float32)
            a, b = np.zeros(n + 2, np.float32), np.zeros(n + 2, np.float32)
            R.ffref_tx_run(rc, ptr(a, f32p), ptr(x.copy(), f32p), 4)
            O.ffo_dct_run(inv, n, scale, ptr(b, f32p), ptr(x, f32p))
            assert np.array_equal(a[:n].view(np.uint32), b[:n].view(np.uint32)), (scale, rep)
        R.ffref_tx_free(rc)


@pytest.mark.parametrize("mode", [1, 2], ids=["r2r", "r2i"])
@pytest.mark.parametrize("len_", [8, 16, 64, 512, 1024, 4096])
def test_rdft_half_float(len_, mode):
    """AV_TX_FLOAT_RDFT with AV_TX_REAL_TO_REAL / AV_TX_REAL_TO_IMAGINARY (ff_tx_rdft_r2r / _r2i, forward only): len/2 + 1 real
    resp. len/2 imaginary parts, bit-identical — including r2i's last value, which the reference leaves as the FFT produced it"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(7 * len_ + mode)
    nout = len_ // 2 + (mode == 1)
    for scale in (1.0, 1.0 / len_, -0.37):
        rc = R.ffref_tx_create(6, 0, len_, scale, 1 << (2 + mode))   # AV_TX_REAL_TO_REAL = 1 << 3, _IMAGINARY = 1 << 4 (tx.h:184-185)
        assert rc
        assert not R.ffref_tx_create(6, 1, len_, scale, 1 << (2 + mode))   # FF_TX_FORWARD_ONLY
        for rep in range(3):
            x = (rng.standard_normal(len_) * 10.0 ** float(rng.integers(-3, 4))).astype(np.float32)
            a, b = np.zeros(len_ + 2, np.float32), np.zeros(len_ + 2, np.float32)   # the reference's FFT lands in dst first
            R.ffref_tx_run(rc, ptr(a, f32p), ptr(x.copy(), f32p), 4)
            O.ffo_rdft_half_run(mode, len_, scale, ptr(b, f32p), ptr(x, f32p))
            assert np.array_equal(a[:nout].view(np.uint32), b[:nout].view(np.uint32)), (scale, rep)
        R.ffref_tx_free(rc)


def test_dct_vs_definition():
    """the float DCT-II against the double-precision cosine sum it implements (av_tx's scaling: X[k] = 2 sum x[j] cos(pi (2j + 1)
    k / 2n)), and the DCT-III as its inverse up to a constant factor"""
    O = ffi.oracle()
    n = 256
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, n).astype(np.float32)
    j = np.arange(n)
    want = 2 * np.array([(x.astype(np.float64) * np.cos(np.pi * (2 * j + 1) * k / (2 * n))).sum() for k in range(n)])
    out = np.zeros(n + 2, np.float32)
    O.ffo_dct_run(0, n, 1.0, ptr(out, f32p), ptr(x, f32p))
    assert np.abs(out[:n] - want).max() <= 2.0 ** -18 * np.abs(want).max() * 8
    back = np.zeros(n + 2, np.float32)
    src = np.zeros(n + 2, np.float32); src[:n] = out[:n]
    O.ffo_dct_run(1, n, 1.0, ptr(back, f32p), ptr(src, f32p))
    ratio = back[:n] / x
    assert np.allclose(ratio, ratio[0], rtol=1e-3), ratio[:4]


AAC_WINDOWS = ((0, 1024), (1, 128), (2, 1024), (3, 128))       # sine_1024, sine_128, kbd_long_1024, kbd_short_128
AAC_SCALES = (2.0 ** -25, 2.0 ** -22)                            # MDCT_INIT's scale_float for 1024 / 128 (aacdec.c:1267-1285)


def aac_ref_windows():
    R = ffi.ref()
    return [np.ctypeslib.as_array(R.ffref_aac_window(w), (n,)).copy() for w, n in AAC_WINDOWS]


def aac_sequences(rng, nframes):
    """a legal window-sequence walk (ISO 14496-3 4.5.2.3.3 transitions) with window shapes switching at random"""
    nxt = {0: (0, 0, 0, 1), 1: (2,), 2: (2, 2, 3), 3: (0, 0, 1)}
    seq, kb = [0], [int(rng.integers(0, 2))]
    for _ in range(nframes - 1):
        seq.append(int(rng.choice(nxt[seq[-1]])))
        kb.append(int(rng.integers(0, 2)) if rng.integers(0, 4) == 0 else kb[-1])
    return np.array(seq, np.int32), np.array(kb, np.int32)


def aac_oracle_run(O, windows, coeffs, seq, kb, saved):
    """frames of one channel through the oracle: coeffs [nf, 1024]; returns out [nf, 1024], saved updated in place"""
    m1024, m128 = O.ffo_mdct_create(1, 1024, AAC_SCALES[0]), O.ffo_mdct_create(1, 128, AAC_SCALES[1])
    wp = (f32p * 4)(*[ptr(w, f32p) for w in windows])
    out = np.zeros_like(coeffs)
    prev = (0, int(kb[0]))
    for f in range(len(coeffs)):
        s2 = np.array([seq[f], prev[0]], np.int32); k2 = np.array([kb[f], prev[1]], np.int32)
        O.ffo_aac_imdct_and_windowing(m1024, m128, wp, ptr(np.ascontiguousarray(coeffs[f]), f32p), ptr(s2, i32p), ptr(k2, i32p),
                                      ptr(saved, f32p), ptr(out[f], f32p))
        prev = (int(seq[f]), int(kb[f]))
    O.ffo_mdct_free(m1024); O.ffo_mdct_free(m128)
    return out


def test_aac_windows():
    """the sine tables are bit-identical (same libm), the Kaiser-Bessel ones within 1 ulp (another I0 evaluation)"""
    O = ffi.oracle()
    ref = aac_ref_windows()
    for (w, n), r in zip(AAC_WINDOWS, ref):
        mine = np.zeros(n, np.float32)
        if w < 2:
            O.ffo_aac_sine_window(ptr(mine, f32p), n)
            assert np.array_equal(mine.view(np.uint32), r.view(np.uint32)), w
        else:
            O.ffo_aac_kbd_window(ptr(mine, f32p), 4.0 if n == 1024 else 6.0, n)
            assert np.abs(mine.view(np.int32).astype(np.int64) - r.view(np.int32)).max() <= 1, w


def test_aac_imdct_and_windowing():
    """AACDecDSP.imdct_and_windowing of the float decoder, frame after frame with the overlap state carried along: every window
    sequence transition and shape switch, bit-identical"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(2700)
    windows = aac_ref_windows()
    nf = 120
    seq, kb = aac_sequences(rng, nf)
    assert set(seq.tolist()) == {0, 1, 2, 3}
    coeffs = (rng.standard_normal((nf, 1024)) * 3000.0 * 10.0 ** rng.integers(-2, 2, (nf, 1))).astype(np.float32)
    sa = (rng.standard_normal(512) * 0.1).astype(np.float32)
    sb = sa.copy()
    got = aac_oracle_run(O, windows, coeffs, seq, kb, sb)
    prev = (0, int(kb[0]))
    for f in range(nf):
        s2 = np.array([seq[f], prev[0]], np.int32); k2 = np.array([kb[f], prev[1]], np.int32)
        out = np.zeros(1024, np.float32)
        assert R.ffref_aac_imdct_and_windowing(ptr(np.ascontiguousarray(coeffs[f]), f32p), ptr(s2, i32p), ptr(k2, i32p), ptr(sa, f32p),
                                               ptr(out, f32p)) == 0
        assert np.array_equal(out.view(np.uint32), got[f].view(np.uint32)), (f, seq[f], prev)
        prev = (int(seq[f]), int(kb[f]))
    assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32))


#: FfoAacTnsFilter (oracle/ffo.h)
TNS_FILTER_DTYPE = np.dtype([("start", np.int32), ("size", np.int32), ("inc", np.int32), ("order", np.int32), ("coef", np.float32, 20)])


def aac_tns_case(rng, short):
    """TemporalNoiseShaping + the IndividualChannelStream fields apply_tns reads, as the bitstream parser leaves them
    (decode_tns, aacdec.c): reflection coefficients from the tns_tmp2_map tables' range, orders up to the profile's limit"""
    nw, n = (8, 128) if short else (1, 1024)
    num_swb = 14 if short else 49
    cuts = np.sort(rng.choice(np.arange(4, n, 4), num_swb - 1, replace=False))
    swb = np.concatenate(([0], cuts, [n])).astype(np.uint16)
    n_filt = np.zeros(8, np.int32); length = np.zeros((8, 4), np.int32); direction = np.zeros((8, 4), np.int32)
    order = np.zeros((8, 4), np.int32); coef = np.zeros((8, 4, 20), np.float32)
    for w in range(nw):
        n_filt[w] = rng.integers(0, 2 if short else 4)
        for f in range(n_filt[w]):
            length[w, f] = rng.integers(1, num_swb)
            direction[w, f] = rng.integers(0, 2)
            order[w, f] = rng.integers(0, 8 if short else 21)
            coef[w, f, :order[w, f]] = np.sin(rng.uniform(-1.4, 1.4, order[w, f])).astype(np.float32)
    tns_max_bands = int(rng.integers(num_swb // 2, num_swb + 1))
    max_sfb = int(rng.integers(num_swb // 2, num_swb + 1)) if rng.integers(0, 8) else 0
    return dict(n_filt=n_filt, length=length, direction=direction, order=order, coef=coef, num_windows=nw, num_swb=num_swb, swb=swb,
                tns_max_bands=tns_max_bands, max_sfb=max_sfb)


def aac_tns_filters(O, c):
    rec = np.zeros(32, TNS_FILTER_DTYPE)
    n = O.ffo_aac_tns_filters(rec.ctypes.data, ptr(c["n_filt"], i32p), ptr(c["length"], i32p), ptr(c["direction"], i32p), ptr(c["order"], i32p),
                              ptr(c["coef"], f32p), c["num_windows"], c["num_swb"], c["swb"].ctypes.data_as(C.POINTER(C.c_uint16)),
                              c["tns_max_bands"], c["max_sfb"])
    return rec[:n]


def test_aac_apply_tns():
    """AACDecDSP.apply_tns, float: the decoder's all-pole filter and the LTP path's moving-average one, long and short windows,
    both directions, orders 0..20, band limits that clip or empty the range - bit-identical"""
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(2720)
    nrec = 0
    for rep in range(300):
        c = aac_tns_case(rng, rep & 1)
        decode = int(rep % 3 != 0)
        a = (rng.standard_normal(1024) * 10.0 ** float(rng.integers(-2, 4))).astype(np.float32)
        b = a.copy()
        assert R.ffref_aac_apply_tns(ptr(a, f32p), ptr(c["n_filt"], i32p), ptr(c["length"], i32p), ptr(c["direction"], i32p), ptr(c["order"], i32p),
                                     ptr(c["coef"], f32p), c["num_windows"], c["num_swb"], c["swb"].ctypes.data_as(C.POINTER(C.c_uint16)),
                                     c["tns_max_bands"], c["max_sfb"], decode) == 0
        rec = aac_tns_filters(O, c)
        nrec += len(rec)
        for r in rec:
            O.ffo_aac_tns_run(ptr(b, f32p), r.ctypes.data if hasattr(r, "ctypes") else np.array(r).ctypes.data, decode)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (rep, decode)
    assert nrec > 300


@pytest.mark.parametrize("inv", [0, 1])
def test_mdct_vs_naive(inv):
    """the float transform stays within 2^-18 * max|ref| of the double-precision cosine sum"""
    O = ffi.oracle()
    n = 256
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, 2 * n if not inv else n).astype(np.float32)
    oc = O.ffo_mdct_create(inv, n, 1.0)
    out = np.zeros(n, np.float32); nv = np.zeros(n, np.float64)
    O.ffo_mdct_run(oc, ptr(out, f32p), ptr(x, f32p), 4)
    (O.ffo_mdct_naive_inv if inv else O.ffo_mdct_naive_fwd)(n, 1.0, nv.ctypes.data_as(C.POINTER(C.c_double)), ptr(x, f32p))
    O.ffo_mdct_free(oc)
    assert np.abs(out - nv).max() <= 2.0 ** -18 * np.abs(nv).max() * 4


@pytest.mark.parametrize("dst", ["rgb24", "bgra"])
@pytest.mark.parametrize("sf,sw,sh,dw,dh,flags", [("yuv422p", 64, 16, 64, 16, ffi.SWS_BICUBIC), ("yuv422p", 1078, 6, 1078, 6, ffi.SWS_BICUBIC),
                                                  ("yuv422p", 64, 40, 160, 88, ffi.SWS_BICUBIC), ("yuv422p", 96, 54, 48, 28, ffi.SWS_BILINEAR),
                                                  ("yuv422p", 96, 54, 120, 54, ffi.SWS_BICUBIC),
                                                  ("yuv422p", 64, 40, 64, 40, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND)])
def test_422_to_packed_rgb(dst, sf, sw, sh, dw, dh, flags):
    """4:2:2 planar sources to packed RGB: the chroma banks run from the source's own chroma plane to dstW / 2 x dstH
    (libswscale/utils.c:1359-1397); equal sizes included — yuv422p has a table converter of its own in the reference
    (YUV422FUNC, yuv2rgb.c:238-281: every line its own chroma row), whose bytes are what the scaler's one-tap path gives.
    (4:4:4 sources switch the reference to its full-chroma writers, utils.c:1276-1285: test_full_chroma_rgb.)"""
    from ffmpeg_amd import swscale as S
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(sw + dw + len(dst) + len(sf))
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=3)
    R_ctx = R.ffref_sws_create(sw, sh, PIX[sf], dw, dh, PIX[dst], flags, 1)
    assert R_ctx
    want = ffi.alloc_frame(PIX[dst], dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert R.ffref_sws_scale(R_ctx, sp, ss, 0, sh, dp, ds) == dh
    R.ffref_sws_free(R_ctx)
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[dst], flags)
    # equal sizes without ACCURATE_RND: the host tables name the table converter now (round 5; its own 4:2:2 form is pinned in
    # tests/test_sws_unscaled_forms_cpu.py) — the banks they carry still describe the scaler's one-tap path, whose bytes are the same
    assert ht.unscaled_yuv2rgb == (sw == dw and sh == dh and not flags & ffi.SWS_ACCURATE_RND)
    t = ffi.make_otables(sw, sh, PIX[sf], dw, dh, PIX[dst], flags, ht.banks(), ht.coeffs())
    got = ffi.alloc_frame(PIX[dst], dw, dh)
    gp, gs = ffi.planes(got)
    assert O.ffo_sws_scale_frame(C.byref(t), sp, ss, gp, gs) == dh
    assert np.array_equal(got[0], want[0]), "%d bytes differ" % (got[0] != want[0]).sum()
    assert S.HostTables(sw, sh, PIX["yuv444p"], dw, dh, PIX[dst], flags).full() is not None   # 4:4:4: the full-chroma writers (test_full_chroma_rgb)


FULL_CHR_CASES = [("rgb24", "yuv444p", 64, 36, 128, 72, 4), ("bgr24", "yuv444p", 97, 53, 60, 41, 4), ("rgba", "yuv444p", 40, 30, 40, 30, 4 | 0x40000),
                  ("rgb24", "yuv420p", 64, 36, 128, 72, 4 | 0x2000), ("bgra", "yuv420p", 66, 38, 131, 73, 4), ("rgb24", "yuv422p", 64, 36, 96, 54, 2 | 0x2000),
                  ("argb", "nv12", 64, 36, 127, 71, 4 | 0x40000), ("rgb24", "yuv444p", 48, 32, 48, 64, 2), ("bgr24", "yuv420p", 80, 60, 160, 60, 0x10 | 0x2000),
                  ("rgb24", "yuv444p", 128, 64, 32, 16, 4 | 0x40000), ("abgr", "yuv420p", 64, 36, 64, 36, 4 | 0x2000 | 0x40000)]


@pytest.mark.parametrize("dst,sf,sw,sh,dw,dh,flags", FULL_CHR_CASES)
def test_full_chroma_rgb(dst, sf, sw, sh, dw, dh, flags):
    """SWS_FULL_CHR_H_INT on packed RGB targets — asked for (0x2000), or forced by an odd width or a 4:4:4 source (utils.c:1270-1290):
    chroma keeps full horizontal resolution and the yuv2rgb_full_{1,2,X} writers run (output.c:1998-2310).  Our banks and the six
    coefficients (host restatement) + the oracle's writers == the reference's frame; the coefficients and the effective flag == the
    reference context's own"""
    from ffmpeg_amd import swscale as S
    R, O = ffi.ref(), ffi.oracle()
    rng = np.random.default_rng(sw + dw + len(dst) + len(sf) + flags)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=3)
    R_ctx = R.ffref_sws_create(sw, sh, PIX[sf], dw, dh, PIX[dst], flags, 1)
    assert R_ctx
    assert R.ffref_sws_flags(R_ctx) & 0x2000, "the reference runs this case with full chroma"
    rk = (C.c_int * 6)()
    R.ffref_sws_full_coeffs(R_ctx, rk)
    want = ffi.alloc_frame(PIX[dst], dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert R.ffref_sws_scale(R_ctx, sp, ss, 0, sh, dp, ds) == dh
    R.ffref_sws_free(R_ctx)
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[dst], flags)
    assert ht.full() == list(rk) and ht.t.flags & 0x2000 and ht.t.hChr.n == dw
    if ht.unscaled_yuv2rgb:     # the table converter of equal-size yuv420p -> RGB without ACCURATE_RND ignores the flag (swscale_unscaled.c:2425-2431)
        return
    t = ffi.make_otables(sw, sh, PIX[sf], dw, dh, PIX[dst], flags, ht.banks(), ht.coeffs(), full=ht.full())
    got = ffi.alloc_frame(PIX[dst], dw, dh)
    gp, gs = ffi.planes(got)
    assert O.ffo_sws_scale_frame(C.byref(t), sp, ss, gp, gs) == dh
    assert np.array_equal(got[0], want[0]), "%d bytes differ" % (got[0] != want[0]).sum()


ALPHA_CASES = [("yuv420p", "yuva420p", 64, 36, 128, 72, 4), ("yuv422p", "yuva444p", 65, 37, 40, 30, 2), ("nv12", "yuva422p", 64, 36, 96, 54, 4 | 0x40000),
               ("yuva420p", "yuv420p", 64, 36, 128, 72, 4), ("yuva444p", "rgb24", 64, 36, 100, 50, 4), ("yuva422p", "nv12", 66, 38, 33, 19, 2),
               ("yuva420p", "bgr24", 64, 36, 64, 36, 4), ("yuv420p", "yuva420p", 64, 36, 64, 36, 4), ("yuva444p", "yuv420p", 48, 32, 48, 32, 4)]


@pytest.mark.parametrize("sf,dst,sw,sh,dw,dh,flags", ALPHA_CASES)
def test_alpha_on_one_side(sf, dst, sw, sh, dw, dh, flags):
    """An alpha plane on one side only.  As a source it is not read (needAlpha = isALPHA(src) && isALPHA(dst), utils.c:1398); as a target it
    is filled with 255 (ff_swscale, swscale.c:536-553; planarCopyWrapper's fillPlane for the equal-size copy, swscale_unscaled.c:2150-2160).
    So the reference's frame == its frame for the base formats, plus an opaque plane — which is what the host tables restate
    (`dst_alpha_fill`, both formats mapped to their base)."""
    from ffmpeg_amd import swscale as S
    R = ffi.ref()
    base = {"yuva420p": "yuv420p", "yuva422p": "yuv422p", "yuva444p": "yuv444p"}
    rng = np.random.default_rng(sw + dw + len(dst) + len(sf) + flags)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=3)
    out = []
    for a, b in ((sf, dst), (base.get(sf, sf), base.get(dst, dst))):
        ctx = R.ffref_sws_create(sw, sh, PIX[a], dw, dh, PIX[b], flags, 1)
        assert ctx
        want = ffi.alloc_frame(PIX[b], dw, dh)
        for p in want:
            p[:] = 7
        sp, ss = ffi.planes(src)
        dp, ds = ffi.planes(want)
        assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, dp, ds) == dh
        R.ffref_sws_free(ctx)
        out.append(want)
    n = len(out[1])
    for p, q in zip(out[0][:n], out[1]):
        assert np.array_equal(p, q)
    if dst in base:
        assert len(out[0]) == 4 and (out[0][3] == 255).all()
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[dst], flags)
    assert ht.t.dst_alpha_fill == (dst in base) and ht.t.srcFormat == PIX[base.get(sf, sf)] and ht.t.dstFormat == PIX[base.get(dst, dst)]
    # (a source alpha plane into the alpha byte of packed RGB is alpha on BOTH sides: dst_alpha_fill 2, tests/test_sws_unscaled_forms_cpu.py)
    assert S.HostTables(sw, sh, PIX["yuva420p"], 2 * sw, sh, PIX["rgba"], flags).t.dst_alpha_fill == 2


ALPHA2_CASES = [("yuva420p", "yuva420p", 64, 36, 128, 72, 4), ("yuva420p", "yuva444p", 65, 37, 40, 30, 2), ("yuva444p", "yuva422p", 64, 36, 96, 54, 4 | 0x40000),
                ("yuva422p", "yuva420p", 66, 38, 33, 19, 2), ("yuva420p", "yuva420p", 96, 54, 48, 27, 4), ("yuva444p", "yuva420p", 48, 32, 100, 50, 0x200)]


@pytest.mark.parametrize("sf,dst,sw,sh,dw,dh,flags", ALPHA2_CASES)
def test_alpha_on_both_sides_is_the_luma_scaler(sf, dst, sw, sh, dw, dh, flags):
    """Planar YUVA on both sides: the reference scales the alpha plane with the LUMA banks and the luma dither (lum_h_scale and
    lum_planar_vscale on plane 3, hscale.c:63-79, vscale.c:57-70).  So its alpha plane == the LUMA plane it produces for the base formats
    when the source's Y plane is replaced by A — which is what libffhip runs (dst_alpha_fill == 2: a second pass of the context) — and
    its other planes are the base conversion's."""
    from ffmpeg_amd import swscale as S
    R = ffi.ref()
    base = {"yuva420p": "yuv420p", "yuva422p": "yuv422p", "yuva444p": "yuv444p"}
    rng = np.random.default_rng(sw + dw + len(dst) + len(sf) + flags)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=3)
    assert len(src) == 4

    def run(a, b, planes):
        ctx = R.ffref_sws_create(sw, sh, PIX[a], dw, dh, PIX[b], flags, 1)
        assert ctx
        want = ffi.alloc_frame(PIX[b], dw, dh)
        for p in want:
            p[:] = 7
        sp, ss = ffi.planes(planes)
        dp, ds = ffi.planes(want)
        assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, dp, ds) == dh
        R.ffref_sws_free(ctx)
        return want
    full = run(sf, dst, src)
    plain = run(base[sf], base[dst], src[:3])
    alpha_as_luma = run(base[sf], base[dst], [src[3], src[1], src[2]])
    assert len(full) == 4
    for p in range(3):
        assert np.array_equal(full[p], plain[p])
    assert np.array_equal(full[3], alpha_as_luma[0]) and not (full[3] == 255).all() and not np.array_equal(full[3], full[0])
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[dst], flags)
    assert ht.t.dst_alpha_fill == 2 and ht.t.srcFormat == PIX[base[sf]] and ht.t.dstFormat == PIX[base[dst]]


def test_sws_scale_frame_slice_threads_equal_one_thread():
    """the CPU baseline's slice-threaded leg (bench.py) goes through sws_scale_frame(), the entry that reaches ff_sws_slice_worker on
    every slice thread (libswscale/swscale.c:1405-1420, 1645-1679): its output is the single-threaded one bit for bit"""
    R = ffi.ref()
    if not hasattr(R, "ffref_sws_scale_frame"):
        pytest.skip("oracle/_ref predates ffref_sws_scale_frame")
    R.ffref_frame_alloc.restype = C.c_void_p
    R.ffref_frame_alloc.argtypes = [C.c_int] * 3
    R.ffref_frame_plane.restype = C.c_void_p
    R.ffref_frame_plane.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    R.ffref_sws_scale_frame.argtypes = [C.c_void_p] * 3
    R.ffref_frame_free.argtypes = [C.c_void_p]
    NV12, sw, sh, dw, dh = 23, 320, 180, 640, 360

    def plane(f, i, rows):
        ls = C.c_int()
        p = R.ffref_frame_plane(f, i, C.byref(ls))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(rows, ls.value))
    rng = np.random.default_rng(9)
    src = R.ffref_frame_alloc(sw, sh, NV12)
    for i, rows in ((0, sh), (1, sh // 2)):
        pl = plane(src, i, rows)
        pl[:] = rng.integers(0, 256, pl.shape, dtype=np.uint8)
    outs = []
    for threads in (1, 4):
        ctx = R.ffref_sws_create(sw, sh, NV12, dw, dh, NV12, ffi.SWS_BICUBIC, threads)
        dst = R.ffref_frame_alloc(dw, dh, NV12)
        assert R.ffref_sws_scale_frame(ctx, dst, src) >= 0
        outs.append([plane(dst, 0, dh)[:, :dw].copy(), plane(dst, 1, dh // 2)[:, :dw].copy()])
        R.ffref_sws_free(ctx)
        R.ffref_frame_free(dst)
    R.ffref_frame_free(src)
    assert outs[0][0].std() > 10
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
