"""GPU parity of the signature-exact host-pointer faces (what ff_*_init_hip() installs into the reference's
function-pointer tables), checkasm style: random buffers, exact compare, incl. the cleared coefficient block."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import PIX, ptr, u8p, i16p, i32p, i8p

pytestmark = pytest.mark.gpu

LF = C.CFUNCTYPE(None, u8p, C.c_ssize_t, C.c_int, C.c_int, i8p)
LFI = C.CFUNCTYPE(None, u8p, C.c_ssize_t, C.c_int, C.c_int)
IDCT = C.CFUNCTYPE(None, u8p, i16p, C.c_ssize_t)
IDCTM = C.CFUNCTYPE(None, u8p, i32p, i16p, C.c_ssize_t, u8p)
IDCT8P = C.CFUNCTYPE(None, C.POINTER(u8p), i32p, i16p, C.c_ssize_t, u8p)
LUMADC = C.CFUNCTYPE(None, i16p, i16p, C.c_int)
CHROMADC = C.CFUNCTYPE(None, i16p, C.c_int)
QPEL = C.CFUNCTYPE(None, u8p, u8p, C.c_ssize_t)
CMP = C.CFUNCTYPE(C.c_int, C.c_void_p, u8p, u8p, C.c_ssize_t, C.c_int)


class H264DSP(C.Structure):   # member order of FFHipH264DSPContext (include/ffhip.h)
    _fields_ = [("v_loop_filter_luma", LF), ("h_loop_filter_luma", LF), ("v_loop_filter_luma_intra", LFI),
                ("h_loop_filter_luma_intra", LFI), ("v_loop_filter_chroma", LF), ("h_loop_filter_chroma", LF),
                ("v_loop_filter_chroma_intra", LFI), ("h_loop_filter_chroma_intra", LFI), ("idct_add", IDCT),
                ("idct8_add", IDCT), ("idct_dc_add", IDCT), ("idct8_dc_add", IDCT), ("idct_add16", IDCTM),
                ("idct8_add4", IDCTM), ("idct_add16intra", IDCTM), ("idct_add8", IDCT8P), ("luma_dc_dequant_idct", LUMADC),
                ("chroma_dc_dequant_idct", CHROMADC), ("add_pixels8_clear", IDCT), ("add_pixels4_clear", IDCT),
                ("h_loop_filter_luma_mbaff", LF), ("h_loop_filter_luma_mbaff_intra", LFI), ("h_loop_filter_chroma_mbaff", LF),
                ("h_loop_filter_chroma_mbaff_intra", LFI)]


class H264Qpel(C.Structure):
    _fields_ = [("put", (QPEL * 16) * 3), ("avg", (QPEL * 16) * 3)]


CHROMA = C.CFUNCTYPE(None, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int)
WEIGHT = C.CFUNCTYPE(None, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int)
BIWEIGHT = C.CFUNCTYPE(None, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)


class H264Chroma(C.Structure):
    _fields_ = [("put", CHROMA * 4), ("avg", CHROMA * 4)]


class H264Weight(C.Structure):
    _fields_ = [("weight", WEIGHT * 4), ("biweight", BIWEIGHT * 4)]


class MECmp(C.Structure):
    _fields_ = [("sad", CMP * 2), ("hadamard8_diff", CMP * 2), ("pix_abs", (CMP * 1) * 2), ("pix_abs_hpel", (CMP * 3) * 2), ("sse", CMP * 2),
                ("nsse", CMP * 2)]


def _lib():
    from ffmpeg_amd import _lib as L
    import torch
    assert torch.cuda.is_available()
    return L.lib()


def test_h264dsp_init_hip():
    L = _lib()
    O = ffi.oracle()
    c = H264DSP()
    assert L.ff_h264dsp_init_hip(C.byref(c), 11, 1) < 0          # 8 / 9 / 10 / 12 / 14 are the depths H.264 defines (h264dsp.c:135-147)
    assert L.ff_h264dsp_init_hip(C.byref(c), 8, 1) == 0
    rng = np.random.default_rng(1)
    stride = 48
    # single-block idcts (tests/checkasm/h264dsp.c:175-240)
    for name, ofn, size in (("idct_add", O.ffo_h264_idct_add, 4), ("idct8_add", O.ffo_h264_idct8_add, 8),
                            ("idct_dc_add", O.ffo_h264_idct_dc_add, 4), ("idct8_dc_add", O.ffo_h264_idct8_dc_add, 8)):
        for _ in range(3):
            dst = rng.integers(0, 256, (16, stride), dtype=np.uint8)
            blk = rng.integers(-600, 600, size * size).astype(np.int16)
            wd, wb = dst.copy(), blk.copy()
            ofn(C.cast(wd.ctypes.data + 2 * stride + 8, u8p), ptr(wb, i16p), stride)
            getattr(c, name)(C.cast(dst.ctypes.data + 2 * stride + 8, u8p), ptr(blk, i16p), stride)
            assert np.array_equal(dst, wd) and np.array_equal(blk, wb), name
    # macroblock dispatchers (tests/checkasm/h264dsp.c:242-326)
    bo = np.array([(i & 1) * 4 + ((i >> 1) & 1) * 4 * stride + ((i >> 2) & 1) * 8 + (i >> 3) * 8 * stride for i in range(16)],
                  np.int32)
    for name, ofn in (("idct_add16", O.ffo_h264_idct_add16), ("idct8_add4", O.ffo_h264_idct8_add4),
                      ("idct_add16intra", O.ffo_h264_idct_add16intra)):
        dst = rng.integers(0, 256, (24, stride), dtype=np.uint8)
        blk = rng.integers(-300, 300, 256).astype(np.int16)
        blk[16:32] = 0; blk[33:48] = 0
        nnzc = rng.integers(0, 3, 40, dtype=np.uint8)
        wd, wb = dst.copy(), blk.copy()
        ofn(C.cast(wd.ctypes.data + 3 * stride + 4, u8p), ptr(bo, i32p), ptr(wb, i16p), stride, ptr(nnzc))
        getattr(c, name)(C.cast(dst.ctypes.data + 3 * stride + 4, u8p), ptr(bo, i32p), ptr(blk, i16p), stride, ptr(nnzc))
        assert np.array_equal(dst, wd) and np.array_equal(blk, wb), name
    # loop filters (tests/checkasm/h264dsp.c:375-470): 32x16 tile, edge in the middle
    names = ["v_loop_filter_luma", "h_loop_filter_luma", "v_loop_filter_chroma", "h_loop_filter_chroma",
             "v_loop_filter_luma_intra", "h_loop_filter_luma_intra", "v_loop_filter_chroma_intra", "h_loop_filter_chroma_intra"]
    for kind, name in enumerate(names):
        changed = 0
        for alpha, beta, t in ((20, 6, 1), (80, 12, 3), (255, 18, 13)):
            base = rng.integers(110, 126, (32, 40)).astype(np.uint8)
            tc0 = np.array([t, -1, 0, t], np.int8)
            wd = base.copy()
            off = 16 * 40 + 16
            O.ffo_h264_loop_filter(kind, C.cast(wd.ctypes.data + off, u8p), 40, alpha, beta, ptr(tc0, i8p))
            got = base.copy()
            if kind < 4:
                getattr(c, name)(C.cast(got.ctypes.data + off, u8p), 40, alpha, beta, ptr(tc0, i8p))
            else:
                getattr(c, name)(C.cast(got.ctypes.data + off, u8p), 40, alpha, beta)
            assert np.array_equal(got, wd), name
            assert (wd != base).any()


def _h264_new_members(c, O, rng, check):
    """idct_add8, the DC transforms and the lossless add_pixels members through table `c`; check(got, want, what)"""
    stride = 32
    # idct_add8: blocks 16..19 (Cb) and 32..35 (Cr), the decoder's 48-entry block_offset[] and 15 x 8 nnz cache
    bo = np.zeros(48, np.int32)
    for i in range(48):
        bo[i] = (i & 1) * 4 + ((i >> 1) & 1) * 4 * stride
    for rep in range(3):
        planes = [rng.integers(0, 256, (16, stride), dtype=np.uint8) for _ in range(2)]
        blk = rng.integers(-300, 300, 768).astype(np.int16)
        blk[17 * 16 + 1:18 * 16] = 0                       # a dc-only candidate
        nnzc = rng.integers(0, 2, 120, dtype=np.uint8)
        want = [a.copy() for a in planes]
        wb = blk.copy()
        dp = (u8p * 2)(*[C.cast(a.ctypes.data + 2 * stride + 8, u8p) for a in want])
        O.ffo_h264_idct_add8(dp, ptr(bo, i32p), ptr(wb, i16p), stride, ptr(nnzc))
        got = [a.copy() for a in planes]
        gp = (u8p * 2)(*[C.cast(a.ctypes.data + 2 * stride + 8, u8p) for a in got])
        c.idct_add8(gp, ptr(bo, i32p), ptr(blk, i16p), stride, ptr(nnzc))
        check(np.stack(got), np.stack(want), "idct_add8 picture")
        check(blk, wb, "idct_add8 coefficients")
        assert (np.stack(want) != np.stack(planes)).any()
    for qmul in (16, 1024, 40000, -7):
        inp = rng.integers(-2000, 2000, 16).astype(np.int16)
        a = rng.integers(-100, 100, 256).astype(np.int16); b = a.copy()
        O.ffo_h264_luma_dc_dequant_idct(ptr(a, i16p), ptr(inp.copy(), i16p), qmul)
        c.luma_dc_dequant_idct(ptr(b, i16p), ptr(inp, i16p), qmul)
        check(b, a, "luma_dc_dequant_idct")
        a = rng.integers(-2000, 2000, 64).astype(np.int16); b = a.copy()
        O.ffo_h264_chroma_dc_dequant_idct(ptr(a, i16p), qmul)
        c.chroma_dc_dequant_idct(ptr(b, i16p), qmul)
        check(b, a, "chroma_dc_dequant_idct")
    for n, name in ((4, "add_pixels4_clear"), (8, "add_pixels8_clear")):
        dst = rng.integers(0, 256, (16, stride), dtype=np.uint8)
        blk = rng.integers(-300, 300, n * n).astype(np.int16)
        wd, wb = dst.copy(), blk.copy()
        O.ffo_h264_add_pixels_clear(n, C.cast(wd.ctypes.data + 2 * stride + 8, u8p), ptr(wb, i16p), stride)
        getattr(c, name)(C.cast(dst.ctypes.data + 2 * stride + 8, u8p), ptr(blk, i16p), stride)
        check(dst, wd, name)
        check(blk, wb, name + " coefficients")
        assert not wb.any()


def _eq(got, want, what):
    assert np.array_equal(got, want), what


def test_h264dsp_dc_dequant_add8_add_pixels():
    """the members a real macroblock needs beside the IDCTs (h264dsp.h:96-108): idct_add8, luma / chroma dc_dequant_idct,
    add_pixels4/8_clear"""
    L = _lib()
    c = H264DSP()
    assert L.ff_h264dsp_init_hip(C.byref(c), 8, 1) == 0
    assert L.ff_h264dsp_init_hip(C.byref(H264DSP()), 8, 2) == 0  # 4:2:2: the chroma members come from the depth-generic faces
    assert L.ff_h264dsp_init_hip(C.byref(H264DSP()), 8, 4) < 0
    before = L.ffhip_shim_fallbacks()
    _h264_new_members(c, ffi.oracle(), np.random.default_rng(21), _eq)
    assert L.ffhip_shim_fallbacks() == before                  # everything ran on the device


def _c_table(O):
    """an H264DSPContext as ff_h264dsp_init() leaves it: the C functions (here: the oracle's, same signatures)"""
    c = H264DSP()
    for name, fn in (("idct_add", O.ffo_h264_idct_add), ("idct8_add", O.ffo_h264_idct8_add), ("idct_dc_add", O.ffo_h264_idct_dc_add),
                     ("idct8_dc_add", O.ffo_h264_idct8_dc_add), ("idct_add16", O.ffo_h264_idct_add16), ("idct8_add4", O.ffo_h264_idct8_add4),
                     ("idct_add16intra", O.ffo_h264_idct_add16intra), ("idct_add8", O.ffo_h264_idct_add8),
                     ("luma_dc_dequant_idct", O.ffo_h264_luma_dc_dequant_idct), ("chroma_dc_dequant_idct", O.ffo_h264_chroma_dc_dequant_idct)):
        setattr(c, name, C.cast(fn, dict(H264DSP._fields_)[name]))
    return c


def test_faces_fall_back_to_the_displaced_c_functions(monkeypatch, measure_build):
    """FFHIP_FAULT=1: every face reports a device failure before touching anything.  A face installed over a C function must
    answer through it (same bytes as the C function alone), one that displaced nothing must leave its operands untouched and
    say so — never return silently with half a result."""
    L = _lib()
    O = ffi.oracle()
    rng = np.random.default_rng(22)
    c = _c_table(O)
    assert L.ff_h264dsp_init_hip(C.byref(c), 8, 1) == 0
    assert L.ff_h264dsp_init_hip(C.byref(c), 8, 1) == 0         # a second init must not make a face its own fallback
    monkeypatch.setenv("FFHIP_FAULT", "1")
    n0 = L.ffhip_shim_fallbacks()
    stride = 48
    for name, ofn, size in (("idct_add", O.ffo_h264_idct_add, 4), ("idct8_add", O.ffo_h264_idct8_add, 8)):
        dst = rng.integers(0, 256, (16, stride), dtype=np.uint8)
        blk = rng.integers(-600, 600, size * size).astype(np.int16)
        wd, wb = dst.copy(), blk.copy()
        ofn(C.cast(wd.ctypes.data + 2 * stride + 8, u8p), ptr(wb, i16p), stride)
        getattr(c, name)(C.cast(dst.ctypes.data + 2 * stride + 8, u8p), ptr(blk, i16p), stride)
        assert np.array_equal(dst, wd) and np.array_equal(blk, wb) and not blk.any(), name
    assert L.ffhip_shim_fallbacks() == n0 + 2
    # a member that displaced nothing: operands untouched, the failure recorded
    dst = rng.integers(0, 256, (16, stride), dtype=np.uint8)
    blk = rng.integers(1, 600, 16).astype(np.int16)
    d0, b0 = dst.copy(), blk.copy()
    c.add_pixels4_clear(C.cast(dst.ctypes.data + 8, u8p), ptr(blk, i16p), stride)
    assert np.array_equal(dst, d0) and np.array_equal(blk, b0)
    assert b"add_pixels4_clear" in L.ffhip_last_error()
    # an int-returning face
    m = MECmp()
    SADF = dict(MECmp._fields_)["sad"]._type_
    calls = []

    def c_sad(ctx, a, b, s, h):
        calls.append(h)
        return 4242
    keep = SADF(c_sad)
    m.sad[0] = keep
    assert L.ff_me_cmp_init_hip(C.byref(m)) == 0
    a = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    assert m.sad[0](None, ptr(a), ptr(a), 16, 16) == 4242 and calls == [16]
    assert m.sad[1](None, ptr(a), ptr(a), 16, 8) == 0           # nothing displaced: 0 and an error text
    monkeypatch.delenv("FFHIP_FAULT")
    assert m.sad[0](None, ptr(a), ptr(a), 16, 16) == 0 and calls == [16]   # back on the device: SAD of a block with itself
    _h264_new_members(c, O, rng, _eq)                           # and the device results are the C results anyway


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("order", ["420-first", "422-first"])
def test_fallbacks_are_kept_per_chroma_format(depth, order, monkeypatch, measure_build):
    """ff_h264dsp_init() picks six members by chroma_format_idc (h264dsp.c:113-132).  A 4:2:0 and a 4:2:2 table of one bit depth
    initialised in one process — either order — each answer, under FFHIP_FAULT=1, through the C function THEY displaced (round 3
    kept one fallback table per depth: the later init's function answered for both)."""
    L = _lib()
    CDC = dict(H264DSP._fields_)["chroma_dc_dequant_idct"]
    HLF = dict(H264DSP._fields_)["h_loop_filter_chroma_intra"]
    seen = []
    keep = {}
    tabs = {}
    for cf in ((1, 2) if order == "420-first" else (2, 1)):
        c = H264DSP()
        keep[cf] = (CDC(lambda b, q, cf=cf: seen.append(("dc", cf))), HLF(lambda p, s, a, b, cf=cf: seen.append(("lf", cf))))
        c.chroma_dc_dequant_idct, c.h_loop_filter_chroma_intra = keep[cf]
        assert L.ff_h264dsp_init_hip(C.byref(c), depth, cf) == 0
        tabs[cf] = c
    monkeypatch.setenv("FFHIP_FAULT", "1")
    blk = np.zeros(64, np.int16 if depth == 8 else np.int32)
    pix = np.zeros((32, 64), np.uint8)
    for cf in (1, 2, 1):
        tabs[cf].chroma_dc_dequant_idct(C.cast(blk.ctypes.data, i16p), 16)
        tabs[cf].h_loop_filter_chroma_intra(C.cast(pix.ctypes.data + 8 * 64 + 16, u8p), 64, 20, 20)
    assert seen == [("dc", 1), ("lf", 1), ("dc", 2), ("lf", 2), ("dc", 1), ("lf", 1)], seen


def test_h264qpel_init_hip():
    """tests/checkasm/h264qpel.c:51-82: put/avg x sizes x 16 positions, src and dst buffers compared"""
    L = _lib()
    c = H264Qpel()
    assert L.ff_h264qpel_init_hip(C.byref(c), 8) == 0
    rng = np.random.default_rng(2)
    stride = 64
    for avg in (0, 1):
        tab = c.avg if avg else c.put
        for size_idx in range(3):
            for mc in range(16):
                src = rng.integers(0, 256, (40, stride), dtype=np.uint8)
                dst = rng.integers(0, 256, (40, stride), dtype=np.uint8)
                wd = dst.copy()
                so, do = 8 * stride + 11, 8 * stride + 16
                ffi.oracle().ffo_h264_qpel(avg, size_idx, mc, C.cast(wd.ctypes.data + do, u8p),
                                           C.cast(src.ctypes.data + so, u8p), stride)
                s0 = src.copy()
                tab[size_idx][mc](C.cast(dst.ctypes.data + do, u8p), C.cast(src.ctypes.data + so, u8p), stride)
                assert np.array_equal(dst, wd) and np.array_equal(src, s0), (avg, size_idx, mc)


def test_me_cmp_init_hip():
    """tests/checkasm/motion.c:37-92: the context pointer is NULL"""
    L = _lib()
    O = ffi.oracle()
    c = MECmp()
    assert L.ff_me_cmp_init_hip(C.byref(c)) == 0
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    b = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    for _ in range(6):
        y1, x1, y2, x2 = rng.integers(0, 40, 4)
        x1 = (x1 // 16) * 16
        pa, pb = C.cast(a.ctypes.data + int(y1) * 64 + int(x1), u8p), C.cast(b.ctypes.data + int(y2) * 64 + int(x2), u8p)
        for h in (8, 16):
            assert c.sad[0](None, pa, pb, 64, h) == O.ffo_sad(16, pa, pb, 64, h)
            assert c.sad[1](None, pa, pb, 64, h) == O.ffo_sad(8, pa, pb, 64, h)
            assert c.pix_abs[0][0](None, pa, pb, 64, h) == O.ffo_sad(16, pa, pb, 64, h)
            assert c.hadamard8_diff[0](None, pa, pb, 64, h) == O.ffo_hadamard8_diff16(pa, pb, 64, h)
        assert c.hadamard8_diff[1](None, pa, pb, 64, 8) == O.ffo_hadamard8_diff8x8(pa, pb, 64)


def test_h264chroma_and_weight_init_hip():
    """tests/checkasm/h264chroma.c shape for the chroma table; slice-header ranges for the weight tables"""
    L = _lib()
    O = ffi.oracle()
    c = H264Chroma()
    assert L.ff_h264chroma_init_hip(C.byref(c), 8) == 0
    rng = np.random.default_rng(4)
    stride = 48
    for avg in (0, 1):
        tab = c.avg if avg else c.put
        for idx, w in enumerate((8, 4, 2)):
            for (x, y) in ((0, 0), (3, 0), (0, 5), (7, 7), (4, 2)):
                src = rng.integers(0, 256, (24, stride), dtype=np.uint8)
                dst = rng.integers(0, 256, (24, stride), dtype=np.uint8)
                wd = dst.copy()
                off = 3 * stride + 8
                O.ffo_h264_chroma_mc(avg, w, C.cast(wd.ctypes.data + off, u8p), C.cast(src.ctypes.data + off, u8p), stride, 8, x, y)
                tab[idx](C.cast(dst.ctypes.data + off, u8p), C.cast(src.ctypes.data + off, u8p), stride, 8, x, y)
                assert np.array_equal(dst, wd), (avg, w, x, y)
    wc = H264Weight()
    assert L.ff_h264dsp_weight_init_hip(C.byref(wc), 8) == 0
    for idx, w in enumerate((16, 8, 4, 2)):
        for ld, wt, ws, of in ((0, 1, 1, 0), (5, 37, -12, 9), (7, -128, 127, -128), (2, 127, 127, 127)):
            blk = rng.integers(0, 256, (20, stride), dtype=np.uint8)
            src = rng.integers(0, 256, (20, stride), dtype=np.uint8)
            a, b = blk.copy(), blk.copy()
            O.ffo_h264_weight(w, C.cast(a.ctypes.data + 56, u8p), stride, 16, ld, wt, of)
            wc.weight[idx](C.cast(b.ctypes.data + 56, u8p), stride, 16, ld, wt, of)
            assert np.array_equal(a, b), ("weight", w, ld, wt, of)
            a, b = blk.copy(), blk.copy()
            O.ffo_h264_biweight(w, C.cast(a.ctypes.data + 56, u8p), C.cast(src.ctypes.data + 56, u8p), stride, 8, ld, wt, ws, of)
            wc.biweight[idx](C.cast(b.ctypes.data + 56, u8p), C.cast(src.ctypes.data + 56, u8p), stride, 8, ld, wt, ws, of)
            assert np.array_equal(a, b), ("biweight", w, ld, wt, ws, of)


# ---------------------------------------------------------------------------------------------
# swscale per-line members with the reference's signatures (ff_sws_init_swscale_hip): tests/checkasm/sw_scale.c's shapes
# ---------------------------------------------------------------------------------------------
HSCALE = C.CFUNCTYPE(None, C.c_void_p, i16p, C.c_int, u8p, i16p, i32p, C.c_int)
PLANE1 = C.CFUNCTYPE(None, i16p, u8p, C.c_int, u8p, C.c_int)
PLANEX = C.CFUNCTYPE(None, i16p, C.c_int, C.POINTER(i16p), u8p, C.c_int, u8p, C.c_int)
NV12CX = C.CFUNCTYPE(None, C.c_int, u8p, i16p, C.c_int, C.POINTER(i16p), C.POINTER(i16p), u8p, C.c_int)


class SwsLine(C.Structure):   # member order of FFHipSwsLineContext (include/ffhip.h)
    _fields_ = [("hyScale", HSCALE), ("hcScale", HSCALE), ("yuv2plane1", PLANE1), ("yuv2planeX", PLANEX), ("yuv2nv12cX", NV12CX)]


@pytest.mark.parametrize("fault", [0, 1])
def test_sws_init_swscale_hip(fault, monkeypatch, measure_build):
    """hyScale / hcScale with checkasm's adversarial coefficients (sw_scale.c:356-458), yuv2plane1 / yuv2planeX with its dither
    offsets (:109-180), yuv2nv12cX (:182-262).  fault = 1: every member answers through the C function it displaced."""
    L = _lib()
    O = ffi.oracle()
    rng = np.random.default_rng(60 + fault)
    lc = SwsLine()
    if fault:   # the table as the C init leaves it (the oracle's functions have the members' shapes)
        seen = []

        def c_hscale(c, dst, dstW, src, filt, pos, fs):
            seen.append("h")
            O.ffo_hscale8to15(dst, dstW, src, filt, pos, fs)
        k1, k2 = HSCALE(c_hscale), PLANEX(lambda f, n, s, d, w, di, o: (seen.append("x"), O.ffo_yuv2planeX8(f, n, s, d, w, di, o))[1])
        lc.hyScale, lc.hcScale, lc.yuv2planeX = k1, k1, k2
    assert L.ff_sws_init_swscale_hip(C.byref(lc), PIX["nv12"], PIX["nv12"]) == 0
    if fault:
        monkeypatch.setenv("FFHIP_FAULT", "1")
    dstW, srcW = 512, 560
    for fs in (4, 8, 16):
        src = rng.integers(0, 256, srcW + 16, dtype=np.uint8)
        filt = rng.integers(-(1 << 14), 1 << 14, (dstW, fs)).astype(np.int16)
        filt[::3] = -((1 << 14) // max(fs - 1, 1))
        filt[::3, 0] = (1 << 15) - 1
        pos = np.sort(rng.integers(0, srcW - fs, dstW)).astype(np.int32)
        want, got = np.zeros(dstW, np.int16), np.zeros(dstW, np.int16)
        O.ffo_hscale8to15(ptr(want, i16p), dstW, ptr(src), ptr(filt, i16p), ptr(pos, i32p), fs)
        (lc.hyScale if fs != 8 else lc.hcScale)(None, ptr(got, i16p), dstW, ptr(src), ptr(filt, i16p), ptr(pos, i32p), fs)
        assert np.array_equal(got, want), fs
    n = 333
    dither = rng.integers(0, 128, 8, dtype=np.uint8)
    for fs in (1, 2, 4, 16):
        lines = rng.integers(-32768, 32768, (fs, n + 3)).astype(np.int16)
        lines2 = rng.integers(-32768, 32768, (fs, n + 3)).astype(np.int16)
        filt = rng.integers(-4096, 8192, fs).astype(np.int16)
        rows = (i16p * fs)(*[ptr(lines[j], i16p) for j in range(fs)])
        rows2 = (i16p * fs)(*[ptr(lines2[j], i16p) for j in range(fs)])
        for off in (0, 3):
            want, got = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
            O.ffo_yuv2planeX8(ptr(filt, i16p), fs, rows, ptr(want), n, ptr(dither), off)
            lc.yuv2planeX(ptr(filt, i16p), fs, rows, ptr(got), n, ptr(dither), off)
            assert np.array_equal(got, want), ("planeX", fs, off)
            if not fault:   # plane1 displaced nothing in the fault table
                O.ffo_yuv2plane1_8(rows[0], ptr(want), n, ptr(dither), off)
                lc.yuv2plane1(rows[0], ptr(got), n, ptr(dither), off)
                assert np.array_equal(got, want), ("plane1", off)
        if not fault:
            for fmt, swap in ((PIX["nv12"], 0), (PIX["nv21"], 1)):
                want, got = np.zeros(2 * n, np.uint8), np.zeros(2 * n, np.uint8)
                O.ffo_yuv2nv12cX(swap, ptr(dither), ptr(filt, i16p), fs, rows, rows2, ptr(want), n)
                lc.yuv2nv12cX(fmt, ptr(dither), ptr(filt, i16p), fs, rows, rows2, ptr(got), n)
                assert np.array_equal(got, want), ("nv12cX", fs, swap)
    if fault:
        assert "h" in seen and "x" in seen


@pytest.mark.parametrize("dst", ["rgb24", "bgra", "argb"])
def test_sws_packed_line_faces(dst):
    """ffhip_sws_yuv2packedX / 2 / 1 == yuv2rgb_{X,2,1}_c_template (the oracle's restatement, pinned to the reference's members)"""
    from ffmpeg_amd import swscale as S
    from test_oracle_vs_ref import packed_line_inputs
    L = _lib()
    O = ffi.oracle()
    ctx = S.SwsContext(64, 16, PIX["yuv420p"], 128, 32, PIX[dst], ffi.SWS_BICUBIC)
    luts = ffi.OLuts()
    k = ffi.OYuv2RgbCoeffs(*[ffi.DEFAULT_COEFFS[n] for n in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
    O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
    lay, bpp, dstW = ffi.RGB_LAYOUT[PIX[dst]], (3 if dst == "rgb24" else 4), 330
    rng = np.random.default_rng(lay + 70)
    for lfs, cfs in ((4, 4), (1, 4), (8, 3)):
        lum, cu, cv, lf, cf = packed_line_inputs(rng, dstW, lfs, cfs)
        rl = (i16p * lfs)(*[ptr(lum[j], i16p) for j in range(lfs)])
        ru = (i16p * cfs)(*[ptr(cu[j], i16p) for j in range(cfs)])
        rv = (i16p * cfs)(*[ptr(cv[j], i16p) for j in range(cfs)])
        want, got = np.zeros(dstW * bpp, np.uint8), np.zeros(dstW * bpp, np.uint8)
        O.ffo_yuv2rgb_X(C.byref(luts), ptr(lf, i16p), rl, lfs, ptr(cf, i16p), ru, rv, cfs, ptr(want), dstW, lay)
        assert L.ffhip_sws_yuv2packedX(ctx._c, lf.ctypes.data, C.cast(rl, C.c_void_p), lfs, cf.ctypes.data, C.cast(ru, C.c_void_p),
                                       C.cast(rv, C.c_void_p), cfs, None, got.ctypes.data, dstW, 3) == 0
        assert np.array_equal(got, want), (lfs, cfs)
    lum, cu, cv, _, _ = packed_line_inputs(rng, dstW, 2, 2)
    r2 = [(i16p * 2)(ptr(x[0], i16p), ptr(x[1], i16p)) for x in (lum, cu, cv)]
    for ya, uva in ((0, 0), (1234, 4000), (4096, 2048)):
        want, got = np.zeros(dstW * bpp, np.uint8), np.zeros(dstW * bpp, np.uint8)
        O.ffo_yuv2rgb_2(C.byref(luts), r2[0], r2[1], r2[2], ptr(want), dstW, ya, uva, lay)
        assert L.ffhip_sws_yuv2packed2(ctx._c, C.cast(r2[0], C.c_void_p), C.cast(r2[1], C.c_void_p), C.cast(r2[2], C.c_void_p), None,
                                       got.ctypes.data, dstW, ya, uva, 3) == 0
        assert np.array_equal(got, want), (ya, uva)
    for uva in (0, 1000, 4096):
        want, got = np.zeros(dstW * bpp, np.uint8), np.zeros(dstW * bpp, np.uint8)
        O.ffo_yuv2rgb_1(C.byref(luts), ptr(lum[0], i16p), r2[1], r2[2], ptr(want), dstW, uva, lay)
        assert L.ffhip_sws_yuv2packed1(ctx._c, lum[0].ctypes.data, C.cast(r2[1], C.c_void_p), C.cast(r2[2], C.c_void_p), None, got.ctypes.data,
                                       dstW, uva, 3) == 0
        assert np.array_equal(got, want), uva
    assert L.ffhip_sws_yuv2packed1(ctx._c, lum[0].ctypes.data, C.cast(r2[1], C.c_void_p), C.cast(r2[2], C.c_void_p), None, got.ctypes.data,
                                   dstW + 1, 0, 3) < 0                   # odd widths are the C function's
    ctx.close()
