"""GPU parity: H264PredContext (intra prediction, 8 bits, 4:2:0) vs the oracle, through the C ABI - the batch face and the
signature-exact host faces."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p, i16p, i32p
from test_oracle_vs_ref import H264_PRED_KINDS, h264_pred_kind, h264_pred_grid, h264_pred_apply, h264_pred_plane

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def hip_pred_apply(kind, pic, recs, coeffs):
    """the batch face over (x, y, mode, flags, aux) rows; returns the picture (coeffs updated in place)"""
    from ffmpeg_amd import h264
    torch = _torch()
    width = pic.shape[1]
    rec = np.zeros(len(recs), h264.PRED_DTYPE)
    rec["offset"] = recs[:, 1] * width + recs[:, 0]
    rec["aux"], rec["mode"], rec["flags"] = recs[:, 4], recs[:, 2], recs[:, 3]
    order = np.random.default_rng(len(recs)).permutation(len(recs))        # the order of independent blocks is free
    d_pic = torch.from_numpy(pic).cuda()
    d_co = None if coeffs is None else torch.from_numpy(coeffs).cuda()
    d_rec = torch.from_numpy(rec[order].view(np.uint8).reshape(len(recs), 12).copy()).cuda()
    h264.pred_batch(kind, d_pic, width, d_rec, len(recs), coeffs=d_co)
    torch.cuda.synchronize()
    if coeffs is not None:
        coeffs[:] = d_co.cpu().numpy()
    return d_pic.cpu().numpy()


@pytest.mark.parametrize("kind", range(8))
def test_h264_pred_batch(kind):
    """a picture's worth of independent blocks: every mode and flag combination, random / saturated / smooth content, picture
    widths on and off the dword grid"""
    O = ffi.oracle()
    n = h264_pred_kind(kind)[0]
    for width, height, content in ((1283, 360, 0), (1280, 352, 1), (642, 200, 2)):
        rng = np.random.default_rng(2650 + 10 * kind + content)
        if content == 0:
            pic = rng.integers(0, 256, (height, width), dtype=np.uint8)
        elif content == 1:
            pic = rng.choice(np.array([0, 255], np.uint8), (height, width))
        else:
            pic = np.clip(np.add.outer(np.arange(height) * 2, np.arange(width) * -1) + 200 + rng.integers(-5, 6, (height, width)), 0, 255).astype(np.uint8)
        recs = h264_pred_grid(rng, kind, height, width)
        coeffs = wc = None
        if 4 <= kind < 7:
            coeffs = rng.integers(-300, 301, (len(recs) + 1) * n * n).astype(np.int16)
            coeffs[:n * n] = rng.choice(np.array([-32768, 32767], np.int16), n * n)
            wc = coeffs.copy()
        want = pic.copy()
        h264_pred_apply(O, "ffo", kind, want, recs, wc)
        got = hip_pred_apply(kind, pic.copy(), recs, coeffs)
        assert (want != pic).sum() > 500
        assert np.array_equal(got, want), "%d mismatches, first %s" % ((got != want).sum(), np.argwhere(got != want)[:3])
        if 4 <= kind < 7:
            assert np.array_equal(coeffs, wc) and coeffs[-n * n:].any()      # consumed blocks cleared, the spare one untouched


def test_h264_pred_picture_edge():
    """blocks on the picture's first row / column / corner with the modes a decoder uses there: only existing neighbours are read
    (the buffer starts at the picture's first sample)"""
    from ffmpeg_amd import h264
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(2690)
    width = 64
    pic = rng.integers(0, 256, (64, width), dtype=np.uint8)
    #           kind, x, y, mode, flags
    cases = [(0, 0, 0, 11, 0), (0, 20, 0, 1, 0), (0, 32, 0, 9, 0), (0, 44, 0, 8, 0), (0, 0, 20, 0, 0), (0, 0, 32, 10, 0),
             (1, 0, 0, 11, 0), (1, 16, 0, 1, 0), (1, 40, 0, 9, 0), (1, 0, 24, 0, 2), (1, 0, 40, 10, 0), (1, 0, 56, 3, 2),
             (2, 0, 0, 6, 0), (2, 24, 0, 1, 0), (2, 40, 0, 4, 0), (2, 0, 16, 2, 0), (2, 0, 32, 5, 0), (2, 52, 0, 9, 0),
             (3, 0, 0, 6, 0), (3, 32, 0, 1, 0), (3, 0, 24, 2, 0), (3, 0, 44, 5, 0)]
    for kind in range(4):
        recs = np.array([(x, y, m, f, 0) for k, x, y, m, f in cases if k == kind], np.int32)
        want = pic.copy()
        h264_pred_apply(O, "ffo", kind, want, recs)
        got = hip_pred_apply(kind, pic.copy(), recs, None)
        assert np.array_equal(got, want), kind


def test_h264_pred_host_faces():
    """ff_h264_pred_init_hip(): every member the reference fills for the H.264 codec, called as the decoder calls them"""
    from ffmpeg_amd import h264
    _torch()
    O = ffi.oracle()
    h = h264.pred_init()
    rng = np.random.default_rng(2691)
    at = lambda a: a.ctypes.data + 16 * 48 + 16
    for kind in range(4):
        n, nmodes, name = H264_PRED_KINDS[kind]
        for mode in range(nmodes):
            for rep in range(2):
                a = h264_pred_plane(rng, rep + mode); b = a.copy()
                tr = rng.integers(0, 256, 4, dtype=np.uint8)
                tl_tr = (1 if mode in (4, 5, 6) else int(rng.integers(0, 2)), int(rng.integers(0, 2)))
                if kind == 0:
                    h.pred4x4[mode](at(a), tr.ctypes.data, 48)
                    O.ffo_h264_pred4x4(mode, C.cast(at(b), u8p), ptr(tr), 48)
                elif kind == 1:
                    h.pred8x8l[mode](at(a), tl_tr[0], tl_tr[1], 48)
                    O.ffo_h264_pred8x8l(mode, C.cast(at(b), u8p), tl_tr[0], tl_tr[1], 48)
                else:
                    getattr(h, name)[mode](at(a), 48)
                    getattr(O, "ffo_h264_" + name)(mode, C.cast(at(b), u8p), 48)
                assert np.array_equal(a, b), (name, mode, rep)
    assert not h.pred4x4[12] and not h.pred16x16[7] and not h.pred8x8_add[0]   # members the reference leaves unset stay NULL
    scan = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (3, 0), (2, 1), (3, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 2), (3, 2), (2, 3), (3, 3)]
    for rep in range(3):
        for name, n in (("pred4x4_add", 4), ("pred8x8l_add", 8), ("pred8x8l_filter_add", 8)):
            for mode in (0, 1):
                a = h264_pred_plane(rng, rep); b = a.copy()
                ca = rng.integers(-300, 301, n * n).astype(np.int16); cb = ca.copy()
                if name == "pred8x8l_filter_add":
                    tl, tr = int(rng.integers(0, 2)), int(rng.integers(0, 2))
                    h.pred8x8l_filter_add[mode](at(a), ca.ctypes.data, tl, tr, 48)
                    O.ffo_h264_pred8x8l_filter_add(mode, C.cast(at(b), u8p), ptr(cb, i16p), tl, tr, 48)
                else:
                    getattr(h, name)[mode](at(a), ca.ctypes.data, 48)
                    getattr(O, "ffo_h264_" + name)(mode, C.cast(at(b), u8p), ptr(cb, i16p), 48)
                assert np.array_equal(a, b) and not ca.any(), (name, mode, rep)
        for name, nb in (("pred8x8_add", 4), ("pred16x16_add", 16)):
            for mode in (2, 1):
                a = h264_pred_plane(rng, rep); b = a.copy()
                ca = rng.integers(-300, 301, nb * 16).astype(np.int16); cb = ca.copy()
                offs = np.array([4 * x + 4 * y * 48 for x, y in scan[:nb]], np.int32)
                getattr(h, name)[mode](at(a), offs.ctypes.data, ca.ctypes.data, 48)
                getattr(O, "ffo_h264_" + name)(mode, C.cast(at(b), u8p), ptr(offs, i32p), ptr(cb, i16p), 48)
                assert np.array_equal(a, b) and not ca.any(), (name, mode, rep)


def test_h264_pred_init_rejects():
    """what this library does not replace keeps the C pointers: other codecs' variants, depths H.264 does not define, 4:2:2"""
    from ffmpeg_amd import h264, _lib
    _torch()
    hctx = h264.H264PredContext()
    L = _lib.lib()
    # (1: AV_CODEC_ID_MPEG1VIDEO has no H264PredContext; 139 / 69: VP8 and RV40 exist at 8 bits, 4:2:0 only)
    for args in ((h264.CODEC_ID_H264, 11, 1), (h264.CODEC_ID_H264, 8, 4), (h264.CODEC_ID_H264, 10, -1), (1, 8, 1), (139, 10, 1), (69, 8, 2)):
        assert L.ff_h264_pred_init_hip(C.byref(hctx), *args) < 0


CODECS = {"svq3": 23, "rv40": 69, "vp7": 178, "vp8": 139}   # AV_CODEC_ID_* (libavcodec/codec_id.h)


@pytest.mark.skipif(not ffi.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("codec", sorted(CODECS))
def test_h264_pred_other_codecs_match_the_reference(codec):
    """ff_h264_pred_init(h, AV_CODEC_ID_SVQ3 / _RV40 / _VP7 / _VP8, 8, 1) (h264pred.c:540-578): the table ff_h264_pred_init_hip()
    leaves for the codec has a face exactly where the reference's has a function (started from a cleared table on both sides), and
    every face == the reference's member on random, saturated and smooth neighbourhoods — the codecs' own forms (down-left / vertical-
    left / horizontal-up with RV40's down-left edge and without, SVQ3's and RV40's planes, VP8's smoothed vertical / horizontal,
    TrueMotion, the 127 / 129 DCs, RV40's chroma DCs) and the members they share with H.264"""
    from ffmpeg_amd import h264
    _torch()
    R = ffi.ref()
    R.ffref_h264_pred_set_codec(CODECS[codec])
    try:
        h = h264.pred_init(CODECS[codec], 8, 1)
        rng = np.random.default_rng(CODECS[codec])
        at = lambda a: a.ctypes.data + 16 * 48 + 16
        checked = 0
        for table, name, nmodes in ((0, "pred4x4", 15), (1, "pred8x8l", 12), (2, "pred8x8", 11), (3, "pred16x16", 9)):
            for mode in range(nmodes):
                has = bool(R.ffref_h264_pred_has(table, mode))
                assert bool(getattr(h, name)[mode]) == has, (codec, name, mode, has)
                if not has:
                    continue
                for rep in range(4):
                    a = h264_pred_plane(rng, rep + mode)
                    if rep == 3:
                        a[:] = rng.choice(np.array([0, 255], np.uint8), a.shape)
                    b = a.copy()
                    tr = rng.integers(0, 256, 4, dtype=np.uint8)
                    if table == 0:
                        h.pred4x4[mode](at(a), tr.ctypes.data, 48)
                        R.ffref_h264_pred4x4(mode, C.cast(at(b), u8p), ptr(tr), 48)
                    elif table == 1:
                        tl, trf = (1 if mode in (4, 5, 6) else int(rng.integers(0, 2))), int(rng.integers(0, 2))
                        h.pred8x8l[mode](at(a), tl, trf, 48)
                        R.ffref_h264_pred8x8l(mode, C.cast(at(b), u8p), tl, trf, 48)
                    else:
                        getattr(h, name)[mode](at(a), 48)
                        getattr(R, "ffref_h264_" + name)(mode, C.cast(at(b), u8p), 48)
                    assert np.array_equal(a, b), (codec, name, mode, rep, np.argwhere(a != b)[:4])
                    checked += 1
        assert checked >= 4 * 40
    finally:
        R.ffref_h264_pred_set_bit_depth(8)   # the shim's context is shared: back to H.264


def test_h264_pred_codec_forms_batch():
    """kind FFHIP_H264_PRED_CODEC of the batch face: many independent blocks of the codecs' own forms in one launch == the host faces'
    results block by block (which test_h264_pred_other_codecs_match_the_reference pins to the reference)"""
    from ffmpeg_amd import h264
    torch = _torch()
    rng = np.random.default_rng(77)
    width, height = 640, 192
    pic = rng.integers(0, 256, (height, width), dtype=np.uint8)
    recs = []
    for by in range(1, height // 32 - 1):
        for bx in range(1, width // 32 - 1):
            v = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18, 19, 20, 21, 32, 33, 34, 35, 36]))
            x, y = 32 * bx + 8, 32 * by + 8
            recs.append((x, y, v, 0, (y - 1) * width + x + 4))
    recs = np.array(recs, np.int64)
    got = hip_pred_apply(8, pic.copy(), recs, None)
    want = pic.copy()
    for x, y, v, _, aux in recs:      # one block at a time through the same face
        one = hip_pred_apply(8, want.copy(), np.array([(x, y, v, 0, aux)], np.int64), None)
        n = 4 if v < 16 else 8 if v < 32 else 16
        want[y:y + n, x:x + n] = one[y:y + n, x:x + n]
        assert np.array_equal(one[:y], want[:y]) and np.array_equal(one[y + n:], want[y + n:])   # nothing outside the block
    assert np.array_equal(got, want)


@pytest.mark.skipif(not ffi.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("depth", [8, 9, 10, 12, 14])
def test_h264_pred_422_matches_the_reference(depth):
    """chroma_format_idc 2 (4:2:2): pred8x8[] are the 8 wide x 16 tall forms (h264pred.c:478-512, h264pred_template.c:477-817) and
    pred8x8_add[] walk eight 4x4 blocks (:1302-1330) — the host faces ff_h264_pred_init_hip(..., 2) installs and the batch face
    (kind FFHIP_H264_PRED8x16) == the reference's own context initialised the same way, every mode, every depth"""
    from ffmpeg_amd import h264, _lib
    torch = _torch()
    R = ffi.ref()
    R.ffref_h264_pred_set_format.argtypes = [C.c_int, C.c_int]
    R.ffref_h264_pred_set_format(depth, 2)
    try:
        hc = h264.pred_init(bit_depth=depth, chroma_format_idc=2)
        rng = np.random.default_rng(170 + depth)
        W, px = 48, 2 if depth > 8 else 1
        dt = np.uint16 if depth > 8 else np.uint8
        stride = W * px

        def patch(extreme):
            a = rng.integers(0, 1 << depth, (40, W)).astype(dt)
            if extreme:
                a[::3] = rng.choice(np.array([0, (1 << depth) - 1], dt), a[::3].shape)
            return a

        def at(a, r, c):
            return C.c_void_p(a.ctypes.data + (r * W + c) * px)
        for rep in range(4):
            for mode in range(11):
                p0 = patch(rep == 1)
                a, b = p0.copy(), p0.copy()
                R.ffref_h264_pred8x8(mode, C.cast(at(a, 8, 16), ffi.u8p), stride)
                hc.pred8x8[mode](at(b, 8, 16), stride)
                assert np.array_equal(a, b), ("pred8x16", mode, rep)
                assert (a[8:24, 16:24] != p0[8:24, 16:24]).any() or mode == 6 or rep == 1, "an 8 x 16 block is written"
            bo = np.zeros(16, np.int32)     # block_offset[]: 4x4 blocks of the 8 x 16 chroma block, scan order; 4..7 unused
            for i in range(4):
                bo[i] = ((i >> 1) * 4 * W + (i & 1) * 4) * px
                bo[8 + i] = ((2 + (i >> 1)) * 4 * W + (i & 1) * 4) * px
            for mode in (1, 2):             # HOR_PRED8x8 / VERT_PRED8x8: the lossless members
                p0 = patch(rep == 1)
                co = rng.integers(-(1 << depth), 1 << depth, 8 * 16).astype(np.int32 if depth > 8 else np.int16)
                a, b, ca, cb = p0.copy(), p0.copy(), co.copy(), co.copy()
                R.ffref_h264_pred8x8_add(mode, C.cast(at(a, 8, 16), ffi.u8p), bo.ctypes.data_as(C.POINTER(C.c_int)), C.cast(ca.ctypes.data, ffi.i16p), stride)
                hc.pred8x8_add[mode](at(b, 8, 16), C.c_void_p(bo.ctypes.data), C.c_void_p(cb.ctypes.data), stride)
                assert np.array_equal(a, b) and np.array_equal(ca, cb), ("pred8x16_add", mode)
        if depth != 8:
            return
        # the batch face (8-bit): many blocks of mixed modes in one launch
        nb = 300
        plane = patch(False)
        plane = np.tile(plane, (12, 10))[: 24 * 17, : 24 * 18].copy()
        PW = plane.shape[1]
        want = plane.copy()
        recs = np.zeros(nb, dtype=np.dtype([("offset", "<i4"), ("aux", "<i4"), ("mode", "u1"), ("flags", "u1"), ("pad", "u1", (2,))]))
        k = 0
        for by in range(17):
            for bx in range(18):
                if k >= nb:
                    break
                mode = int(rng.integers(0, 11))
                r0, c0 = by * 24 + 4, bx * 24 + 8
                recs[k] = ((r0 * PW + c0) * px, 0, mode, 0, (0, 0))
                R.ffref_h264_pred8x8(mode, C.cast(C.c_void_p(want.ctypes.data + (r0 * PW + c0) * px), ffi.u8p), PW * px)
                k += 1
        d_plane = torch.from_numpy(plane.view(np.uint8).reshape(plane.shape[0], -1)).cuda()
        d_recs = torch.from_numpy(recs.view(np.uint8).reshape(nb, 12)).cuda()
        L = _lib.lib()
        assert L.ffhip_h264_pred_batch_dev(7, d_plane.data_ptr(), PW * px, None, d_recs.data_ptr(), nb, None) == 0
        torch.cuda.synchronize()
        got = d_plane.cpu().numpy().view(dt).reshape(plane.shape)
        assert np.array_equal(got, want), "%d samples differ" % (got != want).sum()
    finally:
        R.ffref_h264_pred_set_format(8, 1)


@pytest.mark.skipif(not ffi.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("depth", [9, 10, 12, 14])
def test_h264_pred_above_8_bits_matches_the_reference(depth):
    """the host faces ff_h264_pred_init_hip() installs at 9 / 10 / 12 / 14 bits == the reference's own instantiations of
    h264pred_template.c at that depth (libavcodec/h264pred.c:448-538; oracle/_ref), every mode of every table, 16-bit samples"""
    from ffmpeg_amd import h264
    _torch()
    R = ffi.ref()
    R.ffref_h264_pred_set_bit_depth.argtypes = [C.c_int]
    R.ffref_h264_pred_set_bit_depth(depth)
    try:
        hc = h264.pred_init(bit_depth=depth)
        rng = np.random.default_rng(70 + depth)
        W = 48
        stride = W * 2

        def patch(extreme):
            a = rng.integers(0, 1 << depth, (40, W)).astype(np.uint16)
            if extreme:
                a[::3] = rng.choice(np.array([0, (1 << depth) - 1], np.uint16), a[::3].shape)
            return a

        def at(a, r, c):
            return C.c_void_p(a.ctypes.data + (r * W + c) * 2)
        for rep in range(3):
            for mode in range(12):
                p0 = patch(rep == 1)
                a, b = p0.copy(), p0.copy()
                tr = rng.integers(0, 1 << depth, 8).astype(np.uint16)
                R.ffref_h264_pred4x4(mode, C.cast(at(a, 8, 16), ffi.u8p), C.cast(tr.ctypes.data, ffi.u8p), stride)
                hc.pred4x4[mode](at(b, 8, 16), C.c_void_p(tr.ctypes.data), stride)
                assert np.array_equal(a, b), ("pred4x4", mode)
                for tl in (0, 1):
                    for trf in (0, 1):
                        a, b = p0.copy(), p0.copy()
                        R.ffref_h264_pred8x8l(mode, C.cast(at(a, 8, 16), ffi.u8p), tl, trf, stride)
                        hc.pred8x8l[mode](at(b, 8, 16), tl, trf, stride)
                        assert np.array_equal(a, b), ("pred8x8l", mode, tl, trf)
            for mode in range(11):
                p0 = patch(rep == 1)
                a, b = p0.copy(), p0.copy()
                R.ffref_h264_pred8x8(mode, C.cast(at(a, 8, 16), ffi.u8p), stride)
                hc.pred8x8[mode](at(b, 8, 16), stride)
                assert np.array_equal(a, b), ("pred8x8", mode)
            for mode in range(7):
                p0 = patch(rep == 1)
                a, b = p0.copy(), p0.copy()
                R.ffref_h264_pred16x16(mode, C.cast(at(a, 8, 16), ffi.u8p), stride)
                hc.pred16x16[mode](at(b, 8, 16), stride)
                assert np.array_equal(a, b), ("pred16x16", mode)
            for mode in (0, 1):                                  # the lossless _add members: int32 coefficients
                for n, rf, hf in ((4, R.ffref_h264_pred4x4_add, hc.pred4x4_add), (8, R.ffref_h264_pred8x8l_add, hc.pred8x8l_add)):
                    p0 = patch(rep == 1)
                    co = rng.integers(-(1 << depth), 1 << depth, n * n).astype(np.int32)
                    a, b, ca, cb = p0.copy(), p0.copy(), co.copy(), co.copy()
                    rf(mode, C.cast(at(a, 8, 16), ffi.u8p), C.cast(ca.ctypes.data, ffi.i16p), stride)
                    hf[mode](at(b, 8, 16), C.c_void_p(cb.ctypes.data), stride)
                    assert np.array_equal(a, b) and np.array_equal(ca, cb), ("add", n, mode)
    finally:
        R.ffref_h264_pred_set_bit_depth(8)
