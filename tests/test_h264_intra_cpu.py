"""CPU pins of the picture pipeline's intra reconstruction, no GPU involved:

  1. oracle/ffo_h264.c's ffo_h264_hl_decode_intra_mb() == the reference's own ff_h264_hl_decode_mb() (libavcodec/h264_mb.c:802,
     compiled in place, driven by oracle/refbuild/ffref_shim_h264mb.c) — samples and the consumed sl->mb;
  2. the product's HOST side (ffhip_h264_intra_pack: which blocks travel, flags, the caller's coefficients consumed as the dsp
     functions consume them) against the oracle's post-state;
  3. the kernel's per-macroblock logic (kernels/h264_intra_mb.h, the code k_h264_intra_frame runs on the GPU) executed by
     oracle/libffemul.so lane by lane on the CPU == the oracle, on whole pictures in decoder order.
"""
import ctypes as C
import os

import numpy as np
import pytest

import ffi
import h264_intra_gen as G

EMUL_SO = os.path.join(ffi.ROOT, "oracle", "libffemul.so")


def _planes(rng, mb_w, mb_h, pad=0):
    sy, sc = mb_w * 16 + pad, mb_w * 8 + pad
    return [rng.integers(0, 256, (mb_h * 16, sy), dtype=np.uint8), rng.integers(0, 256, (mb_h * 8, sc), dtype=np.uint8),
            rng.integers(0, 256, (mb_h * 8, sc), dtype=np.uint8)], [sy, sc, sc]


@pytest.mark.skipif(not os.path.exists(ffi.REF_SO), reason="oracle/_ref not built")
@pytest.mark.parametrize("mtype", [G.I16, G.I4, G.I8, G.PCM])
def test_oracle_hl_decode_mb_vs_reference(mtype):
    O, R = ffi.oracle(), ffi.ref()
    rng = np.random.default_rng(1000 + mtype)
    mb_w, mb_h = 5, 4
    u8 = C.POINTER(C.c_uint8)
    for it in range(400 if mtype != G.PCM else 20):
        mx, my = int(rng.integers(0, mb_w)), int(rng.integers(0, mb_h))
        planes, st = _planes(rng, mb_w, mb_h, pad=int(rng.choice([0, 16])))
        d = G.make_intra_mb(rng, mx, my, mb_w, mb_h, mtype)
        want = [p.copy() for p in planes]
        mb_ref = d["mb"].copy()
        dc = d["luma_dc"].copy()
        at = [want[0].ctypes.data + my * 16 * st[0] + mx * 16, want[1].ctypes.data + my * 8 * st[1] + mx * 8,
              want[2].ctypes.data + my * 8 * st[2] + mx * 8]
        R.ffref_h264_hl_decode_intra_mb(C.cast(at[0], u8), C.cast(at[1], u8), C.cast(at[2], u8), st[0], st[1], mx, my, mb_w, d["type"],
                                        d["pred16"], d["chroma_pred"], G._p(d["pred4"], C.c_uint8), d["topleft"], d["topright"],
                                        G._p(d["nnzc"], C.c_uint8), d["cbp"], G._p(mb_ref, C.c_int16), G._p(dc, C.c_int16),
                                        G._p(d["qmul"], C.c_int32), G._p(d["pcm"], C.c_uint8))
        got = [p.copy() for p in planes]
        mb_o = G.oracle_decode(O, d, got, st)
        for pl in range(3):
            assert np.array_equal(got[pl], want[pl]), "type %d iteration %d plane %d" % (mtype, it, pl)
            assert (want[pl] != planes[pl]).any()
        if mtype != G.PCM:
            assert np.array_equal(mb_o, mb_ref), "consumed coefficients differ (type %d iteration %d)" % (mtype, it)


def _pack(L, d, coefs, ncoef):
    rec = G.to_record(d)
    mb = d["mb"].copy()
    n = C.c_int32(ncoef)
    r = L.ffhip_h264_intra_pack(rec.ctypes.data, d["nnzc"].ctypes.data, mb.ctypes.data, d["luma_dc"].ctypes.data,
                                G._p(d["pcm"], C.c_uint8), G._p(coefs, C.c_int16), C.byref(n), C.c_int32(coefs.size))
    assert r == 0
    return rec, mb, n.value


def _decode_picture(rng, mb_w, mb_h, frac, pad, split=0):
    """a picture's intra macroblocks through oracle (planes `want`) and through pack + emulation (planes `got`)"""
    from ffmpeg_amd import _lib
    L = _lib.lib()
    L.ffhip_h264_intra_pack.restype = C.c_int
    E = C.CDLL(EMUL_SO)
    E.ffemul_h264_intra_set_split(split)  # the kernel's two wavefronts: all luma, then all chroma
    O = ffi.oracle()
    planes, st = _planes(rng, mb_w, mb_h, pad)
    want = [p.copy() for p in planes]
    recs, coefs, ncoef = [], np.zeros(mb_w * mb_h * 400 + 64, np.int16), 0
    rows = np.zeros(mb_h + 1, np.int32)
    for my in range(mb_h):
        for mx in range(mb_w):
            if rng.random() >= frac:
                continue
            d = G.make_intra_mb(rng, mx, my, mb_w, mb_h)
            mb_o = G.oracle_decode(O, d, want, st)
            rec, mb_p, ncoef = _pack(L, d, coefs, ncoef)
            if d["type"] != G.PCM:
                assert np.array_equal(mb_p, mb_o), "host side consumed sl->mb differently from the dsp functions at (%d, %d)" % (mx, my)
            recs.append(rec)
            rows[my + 1] += 1
    rows = np.cumsum(rows).astype(np.int32)
    recs = np.concatenate(recs) if recs else np.zeros(0, G.INTRA_DT)
    got = [p.copy() for p in planes]
    u8 = C.POINTER(C.c_uint8)
    r = E.ffemul_h264_intra_frame(got[0].ctypes.data_as(u8), got[1].ctypes.data_as(u8), got[2].ctypes.data_as(u8), C.c_ssize_t(st[0]),
                                  C.c_ssize_t(st[1]), mb_w, mb_h, C.c_void_p(recs.ctypes.data), G._p(rows, C.c_int32), G._p(coefs, C.c_int16))
    assert r == 0
    return planes, want, got, len(recs)


@pytest.mark.parametrize("mb_w,mb_h,frac,pad,split", [(1, 1, 1.0, 0, 0), (2, 3, 1.0, 4, 0), (8, 6, 1.0, 0, 0), (9, 5, .35, 12, 0), (20, 12, 1.0, 0, 0),
                                                      (8, 6, 1.0, 0, 1), (9, 5, .35, 12, 1)])
def test_kernel_logic_emulated_on_cpu_equals_oracle(mb_w, mb_h, frac, pad, split):
    if not os.path.exists(EMUL_SO):
        pytest.skip("oracle/libffemul.so not built")
    rng = np.random.default_rng(mb_w * 100 + mb_h)
    for it in range(6 if mb_w * mb_h < 100 else 2):
        planes, want, got, n = _decode_picture(rng, mb_w, mb_h, frac, pad, split)
        for pl in range(3):
            bad = np.argwhere(got[pl] != want[pl])
            assert not len(bad), "picture %d plane %d: %d mismatches, first at row %d column %d (%d intra macroblocks)" % (
                it, pl, len(bad), bad[0][0], bad[0][1], n)
        if n:
            assert (want[0] != planes[0]).any()


def test_intra_pack_rejects_bad_arguments():
    from ffmpeg_amd import _lib
    L = _lib.lib()
    L.ffhip_h264_intra_pack.restype = C.c_int
    rng = np.random.default_rng(5)
    d = G.make_intra_mb(rng, 1, 1, 4, 4, G.I4)
    rec = G.to_record(d)
    coefs = np.zeros(64, np.int16)      # too small for a run
    n = C.c_int32(0)
    mb = d["mb"].copy()
    r = L.ffhip_h264_intra_pack(C.c_void_p(rec.ctypes.data), G._p(d["nnzc"], C.c_uint8), G._p(mb, C.c_int16), None, None,
                                G._p(coefs, C.c_int16), C.byref(n), C.c_int32(coefs.size))
    assert r == _lib.ENOMEM and n.value == 0 and np.array_equal(mb, d["mb"])
    rec["type"] = 7
    big = np.zeros(1024, np.int16)
    assert L.ffhip_h264_intra_pack(C.c_void_p(rec.ctypes.data), G._p(d["nnzc"], C.c_uint8), G._p(mb, C.c_int16), None, None,
                                   G._p(big, C.c_int16), C.byref(n), C.c_int32(big.size)) == _lib.EINVAL


@pytest.mark.skipif(not os.path.exists(ffi.REF_SO), reason="oracle/_ref not built")
@pytest.mark.parametrize("depth,mb_w,mb_h,frac", [(10, 1, 1, 1.0), (10, 8, 6, 1.0), (9, 5, 4, 1.0), (12, 9, 5, .4), (14, 6, 4, 1.0), (10, 20, 12, 1.0)])
def test_kernel_logic_emulated_on_cpu_equals_reference_above_8_bits(depth, mb_w, mb_h, frac):
    """uint16_t samples, int32 coefficients (High 10 and the other depths H.264 defines): the product's host side (ffhip_h264_intra_pack_hbd:
    which blocks travel, the luma DCs ahead of the run, I_PCM fields unpacked, the caller's sl->mb consumed) + the kernel's
    per-macroblock logic executed lane by lane on the CPU == the reference's own ff_h264_hl_decode_mb() at that depth (compiled in
    place, hl_decode_mb_simple_16 / _complex), whole pictures in decoder order."""
    if not os.path.exists(EMUL_SO):
        pytest.skip("oracle/libffemul.so not built")
    from ffmpeg_amd import _lib
    L, E, R = _lib.lib(), C.CDLL(EMUL_SO), ffi.ref()
    L.ffhip_h264_intra_pack_hbd.restype = C.c_int
    E.ffemul_h264_intra_set_split(int(depth == 9 or depth == 12))  # two of the cases in the kernel's two-wavefront form
    rng = np.random.default_rng(depth * 1000 + mb_w * 100 + mb_h)
    u8 = C.POINTER(C.c_uint8)
    for it in range(4 if mb_w * mb_h < 100 else 2):
        pad = int(rng.choice([0, 4, 12]))
        sy, sc = mb_w * 16 + pad, mb_w * 8 + pad
        planes = [rng.integers(0, 1 << depth, (mb_h * 16, sy), dtype=np.uint16), rng.integers(0, 1 << depth, (mb_h * 8, sc), dtype=np.uint16),
                  rng.integers(0, 1 << depth, (mb_h * 8, sc), dtype=np.uint16)]
        st = [2 * sy, 2 * sc, 2 * sc]
        want = [p.copy() for p in planes]
        recs, coefs, ncoef = [], np.zeros(mb_w * mb_h * 816 + 64, np.int16), 0
        rows = np.zeros(mb_h + 1, np.int32)
        for my in range(mb_h):
            for mx in range(mb_w):
                if rng.random() >= frac:
                    continue
                d = G.make_intra_mb(rng, mx, my, mb_w, mb_h, depth=depth)
                mb_r = G.ref_decode(R, d, want, st, mb_w)
                rec, mb_p = G.to_record(d), d["mb"].copy()
                n = C.c_int32(ncoef)
                assert L.ffhip_h264_intra_pack_hbd(depth, rec.ctypes.data, d["nnzc"].ctypes.data, mb_p.ctypes.data, d["luma_dc"].ctypes.data,
                                                   G._p(d["pcm"], C.c_uint8), G._p(coefs, C.c_int16), C.byref(n), C.c_int32(coefs.size)) == 0
                ncoef = n.value
                if d["type"] != G.PCM:
                    assert np.array_equal(mb_p, mb_r), "host side consumed sl->mb differently from the dsp functions at (%d, %d)" % (mx, my)
                recs.append(rec)
                rows[my + 1] += 1
        rows = np.cumsum(rows).astype(np.int32)
        recs = np.concatenate(recs) if recs else np.zeros(0, G.INTRA_DT)
        got = [p.copy() for p in planes]
        r = E.ffemul_h264_intra_frame_bd(depth, C.cast(got[0].ctypes.data, u8), C.cast(got[1].ctypes.data, u8), C.cast(got[2].ctypes.data, u8),
                                         C.c_ssize_t(st[0]), C.c_ssize_t(st[1]), mb_w, mb_h, C.c_void_p(recs.ctypes.data), G._p(rows, C.c_int32),
                                         G._p(coefs, C.c_int16))
        assert r == 0
        for pl in range(3):
            bad = np.argwhere(got[pl] != want[pl])
            assert not len(bad), "picture %d plane %d: %d mismatches, first at row %d column %d (%d intra macroblocks)" % (
                it, pl, len(bad), bad[0][0], bad[0][1], len(recs))
            assert want[pl].max() < (1 << depth)
        if len(recs):
            assert (want[0] != planes[0]).any()


@pytest.mark.skipif(not os.path.exists(ffi.REF_SO), reason="oracle/_ref not built")
@pytest.mark.parametrize("depth,mb_w,mb_h,frac", [(8, 1, 1, 1.0), (8, 8, 6, 1.0), (8, 9, 5, .4), (8, 20, 12, 1.0), (10, 7, 5, 1.0), (12, 6, 4, .5)])
def test_444_planes_as_luma_only_records_equal_reference(depth, mb_w, mb_h, frac):
    """4:4:4 (hl_decode_mb_444, libavcodec/h264_mb_template.c:256-362): the product splits an intra macroblock into three luma-only records
    (ffhip_h264_intra_pack_plane on plane p's slices of the decoder's arrays, qmul[0] = the plane's own) and runs the luma phases of the
    wavefront kernel on each plane.  Host packing + the kernel's per-macroblock logic in its luma-only form (executed lane by lane on the
    CPU) == the reference's own ff_h264_hl_decode_mb() on a 4:4:4 context, whole pictures in decoder order, sl->mb consumed alike."""
    if not os.path.exists(EMUL_SO):
        pytest.skip("oracle/libffemul.so not built")
    import h264_inter_gen as I
    from ffmpeg_amd import _lib
    L, E, R = _lib.lib(), C.CDLL(EMUL_SO), ffi.ref()
    rng = np.random.default_rng(4440 + depth * 1000 + mb_w * 100 + mb_h)
    u8 = C.POINTER(C.c_uint8)
    dt, px = (np.uint8, 1) if depth == 8 else (np.uint16, 2)
    wide = 1 if depth == 8 else 2
    for it in range(4 if mb_w * mb_h < 100 else 2):
        pad = int(rng.choice([0, 4, 12]))
        sy = mb_w * 16 + pad
        planes = [rng.integers(0, 1 << depth, (mb_h * 16, sy), dtype=dt) for _ in range(3)]
        want = [p.copy() for p in planes]
        dec = I.Dec(R, "ffref_", depth, mb_w, mb_h, sy * px, sy * px, 0, cfmt=3)
        dec.set_cur([a.ctypes.data for a in want])
        recs = [[], [], []]
        coefs = [np.zeros(mb_w * mb_h * 816 + 64, np.int16) for _ in range(3)]
        ncoef = [0, 0, 0]
        rows = np.zeros((3, mb_h + 1), np.int32)
        for my in range(mb_h):
            for mx in range(mb_w):
                if rng.random() >= frac:
                    continue
                d = G.make_intra_mb(rng, mx, my, mb_w, mb_h, depth=depth, cfmt=3)
                mb_r = dec.decode_intra(d)
                mb_p = d["mb"].copy()
                mb16, dc16 = mb_p.view(np.int16), d["luma_dc"].view(np.int16)
                for p in range(3):
                    rec = G.to_record(d)
                    rec["qmul"][0][0] = d["qmul"][p]
                    n = C.c_int32(ncoef[p])
                    pcm = None if d["pcm"] is None else d["pcm"][(32 * depth * p if depth > 8 else 256 * p):]
                    assert L.ffhip_h264_intra_pack_plane(depth, rec.ctypes.data, d["nnzc"][40 * p:].ctypes.data, mb16[256 * p * wide:].ctypes.data,
                                                         dc16[16 * p * wide:].ctypes.data, None if pcm is None else pcm.ctypes.data,
                                                         coefs[p].ctypes.data, C.byref(n), C.c_int32(coefs[p].size)) == 0
                    ncoef[p] = n.value
                    assert not (rec["cbp"][0] & 0x30) and not (rec["blocks"][0] >> 16)
                    recs[p].append(rec)
                    rows[p, my + 1] += 1
                if d["type"] != G.PCM:
                    assert np.array_equal(mb_p, mb_r), "host side consumed sl->mb differently from the dsp functions at (%d, %d)" % (mx, my)
        dec.close()
        got = [p.copy() for p in planes]
        E.ffemul_h264_intra_set_split(2)           # luma only: cb / cr are not touched
        for p in range(3):
            rs = np.cumsum(rows[p]).astype(np.int32)
            rc = np.concatenate(recs[p]) if recs[p] else np.zeros(0, G.INTRA_DT)
            dummy = np.zeros(16, np.uint8)
            r = E.ffemul_h264_intra_frame_bd(depth, C.cast(got[p].ctypes.data, u8), dummy.ctypes.data_as(u8), dummy.ctypes.data_as(u8),
                                             C.c_ssize_t(sy * px), C.c_ssize_t(0), mb_w, mb_h, C.c_void_p(rc.ctypes.data), G._p(rs, C.c_int32),
                                             G._p(coefs[p], C.c_int16))
            assert r == 0 and not dummy.any()
        E.ffemul_h264_intra_set_split(0)
        for pl in range(3):
            bad = np.argwhere(got[pl] != want[pl])
            assert not len(bad), "picture %d plane %d: %d mismatches, first at row %d column %d" % (it, pl, len(bad), bad[0][0], bad[0][1])
            assert want[pl].max() < (1 << depth)
            if len(recs[0]):
                assert (want[pl] != planes[pl]).any()
