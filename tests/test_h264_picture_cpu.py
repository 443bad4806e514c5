"""The H.264 picture layer's HOST side, without a GPU.

The same experiment as tests/test_gpu_h264_decoder.py — the reference's own ff_h264_hl_decode_mb() / ff_h264_filter_mb() driven twice over
the same decoder state: once with the C dsp tables on host planes (the expected picture), once over the recording members of
integration/avcodec_h264_picture_hip.c into an FFHipH264Picture — but the recorded lists (ffhip_h264_picture_lists) are then executed on the
CPU by oracle/emul_h264_picture.cpp: flush()'s stage order, the oracle's dsp functions (pinned to the reference), the wavefront kernel's
per-macroblock phases lane by lane for the intra records.  Recording needs no device, so this pins on any machine what the recorder and
libffhip's host code decide: which member becomes which record on which plane, offsets, FFHIP_MC_EMU coordinates on unpadded references,
the scratchpad's bookkeeping, weights, the residual dispatch, the intra packing, the edge tables — at 4:2:0 and 4:4:4, 8 bits and above."""
import ctypes as C
import os

import numpy as np
import pytest

import ffi
import h264_intra_gen as G
import h264_inter_gen as I

EMUL_SO = os.path.join(ffi.ROOT, "oracle", "libffemul.so")
pytestmark = pytest.mark.skipif(not (ffi.have_ref() and I.have_ref_hip() and os.path.exists(EMUL_SO)), reason="oracle/_ref or libffemul.so not built")


class Lists(C.Structure):                       # == FFHipH264PictureLists (include/ffhip.h)
    _fields_ = [("mb_w", C.c_int), ("mb_h", C.c_int), ("bit_depth", C.c_int), ("chroma_format_idc", C.c_int),
                ("qpel", (C.c_void_p * 3) * 3), ("nqpel", (C.c_int * 3) * 3), ("cmc", (C.c_void_p * 3) * 2), ("ncmc", (C.c_int * 3) * 2),
                ("wt", C.c_void_p * 3), ("nwt", C.c_int * 3), ("idct_off", (C.c_void_p * 4) * 3), ("idct_coef", (C.c_void_p * 4) * 3),
                ("nidct", (C.c_int * 4) * 3), ("intra", C.c_void_p * 3), ("nintra", C.c_int * 3), ("intra_coef", C.c_void_p * 3),
                ("nintra_coef", C.c_int * 3), ("edges", C.c_void_p * 3), ("intra_c422", C.c_void_p), ("nintra_c422", C.c_int),
                ("intra_c422_coef", C.c_void_p), ("nintra_c422_coef", C.c_int),
                ("addpx_off", (C.c_void_p * 2) * 3), ("addpx_coef", (C.c_void_p * 2) * 3), ("naddpx", (C.c_int * 2) * 3)]


def _env():
    from ffmpeg_amd import _lib
    L = _lib.lib()                              # libffhip.so first: libffref_hip.so binds to the instance the package uses
    RH = C.CDLL(I.REF_HIP_SO)
    E = C.CDLL(EMUL_SO)
    E.ffemul_h264_picture_flush.argtypes = [C.c_void_p] * 4
    RH.ffrefhip_h264dec_record_begin.argtypes = [C.c_void_p] * 5
    RH.ffrefhip_h264dec_record_begin.restype = None
    return _lib, L, ffi.ref(), RH, E


class HostPicture:
    """an FFHipH264Picture made where there may be no device: records only"""

    def __init__(self, _lib, L, mb_w, mb_h, depth, cfmt):
        self.L, self.p = L, _lib.vp()
        assert L.ffhip_h264_picture_create_fmt(C.byref(self.p), mb_w, mb_h, depth, cfmt) == 0
        L.ffhip_h264_picture_begin(self.p)

    def lists(self):
        ls = Lists()
        assert self.L.ffhip_h264_picture_lists(self.p, C.byref(ls)) == 0
        return ls

    def close(self):
        self.L.ffhip_h264_picture_free(C.byref(self.p))


def _cpu_flush(E, ls, planes, strides, refs):
    dp = (C.c_void_p * 3)(*[a.ctypes.data for a in planes])
    rp = (C.c_void_p * 3)(*[a.ctypes.data for a in refs])
    st = (C.c_int * 3)(*strides)
    assert E.ffemul_h264_picture_flush(C.byref(ls), dp, st, rp) == 0


def _run_picture(depth, mb_w, mb_h, nref, mvr, p_intra, weights, cfmt, seed, bypass=0):
    _lib, L, R, RH, E = _env()
    rng = np.random.default_rng(seed)
    px, dt, top = (2, np.uint16, 1 << depth) if depth > 8 else (1, np.uint8, 256)
    W, H = mb_w * 16, mb_h * 16
    sy = W + int(rng.integers(0, 3)) * 16                   # row pitches in samples: NO border around a picture, only row padding
    sc, HC = (sy, H) if cfmt == 3 else (W // 2 + 16, H if cfmt == 2 else H // 2)    # 4:2:2: chroma half as wide, as tall
    ls_, uvls = sy * px, sc * px
    strides = [ls_, uvls, uvls]
    rows = [H, HC, HC]
    refs = [rng.integers(0, top, (nref * rows[pl], strides[pl] // px), dtype=dt) for pl in range(3)]
    if cfmt == 0:                                            # monochrome: the decoder's frames hold mid-grey chroma
        refs[1][:], refs[2][:] = top >> 1, top >> 1
    cpu = I.Dec(R, "ffref_", depth, mb_w, mb_h, ls_, uvls, 0, cfmt=cfmt)
    rec = I.Dec(RH, "ffrefhip_", depth, mb_w, mb_h, ls_, uvls, 1, cfmt=cfmt)
    for lst in (0, 1):
        for i in range(nref):
            j = i if lst == 0 else nref - 1 - i
            at = [refs[pl].ctypes.data + j * rows[pl] * strides[pl] for pl in range(3)]
            cpu.set_ref(lst, i, at)
            rec.set_ref(lst, i, at)
    pw = I.make_pwt(rng, weights, depth, nref)
    cpu.set_pwt(pw)
    rec.set_pwt(pw)
    dst0 = [rng.integers(0, top, (rows[pl], strides[pl] // px), dtype=dt) for pl in range(3)]
    want, got = [a.copy() for a in dst0], [a.copy() for a in dst0]
    cpu.set_cur([a.ctypes.data for a in want])
    rec.set_cur([a.ctypes.data for a in got])               # record mode: addresses only, nothing is read or written through them
    pic = HostPicture(_lib, L, mb_w, mb_h, depth, cfmt)
    RH.ffrefhip_h264dec_record_begin(rec.d, pic.p, *[r.ctypes.data for r in refs])
    nbyp = 0
    for my in range(mb_h):
        for mx in range(mb_w):
            if bypass:                                       # the lossless bypass: about half the macroblocks have QP'Y = 0
                on = rng.random() < 0.5
                nbyp += on
                cpu.set_bypass(bypass, on)
                rec.set_bypass(bypass, on)
            if rng.random() < p_intra:
                d = G.make_intra_mb(rng, mx, my, mb_w, mb_h, depth=depth, cfmt=cfmt)
                a, b = cpu.decode_intra(d), rec.decode_intra(d)
                assert d["type"] == G.PCM or np.array_equal(a, b)         # sl->mb consumed alike
            else:
                m = I.make_inter_mb(rng, cpu.bits, mx, my, nref, mvr, depth=depth, cfmt=cfmt)
                assert np.array_equal(cpu.decode_inter(m), rec.decode_inter(m))
    for pl in range(3):
        assert np.array_equal(got[pl], dst0[pl])             # recording touched no sample
    ls = pic.lists()
    if bypass:
        assert nbyp > 0 and (p_intra >= 1 or sum(ls.naddpx[pl][k] for pl in range(3) for k in range(2)) > 0)
        assert cfmt != 3 or p_intra >= 1 or all(ls.naddpx[pl][0] for pl in range(3))     # 4:4:4: add_pixels on Cb / Cr too
    assert (ls.mb_w, ls.mb_h, ls.bit_depth, ls.chroma_format_idc) == (mb_w, mb_h, depth, cfmt or 1)       # monochrome: the 4:2:0 object
    if cfmt == 3:
        assert not any(ls.ncmc[c][s] for c in range(2) for s in range(3))
        assert p_intra >= 1 or all(ls.nqpel[pl][0] for pl in range(3))
    elif cfmt == 2:
        assert ls.nintra_c422 == ls.nintra[0] and not ls.nintra[1] and (p_intra <= 0 or ls.nintra_c422)
    else:
        assert not any(ls.nqpel[pl][s] for pl in (1, 2) for s in range(3)) and not ls.nintra[1] and not ls.nintra[2]
    _cpu_flush(E, ls, got, strides, refs)
    for pl in range(3):
        assert (want[pl] != dst0[pl]).sum() > 100 and want[pl].max() < top
        bad = got[pl] != want[pl]
        assert not bad.any(), "plane %d: %d mismatches, first at %s" % (pl, bad.sum(), np.argwhere(bad)[0])
    pic.close()
    cpu.close()
    rec.close()


@pytest.mark.parametrize("depth,mb_w,mb_h,nref,mvr,p_intra,weights", [
    (8, 6, 4, 2, 40, 0.0, 0), (8, 6, 4, 2, 600, 0.0, 0), (8, 11, 7, 3, 4000, 0.0, 1), (8, 11, 7, 3, 300, 0.0, 2), (8, 20, 11, 2, 120, .15, 1),
    (8, 9, 5, 1, 64, 1.0, 0), (10, 7, 5, 2, 600, .2, 1), (12, 6, 4, 2, 400, .2, 2)])
def test_recorded_picture_executed_on_cpu_equals_reference_420(depth, mb_w, mb_h, nref, mvr, p_intra, weights):
    _run_picture(depth, mb_w, mb_h, nref, mvr, p_intra, weights, 1, seed=depth * 1000 + mb_w * 31 + mvr + weights)


@pytest.mark.parametrize("depth,mb_w,mb_h,nref,mvr,p_intra,weights", [
    (8, 6, 4, 2, 40, 0.0, 0), (8, 11, 7, 3, 2000, 0.0, 1), (8, 11, 7, 3, 300, 0.0, 2), (8, 9, 5, 1, 64, 1.0, 0), (8, 20, 11, 2, 120, .15, 1),
    (10, 7, 5, 2, 600, .2, 1), (12, 6, 4, 2, 500, .3, 0), (14, 6, 4, 2, 400, .2, 2)])
def test_recorded_picture_executed_on_cpu_equals_reference_444(depth, mb_w, mb_h, nref, mvr, p_intra, weights):
    """hl_decode_mb_444 (libavcodec/h264_mb_template.c:256-362) over the recording members: the luma tables on Cb / Cr, three luma-only
    intra records per macroblock"""
    _run_picture(depth, mb_w, mb_h, nref, mvr, p_intra, weights, 3, seed=4440000 + depth * 1000 + mb_w * 31 + mvr + weights)


@pytest.mark.parametrize("depth,mb_w,mb_h,p_intra,cfmt", [(8, 6, 4, .2, 1), (8, 20, 11, .15, 1), (10, 7, 5, .2, 1), (8, 6, 4, .2, 3), (8, 20, 11, .15, 3),
                                                           (8, 9, 5, 1.0, 3), (10, 7, 5, .2, 3), (12, 6, 4, .3, 3), (8, 7, 5, .2, 0), (10, 6, 4, .3, 0)])
def test_recorded_deblocking_executed_on_cpu_equals_reference(depth, mb_w, mb_h, p_intra, cfmt):
    """ff_h264_filter_mb() (libavcodec/h264_loopfilter.c:716) over the recording loop-filter members -> the picture's edge tables ->
    the oracle's frame-order filter == the reference's C filter, macroblock by macroblock in raster order; 4:4:4: the luma members on all
    three planes (h264_loopfilter.c:601-703); monochrome: no chroma edge is filtered (`chroma = CHROMA(h) && ...`, :726), the planes stay"""
    _lib, L, R, RH, E = _env()
    rng = np.random.default_rng(depth * 100 + mb_w + mb_h + cfmt)
    px, dt = (2, np.uint16) if depth > 8 else (1, np.uint8)
    W, H = mb_w * 16, mb_h * 16
    sy = W + 32
    sc, HC = (sy, H) if cfmt == 3 else (W // 2 + 16, H if cfmt == 2 else H // 2)
    strides = [sy * px, sc * px, sc * px]
    mid, amp = 1 << (depth - 1), 20 << (depth - 8)
    dst0 = [(mid + rng.integers(-amp, amp + 1, (r, s))).astype(dt) for r, s in ((H, sy), (HC, sc), (HC, sc))]
    want, got = [a.copy() for a in dst0], [a.copy() for a in dst0]
    cpu = I.Dec(R, "ffref_", depth, mb_w, mb_h, strides[0], strides[1], 0, cfmt=cfmt)
    rec = I.Dec(RH, "ffrefhip_", depth, mb_w, mb_h, strides[0], strides[1], 1, cfmt=cfmt)
    cpu.set_cur([a.ctypes.data for a in want])
    rec.set_cur([a.ctypes.data for a in got])
    pic = HostPicture(_lib, L, mb_w, mb_h, depth, cfmt)
    RH.ffrefhip_h264dec_record_begin(rec.d, pic.p, *[a.ctypes.data for a in got])
    for st in I.make_filter_picture(rng, cpu.bits, mb_w, mb_h, depth, p_intra):
        cpu.filter_mb(st["mb_x"], st["mb_y"], st)
        rec.filter_mb(st["mb_x"], st["mb_y"], st)
    ls = pic.lists()
    assert all(ls.edges[pl] for pl in range(3 if cfmt else 1))
    _cpu_flush(E, ls, got, strides, got)
    for pl in range(3):
        assert (want[pl] != dst0[pl]).sum() > 50 if cfmt or not pl else np.array_equal(want[pl], dst0[pl])
        bad = got[pl] != want[pl]
        assert not bad.any(), "plane %d: %d mismatches, first at %s" % (pl, bad.sum(), np.argwhere(bad)[0])
    pic.close()
    cpu.close()
    rec.close()


def test_flush_without_a_device_is_refused_by_name():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    _lib, L, R, RH, E = _env()
    pic = HostPicture(_lib, L, 4, 4, 8, 1)
    a = np.zeros((64, 64), np.uint8)
    dp = (C.c_void_p * 3)(a.ctypes.data, a.ctypes.data, a.ctypes.data)
    st = (C.c_int * 3)(64, 64, 64)
    assert L.ffhip_h264_picture_flush(pic.p, dp, st, dp, None) == _lib.ENOSYS
    pic.close()


# ---- field pictures (PAFF) -------------------------------------------------------------------------------------------------------------
def _field_refs(rng, dec_list, refs, rows, strides, nref):
    """both lists: fields of the nref reference FRAMES, either parity, in different orders (same- and opposite-parity prediction)"""
    picks = [(int(rng.integers(0, nref)), int(rng.integers(1, 3))) for _ in range(2 * nref)]
    for lst in (0, 1):
        for i in range(nref):
            j, par = picks[lst * nref + i]
            for d, base in dec_list:
                d.set_ref_field(lst, i, [base[pl] + j * rows[pl] * strides[pl] for pl in range(3)], par)


def _run_field_frame(depth, mb_w, fmb_h, nref, mvr, p_intra, weights, cfmt, seed, deblock=False):
    """a frame decoded as two FIELD pictures (top, then bottom): every macroblock a field macroblock of the decoder's own numbering
    (mb_y = 2 * row + bottom), the references fields of frames (pic_as_field, h264_refs.c:39-48) — the recorder hands libffhip each field
    as a picture of its own: every second line of the frame buffer, half the height, twice the line size"""
    _lib, L, R, RH, E = _env()
    rng = np.random.default_rng(seed)
    px, dt, top = (2, np.uint16, 1 << depth) if depth > 8 else (1, np.uint8, 256)
    mb_h = 2 * fmb_h                                        # the frame's macroblock rows
    W, H = mb_w * 16, mb_h * 16
    sy = W + int(rng.integers(0, 3)) * 16
    sc, HC = (sy, H) if cfmt == 3 else (W // 2 + 16, H if cfmt == 2 else H // 2)
    strides = [sy * px, sc * px, sc * px]
    rows = [H, HC, HC]
    mid, amp = 1 << (depth - 1), 20 << (depth - 8)
    if deblock:
        dst0 = [(mid + rng.integers(-amp, amp + 1, (rows[pl], strides[pl] // px))).astype(dt) for pl in range(3)]
    else:
        dst0 = [rng.integers(0, top, (rows[pl], strides[pl] // px), dtype=dt) for pl in range(3)]
    refs = [rng.integers(0, top, (nref * rows[pl], strides[pl] // px), dtype=dt) for pl in range(3)]
    want, got = [a.copy() for a in dst0], [a.copy() for a in dst0]
    cpu = I.Dec(R, "ffref_", depth, mb_w, mb_h, strides[0], strides[1], 0, cfmt=cfmt)
    rec = I.Dec(RH, "ffrefhip_", depth, mb_w, mb_h, strides[0], strides[1], 1, cfmt=cfmt)
    cpu.set_cur([a.ctypes.data for a in want])
    rec.set_cur([a.ctypes.data for a in got])
    INTERLACED = cpu.bits[14]
    for ps in (1, 2):                                       # PICT_TOP_FIELD, PICT_BOTTOM_FIELD
        bottom = ps == 2
        cpu.set_field(ps)
        rec.set_field(ps)
        pw = I.make_pwt(rng, weights, depth, nref)
        cpu.set_pwt(pw)
        rec.set_pwt(pw)
        _field_refs(rng, [(cpu, [r.ctypes.data for r in refs]), (rec, [r.ctypes.data for r in refs])], refs, rows, strides, nref)
        pic = HostPicture(_lib, L, mb_w, fmb_h, depth, cfmt)
        RH.ffrefhip_h264dec_record_begin(rec.d, pic.p, *[(got if deblock else refs)[pl].ctypes.data for pl in range(3)])
        if deblock:
            for st in I.make_filter_picture(rng, cpu.bits, mb_w, fmb_h, depth, p_intra, extra_type=INTERLACED):
                cpu.filter_mb(st["mb_x"], 2 * st["mb_y"] + bottom, st)
                rec.filter_mb(st["mb_x"], 2 * st["mb_y"] + bottom, st)
        else:
            for fy in range(fmb_h):
                for mx in range(mb_w):
                    if rng.random() < p_intra:
                        d = G.make_intra_mb(rng, mx, fy, mb_w, fmb_h, depth=depth, cfmt=cfmt)
                        d["mb_y"] = 2 * fy + bottom
                        a, b = cpu.decode_intra(d), rec.decode_intra(d)
                        assert d["type"] == G.PCM or np.array_equal(a, b)
                    else:
                        m = I.make_inter_mb(rng, cpu.bits, mx, 2 * fy + bottom, nref, mvr, depth=depth, cfmt=cfmt, extra_type=INTERLACED)
                        assert np.array_equal(cpu.decode_inter(m), rec.decode_inter(m))
        ls = pic.lists()
        assert (ls.mb_w, ls.mb_h) == (mb_w, fmb_h)
        # the field as a picture: the frame plane's address (+ one line for the bottom field), twice the line size
        dp = (C.c_void_p * 3)(*[got[pl].ctypes.data + bottom * strides[pl] for pl in range(3)])
        rp = (C.c_void_p * 3)(*[(got if deblock else refs)[pl].ctypes.data for pl in range(3)])
        st2 = (C.c_int * 3)(*[2 * s for s in strides])
        assert E.ffemul_h264_picture_flush(C.byref(ls), dp, st2, rp) == 0
        pic.close()
        for pl in range(3):                                  # the other field's lines: untouched so far / still what the first pass left
            assert np.array_equal(got[pl][1 - bottom::2], want[pl][1 - bottom::2])
    for pl in range(3):
        assert (want[pl][0::2] != dst0[pl][0::2]).sum() > 40 and (want[pl][1::2] != dst0[pl][1::2]).sum() > 40
        bad = got[pl] != want[pl]
        assert not bad.any(), "plane %d: %d mismatches, first at %s" % (pl, bad.sum(), np.argwhere(bad)[0])
    cpu.close()
    rec.close()


@pytest.mark.parametrize("depth,mb_w,fmb_h,nref,mvr,p_intra,weights,cfmt", [
    (8, 6, 3, 2, 40, 0.0, 0, 1), (8, 11, 4, 3, 2000, 0.0, 1, 1), (8, 11, 4, 3, 300, 0.0, 2, 1), (8, 9, 3, 1, 64, 1.0, 0, 1), (8, 20, 6, 2, 120, .15, 1, 1),
    (10, 7, 3, 2, 600, .2, 1, 1), (8, 9, 4, 2, 300, .2, 2, 3), (10, 6, 3, 2, 500, .3, 1, 3)])
def test_recorded_field_pictures_executed_on_cpu_equal_reference(depth, mb_w, fmb_h, nref, mvr, p_intra, weights, cfmt):
    """PAFF: ff_h264_hl_decode_mb() on field macroblocks (mb_linesize = 2 * linesize, block_offset[48..], pic_height halved and the chroma
    vector offset between fields of opposite parity in mc_dir_part(), h264_mb.c:229,289-293; the implicit weights' [mb_y & 1]) over the
    recording members, each field flushed as a picture of its own"""
    _run_field_frame(depth, mb_w, fmb_h, nref, mvr, p_intra, weights, cfmt, seed=7770000 + depth * 1000 + mb_w * 31 + mvr + weights + cfmt)


@pytest.mark.parametrize("depth,mb_w,fmb_h,p_intra,cfmt", [(8, 6, 3, .2, 1), (8, 20, 6, .15, 1), (8, 9, 3, 1.0, 1), (10, 7, 3, .2, 1), (8, 9, 4, .2, 3)])
def test_recorded_field_deblocking_executed_on_cpu_equals_reference(depth, mb_w, fmb_h, p_intra, cfmt):
    """ff_h264_filter_mb() in field pictures (the row above is the field's own, bS 3 on horizontal intra edges: h264_loopfilter.c:550,
    mvy_limit 2): each field's edge tables through the frame-order filter on that field's lines"""
    _run_field_frame(depth, mb_w, fmb_h, 1, 0, p_intra, 0, cfmt, seed=7780000 + depth * 100 + mb_w + fmb_h + cfmt, deblock=True)


# ---- 4:2:2 -------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("depth,mb_w,mb_h,nref,mvr,p_intra,weights", [
    (8, 6, 4, 2, 40, 0.0, 0), (8, 11, 7, 3, 2000, 0.0, 1), (8, 11, 7, 3, 300, 0.0, 2), (8, 9, 5, 1, 64, 1.0, 0), (8, 20, 11, 2, 120, .15, 1),
    (10, 7, 5, 2, 600, .2, 1), (12, 6, 4, 2, 500, 1.0, 0), (14, 6, 4, 2, 400, .2, 2)])
def test_recorded_picture_executed_on_cpu_equals_reference_422(depth, mb_w, mb_h, nref, mvr, p_intra, weights):
    """chroma_format_idc 2: hl_motion_422 (chroma blocks of the luma height, the vertical chroma vector at quarter-sample scale:
    mc_dir_part(), h264_mb.c:289-317), idct_add8_422 after chroma422_dc_dequant_idct (h264idct_template.c:230-252,295-321), intra chroma by
    the pred8x16 functions with eight residual blocks per plane (the record FFHipH264IntraC422 through k_h264_intra_c422's phase body)"""
    _run_picture(depth, mb_w, mb_h, nref, mvr, p_intra, weights, 2, seed=4220000 + depth * 1000 + mb_w * 31 + mvr + weights)


@pytest.mark.parametrize("depth,mb_w,mb_h,p_intra", [(8, 6, 4, .2), (8, 20, 11, .15), (8, 9, 5, 1.0), (10, 7, 5, .2), (12, 6, 4, .3)])
def test_recorded_deblocking_422_executed_on_cpu_equals_reference(depth, mb_w, mb_h, p_intra):
    """ff_h264_filter_mb() at 4:2:2: six chroma edges per macroblock and plane — h_loop_filter_chroma422 on x = 0, 4 over 16 lines,
    v_loop_filter_chroma on y = 0, 4, 8, 12 (filter_mb_dir(), h264_loopfilter.c:601-703)"""
    test_recorded_deblocking_executed_on_cpu_equals_reference(depth, mb_w, mb_h, p_intra, 2)


@pytest.mark.parametrize("depth,mb_w,fmb_h,nref,mvr,p_intra,weights", [(8, 9, 4, 2, 300, .2, 2), (10, 6, 3, 2, 500, .3, 1)])
def test_recorded_field_pictures_422(depth, mb_w, fmb_h, nref, mvr, p_intra, weights):
    _run_field_frame(depth, mb_w, fmb_h, nref, mvr, p_intra, weights, 2, seed=7772000 + depth * 1000 + mb_w * 31 + mvr + weights)


@pytest.mark.parametrize("depth,mb_w,mb_h,nref,mvr,p_intra,weights", [(8, 6, 4, 2, 40, 0.0, 0), (8, 11, 7, 3, 600, .2, 2), (8, 9, 5, 1, 64, 1.0, 0), (10, 7, 5, 2, 300, .3, 1)])
def test_recorded_monochrome_picture_executed_on_cpu_equals_reference(depth, mb_w, mb_h, nref, mvr, p_intra, weights):
    """chroma_format_idc 0: the reference reconstructs a monochrome picture as 4:2:0 with mid-grey chroma through the ordinary members
    (DC_128 chroma prediction, chroma MC from mid-grey references, no chroma residual; I_PCM sets the chroma samples to 1 << (bit_depth - 1):
    h264_mb_template.c:112-148) — the same picture object, the recorder appending the mid-grey I_PCM fields"""
    _run_picture(depth, mb_w, mb_h, nref, mvr, p_intra, weights, 0, seed=4000000 + depth * 1000 + mb_w * 31 + mvr + weights)


@pytest.mark.parametrize("mb_w,mb_h,nref,mvr,p_intra,weights,cfmt,profile", [
    (6, 4, 2, 40, .4, 0, 1, 100), (6, 4, 2, 40, .4, 0, 1, 244), (9, 5, 1, 64, 1.0, 0, 1, 244), (11, 7, 3, 600, .3, 2, 1, 244),
    (6, 4, 2, 40, .4, 0, 3, 244), (9, 5, 1, 64, 1.0, 0, 3, 244), (11, 7, 3, 600, .3, 1, 3, 100), (8, 5, 2, 120, .5, 0, 0, 244),
    (6, 4, 2, 40, .4, 0, 2, 244), (9, 5, 1, 64, 1.0, 0, 2, 244), (11, 7, 3, 600, .3, 2, 2, 100)])
def test_recorded_lossless_picture_executed_on_cpu_equals_reference(mb_w, mb_h, nref, mvr, p_intra, weights, cfmt, profile):
    """The lossless transform bypass with test-written macroblock state (round 6; 8 bits): ff_h264_hl_decode_mb() compiled in place, about
    half the macroblocks with QP'Y = 0 in a stream with sps->transform_bypass — 4:2:0, monochrome, 4:2:2 and 4:4:4 (hl_decode_mb_444: the luma
    forms on all three planes: the case the whole-decoder streams of tests/test_h264_stream_cpu.py do not reach), profile_idc 100 (the
    residual added as samples) and 244 (DPCM for vertically / horizontally predicted blocks: pred4x4_add, pred8x8l_filter_add,
    pred16x16_add, pred8x8_add).  The recorder's lists on the CPU list executor == the reference's reconstruction."""
    _run_picture(8, mb_w, mb_h, nref, mvr, p_intra, weights, cfmt, seed=5000000 + mb_w * 31 + mvr + weights + cfmt * 7 + profile, bypass=profile)


def test_recorder_refuses_what_stays_on_the_c_path_and_stays_refused():
    """ff_h264_hip_hl_decode_mb() / ff_h264_hip_filter_mb(): a field macroblock in a picture whose recording began as a frame (what an
    MBAFF pair brings: MB_FIELD(sl) != the recorder's field flag) answers FFHIP_ENOSYS, and the error sticks — the picture stays on the C path as a
    whole, later macroblocks are not recorded into a half-described picture"""
    _lib, L, R, RH, E = _env()
    rng = np.random.default_rng(77)
    mb_w, mb_h, depth = 4, 3, 8
    W, H = mb_w * 16, mb_h * 16
    strides, rows = [W, W // 2 + 16, W // 2 + 16], [H, H // 2, H // 2]
    refs = [rng.integers(0, 256, (rows[pl], strides[pl]), dtype=np.uint8) for pl in range(3)]
    got = [np.zeros((rows[pl], strides[pl]), np.uint8) for pl in range(3)]
    rec = I.Dec(RH, "ffrefhip_", depth, mb_w, mb_h, strides[0], strides[1], 1)
    for lst in (0, 1):
        rec.set_ref(lst, 0, [r.ctypes.data for r in refs])
    rec.set_pwt(I.make_pwt(rng, 0, depth, 1))
    rec.set_cur([a.ctypes.data for a in got])
    pic = HostPicture(_lib, L, mb_w, mb_h, depth, 1)
    RH.ffrefhip_h264dec_record_begin(rec.d, pic.p, *[r.ctypes.data for r in refs])
    call = rec.fn("h264dec_decode_inter")
    def run(m):
        mb = m["mb"].copy()
        return call(rec.d, m["mb_x"], m["mb_y"], m["mb_type"], m["sub_mb_type"].ctypes.data, m["mv_cache"].ctypes.data, m["ref_cache"].ctypes.data,
                    m["cbp"], m["nnzc"].ctypes.data, mb.ctypes.data, m["qmul_cb"], m["qmul_cr"])
    assert run(I.make_inter_mb(rng, rec.bits, 0, 0, 1, 40)) == 0
    before = pic.lists()
    n0 = sum(before.nqpel[0][s] for s in range(3))
    assert n0 > 0
    rec.set_field(1)                                              # field macroblocks from here on (sl->mb_field_decoding_flag), as in an MBAFF pair
    assert run(I.make_inter_mb(rng, rec.bits, 1, 0, 1, 40, extra_type=rec.bits[14])) == _lib.ENOSYS
    rec.set_field(3)
    assert run(I.make_inter_mb(rng, rec.bits, 2, 0, 1, 40)) == _lib.ENOSYS                                 # ... and nothing after it
    assert sum(pic.lists().nqpel[0][s] for s in range(3)) == n0
    for a in got:
        assert not a.any()
    pic.close()
    rec.close()
