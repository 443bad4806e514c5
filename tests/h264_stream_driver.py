"""Drives oracle/_ref/libffref_h264dec.so — the reference's WHOLE H.264 decoder (oracle/refbuild/ffref_shim_h264dec.c) — over streams
from tests/h264_bitstream.py: once plain, once with the `hip` recorder installed at the decoder's two call sites.  In record mode a
finished picture's lists are executed by `flush` on the decoder's picture arena: the CPU list executor (oracle/emul_h264_picture.cpp)
here, ffhip_h264_picture_flush() on a device mirror in tests/test_gpu_h264_stream.py."""
import ctypes as C
import os

import numpy as np

import ffi
import h264_bitstream as B

DEC_SO = os.path.join(ffi.ROOT, "oracle", "_ref", "libffref_h264dec.so")
EMUL_SO = os.path.join(ffi.ROOT, "oracle", "libffemul.so")
FLUSH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int)
# an MBAFF frame: FLUSH_FN three times (field = 2: the frame macroblocks' object, 3: a field parity's), then this with the FFHipH264Mbaff object
FLUSH_MBAFF_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_int, C.c_int)
_dec = None


def have():
    return os.path.exists(DEC_SO) and os.path.exists(EMUL_SO)


def dec():
    global _dec
    if _dec is None:
        from ffmpeg_amd import _lib
        _lib.lib()                                  # libffhip.so first: the decoder library binds to the instance the package uses
        L = C.CDLL(DEC_SO)
        L.ffref_h264stream_open.restype = C.c_void_p
        L.ffref_h264stream_open.argtypes = [C.c_int, C.c_size_t]
        L.ffref_h264stream_set_flush.argtypes = [C.c_void_p, FLUSH_FN, C.c_void_p]
        L.ffref_h264stream_set_flush.restype = None
        L.ffref_h264stream_set_flush_mbaff.argtypes = [C.c_void_p, FLUSH_MBAFF_FN, C.c_void_p]
        L.ffref_h264stream_set_flush_mbaff.restype = None
        L.ffref_h264stream_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.ffref_h264stream_nframes.argtypes = [C.c_void_p]
        L.ffref_h264stream_stat.argtypes = [C.c_void_p, C.c_int]
        L.ffref_h264stream_stat.restype = C.c_long
        L.ffref_h264stream_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                             C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ffref_h264stream_arena.restype = C.c_void_p
        L.ffref_h264stream_arena.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.ffref_h264stream_set_base_shift.argtypes = [C.c_void_p, C.c_int64]
        L.ffref_h264stream_set_base_shift.restype = None
        L.ffref_h264stream_close.argtypes = [C.c_void_p]
        L.ffref_h264stream_close.restype = None
        _dec = L
    return _dec


class Lists(C.Structure):                       # == FFHipH264PictureLists (include/ffhip.h)
    _fields_ = [("mb_w", C.c_int), ("mb_h", C.c_int), ("bit_depth", C.c_int), ("chroma_format_idc", C.c_int),
                ("qpel", (C.c_void_p * 3) * 3), ("nqpel", (C.c_int * 3) * 3), ("cmc", (C.c_void_p * 3) * 2), ("ncmc", (C.c_int * 3) * 2),
                ("wt", C.c_void_p * 3), ("nwt", C.c_int * 3), ("idct_off", (C.c_void_p * 4) * 3), ("idct_coef", (C.c_void_p * 4) * 3),
                ("nidct", (C.c_int * 4) * 3), ("intra", C.c_void_p * 3), ("nintra", C.c_int * 3), ("intra_coef", C.c_void_p * 3),
                ("nintra_coef", C.c_int * 3), ("edges", C.c_void_p * 3), ("intra_c422", C.c_void_p), ("nintra_c422", C.c_int),
                ("intra_c422_coef", C.c_void_p), ("nintra_c422_coef", C.c_int),
                ("addpx_off", (C.c_void_p * 2) * 3), ("addpx_coef", (C.c_void_p * 2) * 3), ("naddpx", (C.c_int * 2) * 3)]


class MbaffLists(C.Structure):                  # == FFHipH264MbaffLists (include/ffhip.h)
    _fields_ = [("mb_w", C.c_int), ("mb_h", C.c_int), ("recs", C.c_void_p), ("geo", C.c_void_p), ("coefs", C.c_void_p), ("intra_row", C.c_void_p),
                ("nrecs", C.c_int32), ("ncoefs", C.c_int32), ("calls", C.c_void_p * 3), ("pair_end", C.c_void_p * 3), ("ncalls", C.c_int32 * 3),
                ("bit_depth", C.c_int)]


def cpu_flush(arena_base):
    """flush = the recorded lists executed on the host arena (at address arena_base) by the oracle's list executor; returns the callable and
    the counters it keeps"""
    from ffmpeg_amd import _lib
    L = _lib.lib()
    E = C.CDLL(EMUL_SO)
    E.ffemul_h264_picture_flush.argtypes = [C.c_void_p] * 4
    E.ffemul_h264_mbaff_flush.argtypes = [C.c_void_p] * 3
    L.ffhip_h264_mbaff_lists.argtypes = [C.c_void_p, C.c_void_p]
    counts = {"pictures": 0, "inter_blocks": 0, "intra_mbs": 0, "edge_planes": 0, "mbaff_frames": 0, "mbaff_intra_mbs": 0, "mbaff_field_intra_mbs": 0,
              "mbaff_calls": 0, "mbaff_calls_field_stride": 0, "mbaff_calls_mbaff_member": 0, "mbaff_field_inter_blocks": 0}

    def flush(opaque, pic, dst_off, stride, mb_w, mb_h, field):
        ls = Lists()
        if L.ffhip_h264_picture_lists(pic, C.byref(ls)) != 0:
            return -1
        dp = (C.c_void_p * 3)(*[arena_base + dst_off[i] for i in range(3)])
        rp = (C.c_void_p * 3)(arena_base, arena_base, arena_base)
        st = (C.c_int * 3)(stride[0], stride[1], stride[2])
        counts["pictures"] += not (field & 2)         # (an MBAFF frame counts once, in flush_mbaff)
        counts["inter_blocks"] += sum(ls.nqpel[0][k] for k in range(3))
        counts["mbaff_field_inter_blocks"] += sum(ls.nqpel[0][k] for k in range(3)) if field == 3 else 0
        counts["intra_mbs"] += ls.nintra[0]
        if ls.nintra[0]:   # FFHipH264IntraMB: flags at byte 40 (bit 3: the transform bypass), pad[0] at 41: DPCM-coded regions
            raw = np.ctypeslib.as_array((C.c_uint8 * (108 * ls.nintra[0])).from_address(ls.intra[0])).reshape(-1, 108)
            counts["bypass_intra_mbs"] = counts.get("bypass_intra_mbs", 0) + int(((raw[:, 40] >> 3) & 1).sum())
            counts["dpcm_regions"] = counts.get("dpcm_regions", 0) + int(raw[:, 41].sum())
        counts["bypass_inter_blocks"] = counts.get("bypass_inter_blocks", 0) + sum(ls.naddpx[pl][k] for pl in range(3) for k in range(2))
        counts["edge_planes"] += sum(1 for k in range(3) if ls.edges[k])
        return E.ffemul_h264_picture_flush(C.byref(ls), dp, st, rp)

    def flush_mbaff(opaque, chains, dst_off, stride, mb_w, mb_h):
        ml = MbaffLists()
        if L.ffhip_h264_mbaff_lists(chains, C.byref(ml)) != 0:
            return -1
        dp = (C.c_void_p * 3)(*[arena_base + dst_off[i] for i in range(3)])
        st = (C.c_int * 3)(stride[0], stride[1], stride[2])
        counts["pictures"] += 1
        counts["mbaff_frames"] += 1
        counts["mbaff_intra_mbs"] += ml.nrecs
        if ml.nrecs:
            geo = np.ctypeslib.as_array((C.c_uint32 * ml.nrecs).from_address(ml.geo))
            counts["mbaff_field_intra_mbs"] += int(((geo >> 24) & 1).sum())
        for pl in range(3):
            counts["mbaff_calls"] += ml.ncalls[pl]
            if ml.ncalls[pl]:
                raw = np.ctypeslib.as_array((C.c_uint8 * (12 * ml.ncalls[pl])).from_address(ml.calls[pl])).reshape(-1, 12)
                counts["mbaff_calls_field_stride"] += int((raw[:, 7] & 1).sum())
                counts["mbaff_calls_mbaff_member"] += int(((raw[:, 7] >> 1) & 1).sum())
        return E.ffemul_h264_mbaff_flush(C.byref(ml), dp, st)
    flush.mbaff = flush_mbaff
    return flush, counts


def decode(aus, make_flush=None, arena_bytes=48 << 20, read_back=None, base_shift=0):
    """Decodes the access units; make_flush(arena_base, arena_bytes) -> (flush callable, counters) switches the recorder on.  read_back(
    arena_base, used): called after the drain and before the frames are copied out (the GPU tier downloads its device mirror there).
    Returns (frames, stats, counters): frames = per output frame three numpy planes (copies)."""
    L = dec()
    s = L.ffref_h264stream_open(int(make_flush is not None), arena_bytes)
    assert s
    keep, keep_m, counts = None, None, None
    try:
        base = L.ffref_h264stream_arena(s, None)
        if base_shift:
            L.ffref_h264stream_set_base_shift(s, base_shift)
        if make_flush is not None:
            fn, counts = make_flush(base, arena_bytes)
            keep = FLUSH_FN(fn)
            L.ffref_h264stream_set_flush(s, keep, None)
            if getattr(fn, "mbaff", None) is not None:
                keep_m = FLUSH_MBAFF_FN(fn.mbaff)
                L.ffref_h264stream_set_flush_mbaff(s, keep_m, None)
        for a in aus:
            r = L.ffref_h264stream_decode(s, a, len(a))
            assert r == 0, "avcodec_send_packet / receive_frame: %d" % r
        assert L.ffref_h264stream_decode(s, None, 0) == 0
        stats = {k: L.ffref_h264stream_stat(s, i) for i, k in enumerate(("pictures", "mbs_hl", "mbs_filter", "refused", "errors",
                                                                        "first_error", "damaged", "plain_pictures", "mbs_bipred", "mbs_direct", "mbs_8x8dct", "mbs_weighted",
                                                                        "mbs_implicit", "mbs_b", "mbs_intra8x8", "mbs_field", "mbaff_pictures", "mbs_bypass"))}
        if read_back is not None:
            used = C.c_size_t()
            L.ffref_h264stream_arena(s, C.byref(used))
            read_back(base, used.value)
        frames = []
        for i in range(L.ffref_h264stream_nframes(s)):
            off = (C.c_int64 * 3)()
            ls = (C.c_int * 3)()
            w, h, bd = C.c_int(), C.c_int(), C.c_int()
            assert L.ffref_h264stream_frame(s, i, off, ls, C.byref(w), C.byref(h), C.byref(bd)) == 0
            planes = []
            for pl in range(3):
                rows, cols = (h.value, w.value) if pl == 0 else ((h.value + 1) >> 1, (w.value + 1) >> 1)
                a = np.ctypeslib.as_array((C.c_uint8 * (ls[pl] * rows)).from_address(base + off[pl])).reshape(rows, ls[pl])
                a = a.view(np.uint16) if bd.value > 8 else a
                planes.append(a[:, :cols].copy())
            frames.append(planes)
        return frames, stats, counts
    finally:
        L.ffref_h264stream_close(s)


# ---- the streams -----------------------------------------------------------------------------------------------------------------------
def stream_ip(bit_depth=8, seed=1, mb_w=6, mb_h=5, n=5):
    """I P P P P, one slice per picture, deblocking across the picture (idc 0) with varying offsets, 1..3 references"""
    p = B.Params(mb_w=mb_w, mb_h=mb_h, bit_depth=bit_depth, seed=seed)
    w = B.StreamWriter(p)
    rng = np.random.default_rng(seed + 100)
    pics = [{"type": "I", "slices": [0], "deblock": [(0, 0, 0)]}]
    for k in range(1, n):
        pics.append({"type": "P", "slices": [0], "deblock": [(0, int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))], "num_ref": min(k, 3)})
    return w.stream(pics), w.stats


def stream_slices(bit_depth=8, seed=2, mb_w=7, mb_h=6, n=5):
    """two and three slices per picture, starting mid-row; disable_deblocking_filter_idc 0, 1 and 2 mixed within a picture"""
    p = B.Params(mb_w=mb_w, mb_h=mb_h, bit_depth=bit_depth, seed=seed, init_qp=24, chroma_qp_offset=2)
    w = B.StreamWriter(p)
    rng = np.random.default_rng(seed + 200)
    n_mb = mb_w * mb_h
    pics = []
    for k in range(n):
        ns = 2 + (k % 2)
        cuts = sorted(int(v) for v in rng.choice(np.arange(3, n_mb - 3), size=ns - 1, replace=False))
        idcs = [(0, 2, 1), (2, 0, 2), (1, 2, 0)][k % 3]
        pics.append({"type": "I" if k == 0 else "P", "slices": [0] + cuts,
                     "deblock": [(idcs[i % 3], int(rng.integers(-2, 3)), int(rng.integers(-2, 3))) for i in range(ns)], "num_ref": min(max(k, 1), 2)})
    return w.stream(pics), w.stats


def stream_fields(bit_depth=8, seed=3, mb_w=6, mb_h=6, n=3):
    """PAFF: every frame as two field pictures (top then bottom); the second field of the first frame is a P field on its first field"""
    p = B.Params(mb_w=mb_w, mb_h=mb_h, bit_depth=bit_depth, frame_mbs_only=0, seed=seed)
    w = B.StreamWriter(p)
    pics = []
    for k in range(n):
        pics.append({"type": "I" if k == 0 else "P", "slices": [0], "deblock": [(0, 0, 0)], "field": "top", "num_ref": min(max(2 * k, 1), 4)})
        pics.append({"type": "P", "slices": [0, 5], "deblock": [(0, 1, 1), (2, -1, 0)], "field": "bottom", "second_field": True,
                     "num_ref": min(2 * k + 1, 4)})
    return w.stream(pics), w.stats


def stream_mbaff_and_fields(seed=7, mb_w=6, mb_h=6):
    """mb_adaptive_frame_field_flag = 1: frames are MBAFF frames (frame and field macroblock pairs mixed), field pictures are plain
    fields.  I (MBAFF frame), P field pair, P (MBAFF frame), P field pair."""
    p = B.Params(mb_w=mb_w, mb_h=mb_h, frame_mbs_only=0, mbaff=1, seed=seed)
    w = B.StreamWriter(p)
    pics = [{"type": "I", "slices": [0], "deblock": [(0, 0, 0)]},
            {"type": "P", "slices": [0], "deblock": [(0, 1, 0)], "field": "top", "num_ref": 2},
            {"type": "P", "slices": [0, 7], "deblock": [(0, 0, 0), (2, 1, -1)], "field": "bottom", "second_field": True, "num_ref": 3},
            {"type": "P", "slices": [0, 12], "deblock": [(0, 1, 1), (0, 0, 0)], "num_ref": 2},
            {"type": "P", "slices": [0], "deblock": [(0, -1, 2)], "field": "top", "num_ref": 4},
            {"type": "P", "slices": [0], "deblock": [(0, 0, 0)], "field": "bottom", "second_field": True, "num_ref": 4}]
    return w.stream(pics), w.stats


# ---- round 6: B pictures, weighted prediction, the 8x8 transform, 4:2:2 ---------------------------------------------------------------------
def stream_b(bit_depth=8, seed=21, mb_w=6, mb_h=5, direct_spatial=1, weighted_bipred=0, weighted_pred=0, t8x8=0, chroma_format=1,
             slices=1, nonref=False, gops=2, mbaff=0, lossless=0):
    """I P B B B | P B B B ...: pic_order_cnt_type 0, B pictures between their references in output order and decoded after them (a small
    pyramid: the middle B is itself a reference of the outer two), every B macroblock type, direct_spatial_mv_pred_flag as given.
    nonref: the outer B pictures of each group are non-reference pictures (nal_ref_idc 0)."""
    p = B.Params(mb_w=mb_w, mb_h=mb_h, bit_depth=bit_depth, seed=seed, num_ref_frames=4, poc_type=0, reorder=3, t8x8=t8x8,
                 weighted_pred=weighted_pred, weighted_bipred=weighted_bipred, chroma_format=chroma_format, frame_mbs_only=0 if mbaff else 1,
                 mbaff=mbaff, lossless=lossless)
    w = B.StreamWriter(p)
    rng = np.random.default_rng(seed + 300)
    n_mb = mb_w * mb_h

    def sl():
        if slices == 1:
            return [0], [(0, int(rng.integers(-2, 3)), int(rng.integers(-2, 3)))]
        cuts = sorted(int(v) for v in rng.choice(np.arange(3, n_mb - 3), size=slices - 1, replace=False))
        if mbaff:
            cuts = sorted(set(c & ~1 for c in cuts))   # a slice of an MBAFF frame starts on a macroblock pair
        return [0] + cuts, [((0, 2, 1)[i % 3], int(rng.integers(-2, 3)), int(rng.integers(-2, 3))) for i in range(slices)]
    pics, nref = [], 0

    def add(t, poc, ref=True, **kw):
        nonlocal nref
        s_, d_ = sl()
        avail = max(1, min(nref, p.num_ref_frames))
        d = {"type": t, "slices": s_, "deblock": d_, "poc": poc, "ref": ref, "num_ref": min(avail, kw.pop("n0", 3)),
             "num_ref_l1": min(avail, kw.pop("n1", 2)), "direct_spatial": direct_spatial}
        d.update(kw)
        pics.append(d)
        nref += bool(ref)
    add("I", 0)
    base = 0
    for g in range(gops):
        add("P", base + 16, n0=2)
        add("B", base + 8)
        add("B", base + 4, ref=not nonref)
        add("B", base + 12, ref=not nonref)
        base += 16
    return w.stream(pics), w.stats


def stream_p_features(bit_depth=8, seed=31, mb_w=6, mb_h=5, weighted_pred=0, t8x8=0, chroma_format=1, n=5, fields=False, lossless=0):
    """I P P P P (pic_order_cnt_type 2) with PPS features switched on: weighted_pred_flag (a pred_weight_table per P slice),
    transform_8x8_mode_flag (Intra8x8, transform_size_8x8_flag on inter macroblocks), chroma_format_idc 2"""
    p = B.Params(mb_w=mb_w, mb_h=mb_h, bit_depth=bit_depth, seed=seed, weighted_pred=weighted_pred, t8x8=t8x8, chroma_format=chroma_format,
                 frame_mbs_only=0 if fields else 1, lossless=lossless)
    w = B.StreamWriter(p)
    rng = np.random.default_rng(seed + 400)
    pics = []
    if fields:
        for k in range(n):
            pics.append({"type": "I" if k == 0 else "P", "slices": [0], "deblock": [(0, 0, 0)], "field": "top", "num_ref": min(max(2 * k, 1), 4)})
            pics.append({"type": "P", "slices": [0, 5], "deblock": [(0, 1, 1), (2, -1, 0)], "field": "bottom", "second_field": True,
                         "num_ref": min(2 * k + 1, 4)})
        return w.stream(pics), w.stats
    pics.append({"type": "I", "slices": [0], "deblock": [(0, 0, 0)]})
    for k in range(1, n):
        pics.append({"type": "P", "slices": [0, 11] if k % 2 else [0], "deblock": [(0, int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), (2, 0, 1)][:2 if k % 2 else 1],
                     "num_ref": min(k, 3)})
    return w.stream(pics), w.stats


def stream_mbaff_p(seed=41, mb_w=6, mb_h=6, n=5, t8x8=0, weighted_pred=0, lossless=0, bit_depth=8, chroma_format=1):
    """I P P P P, every picture an MBAFF frame: frame and field macroblock pairs mixed (mb_field_decoding_flag per pair), one to three slices
    starting on macroblock pairs, disable_deblocking_filter_idc 0 / 1 / 2, Intra16x16 with every prediction mode its neighbours allow and
    isolated I_NxN macroblocks, one to three reference frames (a field macroblock: twice as many reference fields)"""
    p = B.Params(mb_w=mb_w, mb_h=mb_h, frame_mbs_only=0, mbaff=1, seed=seed, t8x8=t8x8, weighted_pred=weighted_pred, lossless=lossless,
                 bit_depth=bit_depth, chroma_format=chroma_format)
    w = B.StreamWriter(p)
    rng = np.random.default_rng(seed + 500)
    n_mb = mb_w * mb_h
    pics = []
    for k in range(n):
        ns = 1 + (k % 3)
        cuts = sorted(set(int(v) & ~1 for v in rng.choice(np.arange(4, n_mb - 4), size=ns - 1, replace=False)))
        idcs = [(0, 2, 1), (0, 0, 2), (2, 1, 0)][k % 3]
        pics.append({"type": "I" if k == 0 else "P", "slices": [0] + cuts,
                     "deblock": [(idcs[i % 3], int(rng.integers(-2, 3)), int(rng.integers(-2, 3))) for i in range(len(cuts) + 1)],
                     "num_ref": min(max(k, 1), 3)})
    return w.stream(pics), w.stats


# MBAFF frames (8 bits, 4:2:0): name -> (generator, kwargs, pictures, DECODER statistics that must be non-zero, counters of the list executor
# that must be non-zero)
MBAFF_CASES = {
    "mbaff_p": (stream_mbaff_p, dict(seed=41), 5, ("mbs_field",), ("mbaff_field_intra_mbs", "mbaff_calls_field_stride", "mbaff_calls_mbaff_member",
                                                                   "mbaff_field_inter_blocks")),
    "mbaff_p_8x8_weighted": (stream_mbaff_p, dict(seed=42, t8x8=1, weighted_pred=1), 5, ("mbs_field", "mbs_8x8dct", "mbs_weighted"),
                             ("mbaff_field_intra_mbs", "mbaff_calls_mbaff_member")),
    "mbaff_p_wide": (stream_mbaff_p, dict(seed=43, mb_w=11, mb_h=8, n=4), 4, ("mbs_field",), ("mbaff_field_intra_mbs", "mbaff_calls_mbaff_member")),
    "mbaff_p_cif": (stream_mbaff_p, dict(seed=48, mb_w=22, mb_h=18, n=3, t8x8=1), 3, ("mbs_field", "mbs_8x8dct"), ("mbaff_field_intra_mbs", "mbaff_calls_mbaff_member")),
    "mbaff_p_10": (stream_mbaff_p, dict(seed=61, bit_depth=10, t8x8=1), 5, ("mbs_field", "mbs_8x8dct"), ("mbaff_field_intra_mbs", "mbaff_calls_mbaff_member",
                                                                                                 "mbaff_field_inter_blocks")),
    "mbaff_p_cif_10": (stream_mbaff_p, dict(seed=62, bit_depth=10, mb_w=22, mb_h=18, n=3, weighted_pred=1), 3, ("mbs_field", "mbs_weighted"),
                       ("mbaff_field_intra_mbs", "mbaff_calls_mbaff_member")),
    "mbaff_b_temporal_implicit_10": (stream_b, dict(seed=63, bit_depth=10, mb_h=6, mbaff=1, direct_spatial=0, weighted_bipred=2, t8x8=1, slices=2), 9,
                                     ("mbs_field", "mbs_b", "mbs_bipred", "mbs_direct", "mbs_implicit"), ("mbaff_calls_mbaff_member",)),
    "mbaff_b_spatial": (stream_b, dict(seed=44, mb_h=6, mbaff=1, direct_spatial=1), 9, ("mbs_field", "mbs_b", "mbs_bipred", "mbs_direct"),
                        ("mbaff_calls_mbaff_member",)),
    "mbaff_b_temporal_implicit": (stream_b, dict(seed=45, mb_h=6, mbaff=1, direct_spatial=0, weighted_bipred=2, slices=2), 9,
                                  ("mbs_field", "mbs_b", "mbs_bipred", "mbs_direct", "mbs_implicit"), ("mbaff_calls_mbaff_member",)),
    "mbaff_b_explicit_8x8": (stream_b, dict(seed=46, mb_h=6, mbaff=1, weighted_bipred=1, weighted_pred=1, t8x8=1), 9,
                             ("mbs_field", "mbs_b", "mbs_weighted", "mbs_8x8dct"), ("mbaff_calls_mbaff_member",)),
}

# the lossless transform bypass (8 bits, 4:2:0): name -> (generator, kwargs, pictures, DECODER statistics that must be non-zero)
LOSSLESS_CASES = {
    "lossless_p_high": (stream_p_features, dict(seed=51, lossless=1), 5, ("mbs_bypass",)),
    "lossless_p_predictive": (stream_p_features, dict(seed=52, lossless=2), 5, ("mbs_bypass",)),
    "lossless_p_predictive_8x8": (stream_p_features, dict(seed=53, lossless=2, t8x8=1), 5, ("mbs_bypass", "mbs_intra8x8", "mbs_8x8dct")),
    "lossless_p_high_8x8": (stream_p_features, dict(seed=54, lossless=1, t8x8=1, weighted_pred=1), 5, ("mbs_bypass", "mbs_intra8x8", "mbs_weighted")),
    "lossless_fields_predictive": (stream_p_features, dict(seed=55, lossless=2, fields=True, n=3, mb_h=6), 6, ("mbs_bypass", "mbs_field")),
    "lossless_b_predictive_8x8": (stream_b, dict(seed=56, lossless=2, t8x8=1, weighted_bipred=2), 9, ("mbs_bypass", "mbs_b", "mbs_bipred", "mbs_8x8dct")),
    "lossless_mbaff_predictive": (stream_mbaff_p, dict(seed=57, lossless=2, t8x8=1), 5, ("mbs_bypass", "mbs_field", "mbaff_pictures")),
    "lossless_422_predictive": (stream_p_features, dict(seed=64, lossless=2, chroma_format=2, t8x8=1), 5, ("mbs_bypass", "mbs_intra8x8")),
    "lossless_422_high": (stream_b, dict(seed=65, lossless=1, chroma_format=2, weighted_bipred=1, weighted_pred=1), 9, ("mbs_bypass", "mbs_b", "mbs_weighted")),
    "lossless_cif_predictive": (stream_p_features, dict(seed=58, lossless=2, t8x8=1, mb_w=22, mb_h=18, n=3), 3, ("mbs_bypass", "mbs_intra8x8")),
}

# name -> (generator, kwargs, pictures, statistics of the DECODER that must be non-zero in record mode, writer statistics that must be non-zero)
ROUND6_CASES = {
    "b_spatial_8": (stream_b, dict(bit_depth=8, seed=21, direct_spatial=1), 9, ("mbs_b", "mbs_bipred", "mbs_direct"), ("b_direct16", "b_16", "b_168", "b_88", "b_direct8", "b_skip")),
    "b_spatial_10": (stream_b, dict(bit_depth=10, seed=22, direct_spatial=1), 9, ("mbs_b", "mbs_bipred", "mbs_direct"), ("b_direct16", "b_88", "b_skip")),
    "b_temporal_8": (stream_b, dict(bit_depth=8, seed=23, direct_spatial=0), 9, ("mbs_b", "mbs_bipred", "mbs_direct"), ("b_direct16", "b_direct8", "b_skip")),
    "b_temporal_10": (stream_b, dict(bit_depth=10, seed=24, direct_spatial=0, slices=2), 9, ("mbs_b", "mbs_bipred", "mbs_direct"), ("b_direct16", "b_direct8", "b_skip")),
    "b_explicit_weights_8": (stream_b, dict(bit_depth=8, seed=25, weighted_bipred=1, weighted_pred=1), 9, ("mbs_b", "mbs_bipred", "mbs_weighted"), ("wp_slices",)),
    "b_explicit_weights_10": (stream_b, dict(bit_depth=10, seed=26, weighted_bipred=1, weighted_pred=1, direct_spatial=0), 9, ("mbs_bipred", "mbs_weighted"), ("wp_slices",)),
    "b_implicit_weights_8": (stream_b, dict(bit_depth=8, seed=27, weighted_bipred=2), 9, ("mbs_b", "mbs_bipred", "mbs_implicit"), ()),
    "b_implicit_weights_10": (stream_b, dict(bit_depth=10, seed=28, weighted_bipred=2, direct_spatial=0), 9, ("mbs_bipred", "mbs_implicit"), ()),
    "b_8x8_transform_8": (stream_b, dict(bit_depth=8, seed=29, t8x8=1), 9, ("mbs_b", "mbs_8x8dct", "mbs_intra8x8"), ("i8", "t8x8_inter")),
    "b_8x8_transform_10": (stream_b, dict(bit_depth=10, seed=30, t8x8=1, direct_spatial=0), 9, ("mbs_b", "mbs_8x8dct", "mbs_intra8x8"), ("i8", "t8x8_inter")),
    "b_nonref_three_slices": (stream_b, dict(bit_depth=8, seed=31, nonref=True, slices=3, gops=3), 13, ("mbs_b", "mbs_bipred"), ("b_88",)),
    "p_weighted_8": (stream_p_features, dict(bit_depth=8, seed=32, weighted_pred=1), 5, ("mbs_weighted",), ("wp_slices",)),
    "p_weighted_fields_10": (stream_p_features, dict(bit_depth=10, seed=33, weighted_pred=1, fields=True, n=3, mb_h=6), 6, ("mbs_weighted",), ("wp_slices",)),
    "p_8x8_transform_8": (stream_p_features, dict(bit_depth=8, seed=34, t8x8=1), 5, ("mbs_8x8dct", "mbs_intra8x8"), ("i8", "t8x8_inter")),
    "p_8x8_transform_10": (stream_p_features, dict(bit_depth=10, seed=35, t8x8=1), 5, ("mbs_8x8dct", "mbs_intra8x8"), ("i8", "t8x8_inter")),
    "high422_p_8": (stream_p_features, dict(bit_depth=8, seed=36, chroma_format=2), 5, ("mbs_hl",), ("p88",)),
    "high422_b_all_10": (stream_b, dict(bit_depth=10, seed=37, chroma_format=2, t8x8=1, weighted_bipred=1, weighted_pred=1), 9,
                         ("mbs_b", "mbs_bipred", "mbs_direct", "mbs_8x8dct", "mbs_weighted", "mbs_intra8x8"), ("i8", "wp_slices")),
    "high422_b_implicit_8": (stream_b, dict(bit_depth=8, seed=38, chroma_format=2, weighted_bipred=2, direct_spatial=0, t8x8=1), 9,
                             ("mbs_b", "mbs_implicit", "mbs_8x8dct"), ("b_direct8",)),
    "b_cif_all_8": (stream_b, dict(bit_depth=8, seed=39, mb_w=22, mb_h=18, t8x8=1, weighted_bipred=2, slices=3), 9,
                    ("mbs_b", "mbs_bipred", "mbs_direct", "mbs_8x8dct", "mbs_implicit", "mbs_intra8x8"), ("b_direct16", "b_direct8", "b_skip")),
}
