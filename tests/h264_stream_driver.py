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
                ("intra_c422_coef", C.c_void_p), ("nintra_c422_coef", C.c_int)]


def cpu_flush(arena_base):
    """flush = the recorded lists executed on the host arena (at address arena_base) by the oracle's list executor; returns the callable and
    the counters it keeps"""
    from ffmpeg_amd import _lib
    L = _lib.lib()
    E = C.CDLL(EMUL_SO)
    E.ffemul_h264_picture_flush.argtypes = [C.c_void_p] * 4
    counts = {"pictures": 0, "inter_blocks": 0, "intra_mbs": 0, "edge_planes": 0}

    def flush(opaque, pic, dst_off, stride, mb_w, mb_h, field):
        ls = Lists()
        if L.ffhip_h264_picture_lists(pic, C.byref(ls)) != 0:
            return -1
        dp = (C.c_void_p * 3)(*[arena_base + dst_off[i] for i in range(3)])
        rp = (C.c_void_p * 3)(arena_base, arena_base, arena_base)
        st = (C.c_int * 3)(stride[0], stride[1], stride[2])
        counts["pictures"] += 1
        counts["inter_blocks"] += sum(ls.nqpel[0][k] for k in range(3))
        counts["intra_mbs"] += ls.nintra[0]
        counts["edge_planes"] += sum(1 for k in range(3) if ls.edges[k])
        return E.ffemul_h264_picture_flush(C.byref(ls), dp, st, rp)
    return flush, counts


def decode(aus, make_flush=None, arena_bytes=48 << 20, read_back=None, base_shift=0):
    """Decodes the access units; make_flush(arena_base, arena_bytes) -> (flush callable, counters) switches the recorder on.  read_back(
    arena_base, used): called after the drain and before the frames are copied out (the GPU tier downloads its device mirror there).
    Returns (frames, stats, counters): frames = per output frame three numpy planes (copies)."""
    L = dec()
    s = L.ffref_h264stream_open(int(make_flush is not None), arena_bytes)
    assert s
    keep, counts = None, None
    try:
        base = L.ffref_h264stream_arena(s, None)
        if base_shift:
            L.ffref_h264stream_set_base_shift(s, base_shift)
        if make_flush is not None:
            fn, counts = make_flush(base, arena_bytes)
            keep = FLUSH_FN(fn)
            L.ffref_h264stream_set_flush(s, keep, None)
        for a in aus:
            r = L.ffref_h264stream_decode(s, a, len(a))
            assert r == 0, "avcodec_send_packet / receive_frame: %d" % r
        assert L.ffref_h264stream_decode(s, None, 0) == 0
        stats = {k: L.ffref_h264stream_stat(s, i) for i, k in enumerate(("pictures", "mbs_hl", "mbs_filter", "refused", "errors",
                                                                        "first_error", "damaged", "plain_pictures"))}
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
