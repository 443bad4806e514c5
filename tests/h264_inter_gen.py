"""Synthetic INTER macroblocks in the decoder's own terms, and the driver of the persistent reference decoder
(oracle/refbuild/ffref_shim_h264mb.c: FFRefH264Dec).  The generator produces decoder STATE only — what the entropy decoder and the
motion-vector prediction leave in H264SliceContext (mb_type / sub_mb_type flags, mv_cache / ref_cache in scan8 layout, the weight
tables of the slice, cbp, the non-zero-count cache, sl->mb) — every decision about which dsp member runs with which operands is made
by the reference's own ff_h264_hl_decode_mb() (libavcodec/h264_mb.c:800), in both libraries:

  libffref.so      (prefix ffref_):     C dsp tables on host planes            -> the expected picture
  libffref_hip.so  (prefix ffrefhip_):  integration/avcodec_h264_picture_hip.c's recording members on device addresses -> libffhip
"""
import ctypes as C
import os

import numpy as np

import h264_intra_gen as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_HIP_SO = os.path.join(ROOT, "oracle", "_ref", "libffref_hip.so")
SCAN8 = G.SCAN8
LIST_NOT_USED = -1
u8p = C.POINTER(C.c_uint8)


def have_ref_hip():
    return os.path.exists(REF_HIP_SO)


class Dec:
    """FFRefH264Dec of one of the two libraries"""

    def __init__(self, lib, prefix, depth, mb_w, mb_h, linesize, uvlinesize, record, cfmt=1):
        self.L, self.p, self.depth = lib, prefix, depth
        f = self.fn
        f("h264dec_open_fmt").restype = C.c_void_p
        f("h264dec_open_fmt").argtypes = [C.c_int] * 7
        f("h264dec_close").argtypes = [C.c_void_p]
        f("h264dec_close").restype = None
        f("h264dec_set_cur").argtypes = [C.c_void_p] * 4
        f("h264dec_set_cur").restype = None
        f("h264dec_set_ref").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        f("h264dec_set_ref").restype = None
        f("h264dec_set_pwt").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        f("h264dec_set_pwt").restype = None
        f("h264dec_decode_inter").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_int]
        f("h264dec_decode_intra").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p,
                                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(lib, prefix + "h264dec_filter_mb"):
            f("h264dec_filter_mb").argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 3
        f("h264dec_set_field").argtypes = [C.c_void_p, C.c_int]
        f("h264dec_set_field").restype = None
        f("h264dec_set_bypass").argtypes = [C.c_void_p, C.c_int, C.c_int]
        f("h264dec_set_bypass").restype = None
        f("h264dec_set_ref_field").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        f("h264dec_set_ref_field").restype = None
        self.d = f("h264dec_open_fmt")(depth, mb_w, mb_h, linesize, uvlinesize, record, cfmt)
        assert self.d
        self.bits = [f("h264dec_mb_type_bits")(i) for i in range(15)]     # [14]: MB_TYPE_INTERLACED

    def fn(self, name):
        return getattr(self.L, self.p + name)

    def close(self):
        if self.d:
            self.fn("h264dec_close")(self.d)
        self.d = None

    def set_cur(self, ptrs):
        self.fn("h264dec_set_cur")(self.d, *ptrs)

    def set_ref(self, lst, idx, ptrs):
        self.fn("h264dec_set_ref")(self.d, lst, idx, *ptrs)

    def set_field(self, picture_structure):
        """1 top field, 2 bottom field (PAFF: every macroblock a field macroblock, mb_y = 2 * row + bottom), 3 back to frames"""
        self.fn("h264dec_set_field")(self.d, picture_structure)

    def set_bypass(self, profile_idc, on):
        """profile_idc != 0: the stream has qpprime_y_zero_transform_bypass_flag (244: with the DPCM forms); on: QP'Y = 0 from here on"""
        self.fn("h264dec_set_bypass")(self.d, profile_idc, int(on))

    def set_ref_field(self, lst, idx, ptrs, parity):
        """the field of parity 1 (top) / 2 (bottom) of the frame whose planes are ptrs"""
        self.fn("h264dec_set_ref_field")(self.d, lst, idx, *ptrs, parity)

    def set_pwt(self, w):
        self.fn("h264dec_set_pwt")(self.d, w["use_weight"], w["use_weight_chroma"], w["luma_denom"], w["chroma_denom"], w["luma"].ctypes.data,
                                   w["chroma"].ctypes.data, w["implicit"].ctypes.data)

    def decode_inter(self, m):
        mb = m["mb"].copy()
        r = self.fn("h264dec_decode_inter")(self.d, m["mb_x"], m["mb_y"], m["mb_type"], m["sub_mb_type"].ctypes.data, m["mv_cache"].ctypes.data,
                                            m["ref_cache"].ctypes.data, m["cbp"], m["nnzc"].ctypes.data, mb.ctypes.data, m["qmul_cb"], m["qmul_cr"])
        assert r == 0, "decode_inter: %d" % r
        return mb

    def decode_intra(self, d):
        mb, dc = d["mb"].copy(), d["luma_dc"].copy()
        r = self.fn("h264dec_decode_intra")(self.d, d["mb_x"], d["mb_y"], d["type"], d["pred16"], d["chroma_pred"], d["pred4"].ctypes.data, d["topleft"],
                                            d["topright"], d["nnzc"].ctypes.data, d["cbp"], mb.ctypes.data, dc.ctypes.data, d["qmul"].ctypes.data,
                                            None if d["pcm"] is None else d["pcm"].ctypes.data)
        assert r == 0, "decode_intra: %d" % r
        return mb

    def filter_mb(self, mb_x, mb_y, st):
        r = self.fn("h264dec_filter_mb")(self.d, mb_x, mb_y, st["ints"].ctypes.data, st["mv_cache"].ctypes.data, st["caches"].ctypes.data)
        assert r == 0, "filter_mb: %d" % r


def make_pwt(rng, kind, depth, nref):
    """the slice's prediction-weight state: kind 0 none, 1 explicit (pred_weight_table(), h264_parse.c:30-118: offsets are scaled by
    1 << (bit_depth - 8)), 2 implicit (implicit_weight_table(), h264_slice.c: 64 - a scaled POC distance, kept when in [-64, 128])"""
    w = dict(use_weight=kind, use_weight_chroma=0, luma_denom=0, chroma_denom=0, luma=np.zeros((48, 2, 2), np.int32),
             chroma=np.zeros((48, 2, 2, 2), np.int32), implicit=np.full((48, 48, 2), 32, np.int32))
    if kind == 1:
        w["use_weight_chroma"] = int(rng.random() < .7)
        w["luma_denom"], w["chroma_denom"] = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        for i in range(nref):
            for l in range(2):
                flag = rng.random() < .7
                w["luma"][i, l] = (int(rng.integers(-128, 128)), int(rng.integers(-128, 128)) << (depth - 8)) if flag else (1 << w["luma_denom"], 0)
                for c in range(2):
                    flag = rng.random() < .7
                    w["chroma"][i, l, c] = (int(rng.integers(-128, 128)), int(rng.integers(-128, 128)) << (depth - 8)) if flag else (1 << w["chroma_denom"], 0)
    elif kind == 2:
        w["use_weight"] = 2
        w["use_weight_chroma"] = 2
        w["luma_denom"] = w["chroma_denom"] = 5
        v = rng.integers(-64, 129, (nref, nref))
        v[rng.random((nref, nref)) < .3] = 32
        w["implicit"][:nref, :nref, 0] = v
        w["implicit"][:nref, :nref, 1] = v
    return w


def make_inter_mb(rng, B, mb_x, mb_y, nref, mvr, depth=8, bipred=True, residual=True, cfmt=1, extra_type=0):
    """B: the header's MB_TYPE_* values (Dec.bits).  mvr: motion-vector range in quarter samples — any size, the references have no
    border.  cfmt 3 (4:4:4): the residual of every plane is luma-type (cbp & 15 and the transform size count for all three), no chroma
    DC / AC part.  Returns the macroblock's state as a dict."""
    T16, T16x8, T8x16, T8x8, P0L0, P1L0, P0L1, P1L1, DCT8 = B[:9]
    cdt = np.int16 if depth == 8 else np.int32
    m = dict(mb_x=mb_x, mb_y=mb_y, mv_cache=np.zeros((2, 40, 2), np.int16), ref_cache=np.full((2, 40), LIST_NOT_USED, np.int8),
             sub_mb_type=np.zeros(4, np.uint16), nnzc=np.zeros(15 * 8, np.uint8), mb=np.zeros(768, cdt), cbp=0,
             qmul_cb=int(rng.integers(16, 6000)), qmul_cr=int(rng.integers(16, 6000)))

    def lists():
        r = rng.random()
        return (1, 1) if (bipred and r < .4) else (1, 0) if r < .75 or not bipred else (0, 1)

    def fill(blocks, use):
        for l in (0, 1):
            if not use[l]:
                continue
            if rng.random() < .2:       # small motion: the common case, inside the picture
                mv = rng.integers(-24, 25, 2)
            else:
                mv = rng.integers(-mvr, mvr + 1, 2)
            ref = int(rng.integers(0, nref))
            for i in blocks:
                m["mv_cache"][l, SCAN8[i]] = mv
                m["ref_cache"][l, SCAN8[i]] = ref

    shape = int(rng.integers(0, 4))
    mb_type = 0
    if shape == 0:
        use = lists()
        mb_type = T16 | (P0L0 if use[0] else 0) | (P0L1 if use[1] else 0)
        fill(range(16), use)
    elif shape in (1, 2):
        mb_type = T16x8 if shape == 1 else T8x16
        halves = ([0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11, 12, 13, 14, 15]) if shape == 1 else ([0, 1, 2, 3, 8, 9, 10, 11], [4, 5, 6, 7, 12, 13, 14, 15])
        for part, blocks in enumerate(halves):
            use = lists()
            mb_type |= ((P0L0 << part) if use[0] else 0) | ((P0L1 << part) if use[1] else 0)
            fill(blocks, use)
    else:
        mb_type = T8x8 | P0L0 | P1L0 | (P0L1 | P1L1 if bipred else 0)
        for i in range(4):
            sub = int(rng.integers(0, 4))
            use = lists()
            m["sub_mb_type"][i] = [T16, T16x8, T8x16, T8x8][sub] | (P0L0 if use[0] else 0) | (P0L1 if use[1] else 0)
            n = 4 * i
            groups = {0: [[n, n + 1, n + 2, n + 3]], 1: [[n, n + 1], [n + 2, n + 3]], 2: [[n, n + 2], [n + 1, n + 3]],
                      3: [[n], [n + 1], [n + 2], [n + 3]]}[sub]
            ref0 = None
            for g in groups:              # one reference per 8x8 (sub-partitions share it), a vector per sub-partition
                fill(g, use)
                for l in (0, 1):
                    if use[l]:
                        if ref0 is None:
                            ref0 = {}
                        ref0.setdefault(l, int(m["ref_cache"][l, SCAN8[g[0]]]))
                        for b in g:
                            m["ref_cache"][l, SCAN8[b]] = ref0[l]
    if residual:
        mb, nnzc = m["mb"], m["nnzc"]
        if rng.random() < .7:
            m["cbp"] |= int(rng.integers(1, 16))
            use8 = rng.random() < .35
            if use8:
                mb_type |= DCT8
            for p in range(3 if cfmt == 3 else 1):
                o, c = 256 * p, 40 * p
                if use8:
                    for i in range(0, 16, 4):
                        if m["cbp"] & (1 << (i >> 2)):
                            n = G._block(rng, mb, o + 16 * i, 64, depth=depth)
                            for k in range(4):
                                nnzc[c + SCAN8[i + k]] = n
                else:
                    for i in range(16):
                        if m["cbp"] & (1 << (i >> 2)):
                            nnzc[c + SCAN8[i]] = G._block(rng, mb, o + 16 * i, 16, depth=depth)
        cc = 0 if cfmt in (0, 3) else int(rng.integers(0, 3))   # monochrome: no chroma residual
        m["cbp"] |= cc << 4
        sh = depth - 8
        nck = 8 if cfmt == 2 else 4           # 4:2:2: eight blocks per chroma plane
        if cc:
            for pl in (1, 2):
                if rng.random() < .7:
                    nnzc[40 * pl] = 1
                    for k in range(nck):
                        if rng.random() < .7:
                            mb[256 * pl + 16 * k] = int(rng.integers(-1500, 1501)) << sh
                if cc == 2:
                    for k in range(nck):
                        dc = mb[256 * pl + 16 * k]
                        n = G._block(rng, mb, 256 * pl + 16 * k, 16, allow_dc_only=False, depth=depth)
                        mb[256 * pl + 16 * k] = dc
                        nnzc[G.scan8_chroma(pl, k)] = n
    m["mb_type"] = int(mb_type) | extra_type       # (MB_TYPE_INTERLACED on every macroblock of a field picture)
    return m


def make_filter_picture(rng, B, mb_w, mb_h, depth=8, p_intra=.2, extra_type=0):
    """per-macroblock state for ff_h264_filter_mb() as fill_filter_caches() (h264_slice.c:2313) leaves it: types and quantizers are
    consistent across the picture (a macroblock's left / top type IS its neighbour's), the motion / reference / non-zero-count caches
    are drawn per macroblock (the function reads only the current macroblock's caches, border entries included)."""
    T16, T16x8, T8x16, T8x8, P0L0, P1L0, P0L1, P1L1, DCT8, I4, I16 = B[:11]
    qpo = 6 * (depth - 8)
    types = np.zeros((mb_h, mb_w), np.int64)
    qps = rng.integers(12 + qpo, 52 + qpo, (mb_h, mb_w))
    for y in range(mb_h):
        for x in range(mb_w):
            if rng.random() < p_intra:
                t = int(rng.choice([I4, I16, I4 | DCT8]))
            else:
                t = int(rng.choice([T16, T16x8, T8x16, T8x8])) | P0L0 | (P0L1 if rng.random() < .4 else 0)
                if t & (T16x8 | T8x16 | T8x8):
                    t |= P1L0
                if rng.random() < .3:
                    t |= DCT8
            types[y, x] = t | extra_type
    alpha_off, beta_off = int(rng.integers(-6, 7)) * 2, int(rng.integers(-6, 7)) * 2
    cabac = int(rng.integers(0, 2))
    out = []
    for y in range(mb_h):
        for x in range(mb_w):
            qp = int(qps[y, x])
            ints = np.array([types[y, x], types[y, x - 1] if x else 0, types[y - 1, x] if y else 0, qp, int(qps[y, x - 1]) if x else 0,
                             int(qps[y - 1, x]) if y else 0, int(rng.integers(0, 16)) | (int(rng.integers(0, 3)) << 4), int(rng.integers(1, 3)),
                             alpha_off, beta_off, qp, min(qp + 2, 87), cabac], np.int32)
            base = rng.integers(-8, 9, 2)
            mv = np.tile(base, (2, 40, 1)).astype(np.int16)
            noisy = rng.random((2, 40)) < .3
            mv[noisy] += rng.integers(-6, 7, (int(noisy.sum()), 2)).astype(np.int16)
            ref = rng.integers(0, 2, (2, 40)).astype(np.int8)
            if rng.random() < .5:
                ref[:] = ref[0, 0]
            if ints[7] == 1:
                ref[1] = -1
            nnz = (rng.integers(1, 4, 120) * (rng.random(120) < .35)).astype(np.uint8)
            caches = np.concatenate([ref.view(np.uint8).ravel(), nnz])
            out.append(dict(mb_x=x, mb_y=y, ints=ints, mv_cache=mv, caches=caches))
    return out
