"""Synthetic intra macroblocks in the decoder's own terms (H264SliceContext fields), for the picture-pipeline tests:
what h264_cavlc.c / h264_cabac.c leave behind for hl_decode_mb() — prediction modes after ff_h264_check_intra*_pred_mode
(libavcodec/h264_parse.c:134-222), the neighbour-availability masks of fill_decode_caches (libavcodec/h264_mvpred.h:597-639),
the non-zero-count cache (scan8 indexing), sl->mb / sl->mb_luma_dc, cbp.  One slice per picture, no constrained_intra_pred:
a neighbour is available when it lies inside the picture."""
import ctypes as C

import numpy as np

INTRA_DT = np.dtype([("mb_x", "<i2"), ("mb_y", "<i2"), ("type", "u1"), ("pred16", "u1"), ("chroma_pred", "u1"), ("cbp", "u1"),
                     ("topleft_avail", "<u2"), ("topright_avail", "<u2"), ("pred4", "u1", (16,)), ("qmul", "<i4", (3,)),
                     ("flags", "u1"), ("pad", "u1", (3,)), ("nnz", "u1", (24,)), ("coef", "<i4"), ("blocks", "<u4"),
                     ("luma_dc", "<i2", (16,))])
assert INTRA_DT.itemsize == 108
I16, I4, I8, PCM = 0, 1, 2, 3

SCAN8 = [4 + 1 * 8, 5 + 1 * 8, 4 + 2 * 8, 5 + 2 * 8, 6 + 1 * 8, 7 + 1 * 8, 6 + 2 * 8, 7 + 2 * 8,
         4 + 3 * 8, 5 + 3 * 8, 4 + 4 * 8, 5 + 4 * 8, 6 + 3 * 8, 7 + 3 * 8, 6 + 4 * 8, 7 + 4 * 8]


def scan8_chroma(pl, k):
    return 4 + (k & 1) + (5 * pl + 1 + (k >> 1)) * 8


def bxy(i):
    return 4 * ((i & 1) + ((i >> 2) & 1) * 2), 4 * (((i >> 1) & 1) + ((i >> 3) & 1) * 2)


def _coefs(rng, n, big, depth=8):
    lim = (32767 if big else 300) << (depth - 8)           # residuals grow with the depth; dctcoef is int32 above 8 bits
    return rng.integers(-lim, lim + 1, n).astype(np.int16 if depth == 8 else np.int32)


def _block(rng, mb, at, n, allow_dc_only=True, depth=8):
    """fills mb[at:at+n] and returns the block's non-zero count as the entropy decoder would have counted it"""
    r = rng.random()
    big = rng.random() < .05
    if r < .4:
        return 0
    if r < .6 and allow_dc_only:
        mb[at] = _coefs(rng, 1, big, depth)[0] or 7
        return 1
    if r < .7:
        mb[at + int(rng.integers(1, n))] = _coefs(rng, 1, big, depth)[0] or -3
        return 1
    v = _coefs(rng, n, big, depth) * (rng.random(n) < .4)
    if np.count_nonzero(v) < 2:
        v[1], v[n - 1] = 5, -9
    mb[at:at + n] = v
    return int(np.count_nonzero(v))


def make_intra_mb(rng, mx, my, mb_w, mb_h, mtype=None, depth=8, cfmt=1):
    """one intra macroblock's decoder state as a dict; above 8 bits sl->mb / sl->mb_luma_dc hold int32 (dctcoef) and the I_PCM payload is
    the 384 depth-bit fields as they stand in the bitstream (h264_mb_template.c:100-131).  cfmt 3 (4:4:4, hl_decode_mb_444): the three
    planes carry luma-type blocks under ONE set of prediction modes — plane p's blocks at sl->mb + 256 p, its cache rows 5 * 8 * p further,
    its DCs in sl->mb_luma_dc[p] (luma_dc holds 3 x 16), 768 I_PCM fields, no chroma bits in cbp."""
    cdt = np.int16 if depth == 8 else np.int32
    sh = depth - 8
    top, left = my > 0, mx > 0
    tl, tr = top and left, top and mx + 1 < mb_w
    topleft, topright = 0xFFFF, 0xEEEA
    if not top:
        topleft, topright = 0xB3FF, 0x26EA
    if not left:
        topleft &= 0xDF5F
    if not tl:
        topleft &= 0x7FFF
    if not tr:
        topright &= 0xFBFF
    if mtype is None:
        mtype = int(rng.choice([I16, I4, I8, PCM], p=[.3, .35, .3, .05]))
    d = dict(mb_x=mx, mb_y=my, type=mtype, pred16=0, chroma_pred=0, cbp=0, topleft=topleft, topright=topright,
             pred4=np.zeros(16, np.uint8), qmul=rng.integers(16, 6000, 3).astype(np.int32), nnzc=np.zeros(15 * 8, np.uint8),
             mb=np.zeros(768, cdt), luma_dc=np.zeros(48 if cfmt == 3 else 16, cdt), pcm=None, depth=depth)
    if mtype == PCM:
        d["pcm"] = rng.integers(0, 256, (96 if cfmt == 3 else 64 if cfmt == 2 else 32 if cfmt == 0 else 48) * depth, dtype=np.uint8)   # 768 / 512 / 256 / 384 fields
        return d

    def blk_mode(i_top, i_left):           # a 16x16 / chroma mode after ff_h264_check_intra_pred_mode
        ok = [0] + ([1] if i_left else []) + ([2] if i_top else []) + ([3] if i_top and i_left and tl else [])
        m = int(rng.choice(ok))
        if m == 0:
            m = 0 if (i_top and i_left) else 4 if i_left else 5 if i_top else 6
        return m
    d["pred16"] = blk_mode(top, left)
    d["chroma_pred"] = blk_mode(top, left)
    if top and left and rng.random() < .15:
        d["chroma_pred"] = int(rng.integers(7, 11))   # the one-sided DC variants (MBAFF + constrained intra): any macroblock with neighbours
    if cfmt == 0:
        d["chroma_pred"] = 6                          # monochrome: decode_chroma is off, chroma_pred_mode = DC_128_PRED8x8 (h264_cavlc.c)
    step = 4 if mtype == I8 else 1
    for i in range(0, 16, step):
        x, y = bxy(i)
        b_top, b_left = y > 0 or top, x > 0 or left
        ok = [2]
        if b_top:
            ok += [0, 3, 7]
        if b_left:
            ok += [1, 8]
        if b_top and b_left:
            ok += [4, 5, 6]
        m = int(rng.choice(ok))
        if m == 2:
            m = 2 if (b_top and b_left) else 9 if b_left else 10 if b_top else 11
        d["pred4"][i] = m
    mb, nnzc = d["mb"], d["nnzc"]
    for p in range(3 if cfmt == 3 else 1):               # the luma-type blocks of plane p: coefficients + 256 p, cache entries + 40 p
        o, c = 256 * p, 40 * p
        if mtype == I4:
            for i in range(16):
                nnzc[c + SCAN8[i]] = _block(rng, mb, o + 16 * i, 16, depth=depth)
        elif mtype == I8:
            for i in range(0, 16, 4):
                n = _block(rng, mb, o + 16 * i, 64, depth=depth)
                for k in range(4):                       # decode_luma_residual spreads an 8x8 block's count over its four entries
                    nnzc[c + SCAN8[i + k]] = n
        else:
            if rng.random() < .7:
                nnzc[c] = 1                               # scan8[LUMA_DC_BLOCK_INDEX + p]
                d["luma_dc"][16 * p:16 * p + 16] = (rng.integers(-2000, 2001, 16) << sh) * (rng.random(16) < .6)
            for i in range(16):
                n = _block(rng, mb, o + 16 * i, 16, allow_dc_only=False, depth=depth)
                if n:
                    mb[o + 16 * i] = 0                    # the DC travels in mb_luma_dc
                if n == 0 and not nnzc[c] and rng.random() < .3:
                    mb[o + 16 * i] = (int(rng.integers(-500, 501)) << sh) or 11   # idct_add16intra's `else if (block[i * 16])`
                nnzc[c + SCAN8[i]] = int(np.count_nonzero(mb[o + 16 * i:o + 16 * i + 16])) if n else 0
    if cfmt == 3:
        d["cbp"] = int(rng.integers(0, 16))
        return d
    cc = int(rng.integers(0, 3))                         # coded_block_pattern's chroma part: 0 none, 1 DC, 2 DC + AC
    if cfmt == 0:
        cc = 0                                            # monochrome: no chroma residual
    d["cbp"] = (cc << 4) | int(rng.integers(0, 16))
    nck = 8 if cfmt == 2 else 4                          # 4:2:2: eight 4x4 blocks per chroma plane (8 x 16), the cache rows running on
    if cc:
        for pl in (1, 2):
            if rng.random() < .7:
                nnzc[40 * pl] = 1                         # scan8[CHROMA_DC_BLOCK_INDEX + pl - 1]
                for k in range(nck):
                    if rng.random() < .7:
                        mb[256 * pl + 16 * k] = int(rng.integers(-1500, 1501)) << sh
            if cc == 2:
                for k in range(nck):
                    dc = mb[256 * pl + 16 * k]
                    n = _block(rng, mb, 256 * pl + 16 * k, 16, allow_dc_only=False, depth=depth)
                    mb[256 * pl + 16 * k] = dc
                    nnzc[scan8_chroma(pl, k)] = n
    return d


def to_record(d):
    r = np.zeros(1, INTRA_DT)
    r["mb_x"], r["mb_y"], r["type"] = d["mb_x"], d["mb_y"], d["type"]
    r["pred16"], r["chroma_pred"], r["cbp"] = d["pred16"], d["chroma_pred"], d["cbp"]
    r["topleft_avail"], r["topright_avail"] = d["topleft"], d["topright"]
    r["pred4"][0] = d["pred4"]
    r["qmul"][0] = d["qmul"]
    return r


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def oracle_decode(O, d, planes, strides, mb=None, fn="ffo_h264_hl_decode_intra_mb"):
    """hl_decode_mb() for this macroblock on planes (numpy, modified in place); returns the consumed sl->mb"""
    mx, my = d["mb_x"], d["mb_y"]
    mb = d["mb"].copy() if mb is None else mb
    dc = d["luma_dc"].copy()
    at = [planes[0].ctypes.data + my * 16 * strides[0] + mx * 16, planes[1].ctypes.data + my * 8 * strides[1] + mx * 8,
          planes[2].ctypes.data + my * 8 * strides[2] + mx * 8]
    u8 = C.POINTER(C.c_uint8)
    getattr(O, fn)(C.cast(at[0], u8), C.cast(at[1], u8), C.cast(at[2], u8), C.c_ssize_t(strides[0]), C.c_ssize_t(strides[1]),
                   d["type"], d["pred16"], d["chroma_pred"], _p(d["pred4"], C.c_uint8), d["topleft"], d["topright"],
                   _p(d["nnzc"], C.c_uint8), d["cbp"], _p(mb, C.c_int16), _p(dc, C.c_int16), _p(d["qmul"], C.c_int),
                   _p(d["pcm"], C.c_uint8))
    return mb


def ref_decode(R, d, planes, strides, mb_w):
    """the reference's own ff_h264_hl_decode_mb() for this macroblock at d["depth"] (oracle/refbuild/ffref_shim_h264mb.c) on planes
    (numpy uint8 / uint16, modified in place; strides in BYTES); returns the consumed sl->mb"""
    depth = d["depth"]
    ps = 2 if depth > 8 else 1
    mx, my = d["mb_x"], d["mb_y"]
    mb, dc = d["mb"].copy(), d["luma_dc"].copy()
    at = [planes[0].ctypes.data + my * 16 * strides[0] + mx * 16 * ps, planes[1].ctypes.data + my * 8 * strides[1] + mx * 8 * ps,
          planes[2].ctypes.data + my * 8 * strides[2] + mx * 8 * ps]
    u8 = C.POINTER(C.c_uint8)
    f = R.ffref_h264_hl_decode_intra_mb_bd
    f.argtypes = [C.c_int, u8, u8, u8] + [C.c_int] * 8 + [u8, C.c_uint, C.c_uint, u8, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), u8]
    f.restype = C.c_int
    f(depth, C.cast(at[0], u8), C.cast(at[1], u8), C.cast(at[2], u8), strides[0], strides[1], mx, my, mb_w, d["type"], d["pred16"], d["chroma_pred"],
      _p(d["pred4"], C.c_uint8), d["topleft"], d["topright"], _p(d["nnzc"], C.c_uint8), d["cbp"], mb.ctypes.data, dc.ctypes.data,
      _p(d["qmul"], C.c_int32), _p(d["pcm"], C.c_uint8))
    return mb
