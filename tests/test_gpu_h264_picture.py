"""GPU parity of the caller-side batching layer: a synthetic 4:2:0 picture is "decoded" macroblock by
macroblock — every dsp call the reference's hl_decode_mb() / ff_h264_filter_mb() would make is recorded — flushed as a handful
of launches, and compared with the oracle making the same calls one by one in decoder order."""
import ctypes as C

import numpy as np
import pytest

import ffi
import h264_intra_gen as G
from ffi import ptr, u8p

pytestmark = pytest.mark.gpu

QPEL_DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("mcxy", np.uint8), ("size_idx", np.uint8), ("avg", np.uint8),
                    ("flags", np.uint8), ("src_x", np.int16), ("src_y", np.int16)])
CHROMA_DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("w_idx", np.uint8), ("h", np.uint8), ("x", np.uint8),
                      ("y", np.uint8), ("avg", np.uint8), ("flags", np.uint8), ("src_x", np.int16), ("src_y", np.int16), ("pad", np.int16)])
WEIGHT_DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("w_idx", np.uint8), ("height", np.uint8),
                      ("log2_denom", np.uint8), ("bi", np.uint8), ("weightd", np.int16), ("weights", np.int16), ("offset", np.int16),
                      ("pad", np.int16)])
EDGE_DT = np.dtype([("offset", np.int32), ("kind", np.uint8), ("alpha", np.uint8), ("beta", np.uint8), ("pad", np.uint8), ("tc0", np.int8, 4)])
LADDER = np.array([(4, 2), (15, 4), (40, 9), (80, 13), (255, 18)])
IDCT_FN = ["ffo_h264_idct_add", "ffo_h264_idct8_add", "ffo_h264_idct_dc_add", "ffo_h264_idct8_dc_add"]


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _at(a, off):
    return C.cast(a.ctypes.data + int(off), u8p)


def _make_mb(rng, mx, my, W, H, P, sy, sc, p_intra=0.0, depth=8):
    """the dsp calls of one macroblock as plain data: ("mc", plane, stage, record), ("w", plane, record), ("idct", plane, kind,
    offset, block), ("edges", plane, records); an intra macroblock is ("intra", decoder state) + its edges"""
    calls = []
    if rng.random() < p_intra:
        calls.append(("intra", G.make_intra_mb(rng, mx, my, W // 16, H // 16, depth=depth)))
        for pl in (0, 1, 2):
            ne = 8 if pl == 0 else 4
            ed = np.zeros(ne, EDGE_DT)
            sel = rng.integers(0, len(LADDER), ne)
            ed["alpha"], ed["beta"] = LADDER[sel, 0], LADDER[sel, 1]
            ed["kind"] = 4 + (2 if pl else 0)             # bS 4 / 3 around an intra macroblock: the intra filters
            ed["tc0"] = rng.integers(-1, 4, (ne, 4))
            ed["alpha"][rng.random(ne) < .2] = 0
            calls.append(("edges", pl, ed))
        return calls
    inter = rng.random() < .8
    if inter:
        parts = [(0, 0, 0)] if rng.random() < .5 else [(1, 0, 0), (1, 8, 0), (1, 0, 8), (1, 8, 8)]
        for size_idx, px, py in parts:
            size = 16 >> size_idx
            mode = str(rng.choice(["uni", "uni_w", "bi", "bi_w"]))
            x, y = mx * 16 + px, my * 16 + py
            do, cdo = y * sy + x, (y // 2) * sc + x // 2
            for li in range(1 if mode.startswith("uni") else 2):
                stage = 0 if li == 0 else (2 if mode == "bi" else 1)          # FFHIP_H264_MC_PUT / _AVG / _TMP
                refi = int(rng.integers(0, 2))
                dy, dx = (int(v) for v in rng.integers(-20, 21, 2))
                fx, fy = (int(v) for v in rng.integers(0, 4, 2))
                q = np.zeros(1, QPEL_DT)
                q[0] = (do, (refi * (H + 2 * P) + P + y + dy) * sy + P + x + dx, fx + 4 * fy, size_idx, 0, 0, 0, 0)
                calls.append(("mc", 0, stage, q))
                cw = size // 2
                for pl in (1, 2):
                    c = np.zeros(1, CHROMA_DT)
                    c[0] = (cdo, (refi * (H // 2 + P) + P // 2 + y // 2 + dy // 2) * sc + P // 2 + x // 2 + dx // 2, 0 if cw == 8 else 1, cw,
                            int(rng.integers(0, 8)), int(rng.integers(0, 8)), 0, 0, 0, 0, 0)
                    calls.append(("mc", pl, stage, c))
            if mode.endswith("_w"):
                den, of = int(rng.integers(0, 8)), int(rng.integers(-20, 21))
                wd, ws = int(rng.integers(-64, 129)), int(rng.integers(-64, 129))
                for pl in (0, 1, 2):
                    bs = size if pl == 0 else size // 2
                    off = do if pl == 0 else cdo
                    w = np.zeros(1, WEIGHT_DT)
                    w[0] = (off, off, {16: 0, 8: 1, 4: 2}[bs], bs, den, int(mode == "bi_w"), wd, ws if mode == "bi_w" else 0, of, 0)
                    calls.append(("w", pl, w))
    for pl in (0, 1, 2):
        bsz, st = (16, sy) if pl == 0 else (8, sc)
        base = my * bsz * st + mx * bsz
        use8 = pl == 0 and rng.random() < .3
        step = 8 if use8 else 4
        for by in range(0, bsz, step):
            for bx in range(0, bsz, step):
                r = rng.random()
                if r < .55:
                    continue
                dc = r > .85
                kind = (3 if dc else 1) if use8 else (2 if dc else 0)
                blk = np.zeros(64 if use8 else 16, np.int16)
                if dc:
                    blk[0] = rng.integers(-400, 401)
                else:
                    blk[:] = rng.integers(-200, 201, blk.size) * (rng.random(blk.size) < .4)
                calls.append(("idct", pl, kind, base + by * st + bx, blk))
    for pl in (0, 1, 2):
        ne = 8 if pl == 0 else 4
        ed = np.zeros(ne, EDGE_DT)
        sel = rng.integers(0, len(LADDER), ne)
        ed["alpha"], ed["beta"] = LADDER[sel, 0], LADDER[sel, 1]
        ed["kind"] = np.where(rng.random(ne) < (.25 if inter else .9), 4, 0) + (2 if pl else 0)
        ed["tc0"] = rng.integers(-1, 4, (ne, 4))
        ed["alpha"][rng.random(ne) < .2] = 0
        calls.append(("edges", pl, ed))
    return calls


def _record_one(pic, rng, O, h264, mb_w, mb_h, P, refs, strides, p_intra):
    """one picture recorded into `pic` (begin() .. the last deblock_mb) and decoded by the oracle: returns (dst0, want)"""
    W, H = mb_w * 16, mb_h * 16
    sy, sc = strides[0], strides[1]
    dst0 = [rng.integers(0, 256, (H, sy), dtype=np.uint8), rng.integers(0, 256, (H // 2, sc), dtype=np.uint8),
            rng.integers(0, 256, (H // 2, sc), dtype=np.uint8)]
    want = [a.copy() for a in dst0]
    tmp = [np.zeros_like(a) for a in dst0]                # the oracle's bi-prediction scratch
    edges = [np.zeros(mb_w * mb_h * (8 if pl == 0 else 4), EDGE_DT) for pl in range(3)]
    pic.begin()
    for my in range(mb_h):
        for mx in range(mb_w):
            for call in _make_mb(rng, mx, my, W, H, P, sy, sc, p_intra):
                if call[0] == "intra":
                    # hl_decode_mb() on the oracle's picture now (decoder order); the record runs in flush()'s wavefront
                    d = call[1]
                    mb_o = G.oracle_decode(O, d, want, strides)
                    mb_p = d["mb"].copy()
                    pic.intra_mb(G.to_record(d), d["nnzc"], mb_p, d["luma_dc"], d["pcm"])
                    assert d["type"] == G.PCM or np.array_equal(mb_p, mb_o)       # sl->mb consumed as the dsp functions do
                elif call[0] == "mc":
                    _, pl, stage, rec = call
                    tgt = tmp[pl] if stage == h264.MC_TMP else want[pl]
                    r = rec[0]
                    if pl == 0:
                        pic.mc_luma(stage, rec)
                        O.ffo_h264_qpel(int(stage == h264.MC_AVG), int(r["size_idx"]), int(r["mcxy"]), _at(tgt, r["dst_offset"]),
                                        _at(refs[0], r["src_offset"]), sy)
                    else:
                        pic.mc_chroma(pl, stage, rec)
                        O.ffo_h264_chroma_mc(int(stage == h264.MC_AVG), int(r["h"]), _at(tgt, r["dst_offset"]), _at(refs[pl], r["src_offset"]),
                                             sc, int(r["h"]), int(r["x"]), int(r["y"]))
                elif call[0] == "w":
                    _, pl, rec = call
                    pic.weight(pl, rec)
                    r = rec[0]
                    wpx = [16, 8, 4, 2][int(r["w_idx"])]
                    if r["bi"]:
                        O.ffo_h264_biweight(wpx, _at(want[pl], r["dst_offset"]), _at(tmp[pl], r["src_offset"]), strides[pl], int(r["height"]),
                                            int(r["log2_denom"]), int(r["weightd"]), int(r["weights"]), int(r["offset"]))
                    else:
                        O.ffo_h264_weight(wpx, _at(want[pl], r["dst_offset"]), strides[pl], int(r["height"]), int(r["log2_denom"]),
                                          int(r["weightd"]), int(r["offset"]))
                elif call[0] == "idct":
                    _, pl, kind, off, blk = call
                    host = blk.copy()
                    pic.idct_add(pl, kind, off, host)
                    assert host[0] == 0 and (kind >= 2 or not host.any())       # consumed as the dsp function does
                    ob = blk.copy()
                    getattr(O, IDCT_FN[kind])(_at(want[pl], off), ob.ctypes.data_as(C.POINTER(C.c_int16)), strides[pl])
                else:
                    _, pl, ed = call
                    pic.deblock_mb(pl, mx, my, ed)
                    ne = len(ed)
                    edges[pl][(my * mb_w + mx) * ne:(my * mb_w + mx + 1) * ne] = ed
    O.ffo_h264_deblock_frame(ptr(want[0]), sy, mb_w, mb_h, C.c_void_p(edges[0].ctypes.data))
    for pl in (1, 2):
        O.ffo_h264_deblock_frame_chroma(ptr(want[pl]), sc, mb_w, mb_h, C.c_void_p(edges[pl].ctypes.data))
    return dst0, want


@pytest.mark.parametrize("mb_w,mb_h,pictures,p_intra", [(6, 4, 3, 0.0), (40, 22, 1, 0.0), (6, 4, 4, .3), (40, 22, 1, .15), (11, 7, 2, 1.0),
                                                         (120, 68, 1, 1.0)])
def test_picture_pipeline(mb_w, mb_h, pictures, p_intra):
    from ffmpeg_amd import h264
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(mb_w * 31 + mb_h + int(p_intra * 100))
    P = 32                                                    # reference padding: motion vectors may leave the picture
    W, H = mb_w * 16, mb_h * 16
    sy, sc = W + 2 * P, W // 2 + P                            # one stride per plane for picture, references and scratch
    strides = [sy, sc, sc]
    # two reference pictures per plane inside ONE allocation (the DPB), reached through src_offset
    refs = [rng.integers(0, 256, (2 * (H + 2 * P), sy), dtype=np.uint8), rng.integers(0, 256, (2 * (H // 2 + P), sc), dtype=np.uint8),
            rng.integers(0, 256, (2 * (H // 2 + P), sc), dtype=np.uint8)]
    d_refs = [torch.from_numpy(r).cuda() for r in refs]
    pic = h264.Picture(mb_w, mb_h)
    for it in range(pictures):
        dst0, want = _record_one(pic, rng, O, h264, mb_w, mb_h, P, refs, strides, p_intra)
        d_dst = [torch.from_numpy(a.copy()).cuda() for a in dst0]
        pic.flush(d_dst, strides, d_refs)
        torch.cuda.synchronize()
        for pl in range(3):
            got = d_dst[pl].cpu().numpy()
            assert (want[pl] != dst0[pl]).sum() > 1000
            assert np.array_equal(got, want[pl]), "picture %d plane %d: %d mismatches" % (it, pl, (got != want[pl]).sum())
    pic.close()


@pytest.mark.parametrize("mb_w,mb_h,pictures,p_intra", [(6, 4, 5, 0.3), (20, 11, 3, 0.15), (11, 7, 35, 1.0), (12, 6, 4, 0.0)])
def test_pictures_flush_batch(mb_w, mb_h, pictures, p_intra):
    """ffhip_h264_pictures_flush (round 4): several picture objects flushed together — each picture's own prediction and residual
    launches, then ONE launch of all their intra wavefronts and the in-loop filter of all their planes side by side (pictures that
    do not sit at a constant pitch: plane and edge-record tables) — must leave every picture exactly as its own flush() does, i.e.
    as the oracle decodes it.  35 pictures: two launches; a picture without intra macroblocks and one without inter ones in a batch"""
    from ffmpeg_amd import h264
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(mb_w * 17 + mb_h + pictures)
    P = 32
    W, H = mb_w * 16, mb_h * 16
    sy, sc = W + 2 * P, W // 2 + P
    strides = [sy, sc, sc]
    refs = [rng.integers(0, 256, (2 * (H + 2 * P), sy), dtype=np.uint8), rng.integers(0, 256, (2 * (H // 2 + P), sc), dtype=np.uint8),
            rng.integers(0, 256, (2 * (H // 2 + P), sc), dtype=np.uint8)]
    d_refs = [torch.from_numpy(r).cuda() for r in refs]
    pics, wants, dsts, d0s = [], [], [], []
    for it in range(pictures):
        pic = h264.Picture(mb_w, mb_h)
        pi = p_intra if it != 1 else (0.0 if p_intra < 1.0 else 1.0)      # picture 1 of a mixed batch has no intra macroblock
        dst0, want = _record_one(pic, rng, O, h264, mb_w, mb_h, P, refs, strides, pi)
        pics.append(pic)
        wants.append(want)
        d0s.append(dst0)
        dsts.append([torch.from_numpy(a.copy()).cuda() for a in dst0])
    h264.pictures_flush(pics, dsts, strides, [d_refs] * pictures)
    torch.cuda.synchronize()
    from ffmpeg_amd import _lib
    assert all(_lib.lib().ffhip_h264_picture_status(p_._p) == 0 for p_ in pics)   # every picture of the batch complete
    for it in range(pictures):
        for pl in range(3):
            got = dsts[it][pl].cpu().numpy()
            assert np.array_equal(got, wants[it][pl]), "picture %d plane %d: %d mismatches" % (it, pl, (got != wants[it][pl]).sum())
    # the objects are reusable after a batch: the next picture of each, flushed alone
    dst0, want = _record_one(pics[0], rng, O, h264, mb_w, mb_h, P, refs, strides, p_intra)
    d_dst = [torch.from_numpy(a.copy()).cuda() for a in dst0]
    pics[0].flush(d_dst, strides, d_refs)
    torch.cuda.synchronize()
    for pl in range(3):
        assert np.array_equal(d_dst[pl].cpu().numpy(), want[pl])
    # planes the shared filter launches cannot take are refused BEFORE anything is queued: nothing of any picture is touched
    off = [torch.from_numpy(np.zeros(d0s[0][0].size + 16, np.uint8)).cuda()[4:4 + d0s[0][0].size].view(d0s[0][0].shape[0], -1)] + dsts[1][1:]
    before = [[t.clone() for t in d] for d in (dsts[0], off)]
    with pytest.raises(RuntimeError):
        h264.pictures_flush(pics[:2], [dsts[0], off], strides, [d_refs] * 2)
    torch.cuda.synchronize()
    for d, b in zip((dsts[0], off), before):
        for t, u in zip(d, b):
            assert torch.equal(t, u)
    for p_ in pics:
        p_.close()


@pytest.mark.parametrize("depth,mb_w,mb_h,pictures,p_intra", [(10, 6, 4, 2, 0.0), (10, 40, 22, 1, 0.0), (9, 7, 5, 1, 0.0), (12, 11, 7, 1, 0.0),
                                                               (14, 6, 4, 1, 0.0), (10, 6, 4, 3, .3), (10, 40, 22, 1, .15), (10, 11, 7, 2, 1.0),
                                                               (9, 6, 4, 1, 1.0), (12, 9, 5, 1, .5), (14, 6, 5, 1, 1.0), (10, 120, 68, 1, 1.0)])
def test_picture_pipeline_hbd(depth, mb_w, mb_h, pictures, p_intra):
    """A High 10-class picture (uint16_t samples, int32 coefficients, offsets and strides in bytes): flush() == the same dsp calls made
    one by one in decoder order — the inter stages and the deblocking by the oracle's depth-templated functions (oracle/ffo_h264_hbd.c,
    pinned to the reference's h264dsp / h264qpel / h264chroma instantiations at 9 / 10 / 12 / 14 bits), intra macroblocks by the
    reference's own ff_h264_hl_decode_mb() at that depth (oracle/_ref, ffref_shim_h264mb.c: hl_decode_mb_simple_16 / _complex)."""
    if p_intra and not ffi.have_ref():
        pytest.skip("oracle/_ref not built")
    from ffmpeg_amd import h264
    torch = _torch()
    O = ffi.oracle()
    i16p = C.POINTER(C.c_int16)
    O.ffo_h264_idct_bd.argtypes = [C.c_int, C.c_int, u8p, i16p, C.c_ssize_t]
    O.ffo_h264_qpel_bd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t]
    O.ffo_h264_chroma_mc_bd.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
    O.ffo_h264_weight_bd.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
    O.ffo_h264_biweight_bd.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    O.ffo_h264_deblock_frame_bd.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(mb_w * 31 + mb_h + depth)
    P = 32
    W, H = mb_w * 16, mb_h * 16
    sy, sc = W + 2 * P, W // 2 + P                            # in samples; the records count bytes
    sc += -sc % 4                                             # the intra wavefront moves four samples (8 bytes) per access
    strides = [2 * sy, 2 * sc, 2 * sc]
    top = 1 << depth
    refs = [rng.integers(0, top, (2 * (H + 2 * P), sy), dtype=np.uint16), rng.integers(0, top, (2 * (H // 2 + P), sc), dtype=np.uint16),
            rng.integers(0, top, (2 * (H // 2 + P), sc), dtype=np.uint16)]
    dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(a.shape[0], -1).copy()).cuda()
    d_refs = [dev(r) for r in refs]
    pic = h264.Picture(mb_w, mb_h, bit_depth=depth)
    for it in range(pictures):
        dst0 = [rng.integers(0, top, (H, sy), dtype=np.uint16), rng.integers(0, top, (H // 2, sc), dtype=np.uint16),
                rng.integers(0, top, (H // 2, sc), dtype=np.uint16)]
        want = [a.copy() for a in dst0]
        tmp = [np.zeros_like(a) for a in dst0]
        edges = [np.zeros(mb_w * mb_h * (8 if pl == 0 else 4), EDGE_DT) for pl in range(3)]
        pic.begin()
        for my in range(mb_h):
            for mx in range(mb_w):
                for call in _make_mb(rng, mx, my, W, H, P, sy, sc, p_intra, depth):
                    if call[0] == "intra":
                        d = call[1]
                        mb_r = G.ref_decode(ffi.ref(), d, want, strides, mb_w)
                        mb_p = d["mb"].copy()
                        pic.intra_mb(G.to_record(d), d["nnzc"], mb_p, d["luma_dc"], d["pcm"])
                        assert d["type"] == G.PCM or np.array_equal(mb_p, mb_r)       # sl->mb consumed as the dsp functions do
                    elif call[0] == "mc":
                        _, pl, stage, rec = call
                        rec = rec.copy()
                        rec["dst_offset"] *= 2
                        rec["src_offset"] *= 2
                        tgt = tmp[pl] if stage == h264.MC_TMP else want[pl]
                        r = rec[0]
                        if pl == 0:
                            pic.mc_luma(stage, rec)
                            O.ffo_h264_qpel_bd(depth, int(stage == h264.MC_AVG), int(r["size_idx"]), int(r["mcxy"]), _at(tgt, r["dst_offset"]),
                                               _at(refs[0], r["src_offset"]), strides[0])
                        else:
                            pic.mc_chroma(pl, stage, rec)
                            O.ffo_h264_chroma_mc_bd(depth, int(stage == h264.MC_AVG), int(r["h"]), _at(tgt, r["dst_offset"]), _at(refs[pl], r["src_offset"]),
                                                    strides[pl], int(r["h"]), int(r["x"]), int(r["y"]))
                    elif call[0] == "w":
                        _, pl, rec = call
                        rec = rec.copy()
                        rec["dst_offset"] *= 2
                        rec["src_offset"] *= 2
                        pic.weight(pl, rec)
                        r = rec[0]
                        wpx = [16, 8, 4, 2][int(r["w_idx"])]
                        if r["bi"]:
                            O.ffo_h264_biweight_bd(depth, wpx, _at(want[pl], r["dst_offset"]), _at(tmp[pl], r["src_offset"]), strides[pl], int(r["height"]),
                                                   int(r["log2_denom"]), int(r["weightd"]), int(r["weights"]), int(r["offset"]))
                        else:
                            O.ffo_h264_weight_bd(depth, wpx, _at(want[pl], r["dst_offset"]), strides[pl], int(r["height"]), int(r["log2_denom"]),
                                                 int(r["weightd"]), int(r["offset"]))
                    elif call[0] == "idct":
                        _, pl, kind, off, blk = call
                        blk = blk.astype(np.int32) << (depth - 8)            # dctcoef above 8 bits; residuals grow with the depth
                        host = blk.copy()
                        pic.idct_add(pl, kind, 2 * off, host)
                        assert host[0] == 0 and (kind >= 2 or not host.any())
                        ob = blk.copy()
                        O.ffo_h264_idct_bd(depth, kind, _at(want[pl], 2 * off), C.cast(ob.ctypes.data, i16p), strides[pl])
                    else:
                        _, pl, ed = call
                        pic.deblock_mb(pl, mx, my, ed)
                        ne = len(ed)
                        edges[pl][(my * mb_w + mx) * ne:(my * mb_w + mx + 1) * ne] = ed
        for pl in range(3):
            O.ffo_h264_deblock_frame_bd(depth, int(pl > 0), ptr(want[pl]), strides[pl], mb_w, mb_h, C.c_void_p(edges[pl].ctypes.data))
        d_dst = [dev(a) for a in dst0]
        pic.flush(d_dst, strides, d_refs)
        torch.cuda.synchronize()
        for pl in range(3):
            got = d_dst[pl].cpu().numpy().view(np.uint16)
            assert (want[pl] != dst0[pl]).sum() > 1000 and want[pl].max() < top
            assert np.array_equal(got, want[pl]), "picture %d plane %d: %d mismatches" % (it, pl, (got != want[pl]).sum())
    pic.close()


@pytest.mark.parametrize("wpb", [1, 2, 3])
@pytest.mark.parametrize("depth", [8, 10])
def test_intra_wavefront_workgroup_shapes(wpb, depth, monkeypatch, measure_build):
    """1 / 2 / 3 macroblock rows per workgroup (the product picks the largest of 1..4 whose line buffers fit 64 KB of LDS beside the
    tiles: 3 for a 4K picture above 8 bits, 2 at 8K): the memory hand-off alone, and the mixes of LDS and memory boundaries, give the
    same pictures — mixed and all-intra ones, and rows of a P-picture that have no intra macroblock at all."""
    monkeypatch.setenv("FFHIP_INTRA_WPB", str(wpb))
    if depth == 8:
        test_picture_pipeline(11, 7, 1, 1.0)
        test_picture_pipeline(13, 9, 1, .12)
    else:
        test_picture_pipeline_hbd(depth, 11, 7, 1, 1.0)
        test_picture_pipeline_hbd(depth, 13, 9, 1, .12)
