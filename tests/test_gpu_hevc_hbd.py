"""GPU parity of hevcdsp above 8 bits (Main10 / Main12): the *_hbd batch faces vs the oracle's *_bd functions (pinned to the
reference's 10- and 12-bit template instantiations in tests/test_oracle_vs_ref_hbd.py), bit-exact.  Pixels are uint16, strides
and record offsets in bytes."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p, i16p, i32p
from test_gpu_hevc import _coeffs

pytestmark = pytest.mark.gpu
DEPTHS = [10, 12]
WIDTHS = [2, 4, 6, 8, 12, 16, 24, 32, 48, 64]


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def dev(torch, a):
    """any numpy array as a byte tensor on the device (torch's uint16 support is not relied upon)"""
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


def back(t, like):
    return t.cpu().numpy().view(like.dtype).reshape(like.shape)


def pix(rng, shape, bd, extremes=False):
    a = rng.integers(0, 1 << bd, shape).astype(np.uint16)
    if extremes:
        a[: shape[0] // 2] = rng.choice(np.array([0, (1 << bd) - 1], np.uint16), (shape[0] // 2, shape[1]))
    return a


def at(a, byte_off):
    return C.cast(a.ctypes.data + int(byte_off), u8p)


@pytest.mark.parametrize("bd", DEPTHS)
@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("lg", [2, 3, 4, 5])
def test_hevc_idct_batch_hbd(lg, kind, bd):
    from ffmpeg_amd import hevc
    torch = _torch()
    if kind == hevc.DST_4X4 and lg != 2:
        pytest.skip("transform_4x4_luma is 4x4 only")
    n = 1 << lg
    rng = np.random.default_rng(lg * 10 + kind + bd)
    W, H = 256 + 24, 128                                   # samples
    bw, bh = 256 // n, H // n
    ntu = bw * bh - 3
    coeffs = np.stack([_coeffs(rng, n, t % 4) for t in range(ntu)])
    tus = np.zeros(ntu, hevc.TU_DTYPE)
    order = rng.permutation(bw * bh)[:ntu]
    tus["coeff_offset"] = np.arange(ntu) * n * n
    tus["dst_offset"] = 2 * ((order // bw) * n * W + (order % bw) * n + 5)     # bytes; odd sample columns: dword-unaligned rows
    tus["dst_offset"][::7] = -1
    tus["col_limit"] = rng.integers(0, 2 * n + 6, ntu)
    tus["col_limit"][::11] = 1000
    pic = pix(rng, (H, W), bd, extremes=True)
    want_c, want_p = coeffs.copy(), pic.copy()
    O = ffi.oracle()
    with_dst = kind != hevc.DEQUANT
    for t in range(ntu):
        c = np.ascontiguousarray(want_c[t])
        if kind == hevc.IDCT:
            O.ffo_hevc_idct_bd(bd, lg, ptr(c, i16p), int(tus["col_limit"][t]))
        elif kind == hevc.IDCT_DC:
            O.ffo_hevc_idct_dc_bd(bd, lg, ptr(c, i16p))
        elif kind == hevc.DST_4X4:
            O.ffo_hevc_transform_4x4_luma_bd(bd, ptr(c, i16p))
        elif kind == hevc.DEQUANT:
            O.ffo_hevc_dequant_bd(bd, ptr(c, i16p), lg)
        want_c[t] = c
        if with_dst and tus["dst_offset"][t] >= 0:
            O.ffo_hevc_add_residual_bd(bd, lg, at(want_p, tus["dst_offset"][t]), ptr(c, i16p), 2 * W)
    d_c = torch.from_numpy(coeffs.copy()).cuda()
    d_p = dev(torch, pic)
    d_t = torch.from_numpy(tus.view(np.uint8).reshape(ntu, 12).copy()).cuda()
    hevc.idct_batch(kind, lg, d_c, d_p if with_dst else None, 2 * W, d_t, ntu, bit_depth=bd)
    torch.cuda.synchronize()
    assert np.array_equal(d_c.cpu().numpy(), want_c), "residuals"
    assert np.array_equal(back(d_p, pic), want_p), "picture"
    if with_dst:
        assert (want_p != pic).any()


@pytest.mark.parametrize("bd", DEPTHS)
@pytest.mark.parametrize("chroma", [0, 1])
def test_hevc_deblock_picture_hbd(chroma, bd):
    """all vertical edges of a picture, then all horizontal ones, on smooth 16-bit content that reaches strong / weak / off"""
    from ffmpeg_amd import hevc
    torch = _torch()
    rng = np.random.default_rng(40 + chroma + bd)
    sc = 1 << (bd - 8)
    W, H = 192, 96
    step = 16 if chroma else 8
    base = (rng.integers(30, 220, (H // 8, W // 8)) * sc).repeat(8, 0).repeat(8, 1)
    pic = np.clip(base + rng.integers(-2 * sc, 2 * sc + 1, (H, W)), 0, (1 << bd) - 1).astype(np.uint16)
    pic[:16] = pix(rng, (16, W), bd)
    want = pic.copy()
    O = ffi.oracle()
    d_p = dev(torch, pic)
    for vertical in (1, 0):
        recs = []
        for y in range(0, H, 8) if vertical else range(step, H, step):
            for x in (range(step, W, step) if vertical else range(0, W, 8)):
                e = np.zeros(1, hevc.EDGE_DTYPE)
                e["offset"] = 2 * (y * W + x)
                e["kind"] = (hevc.LF_V_LUMA if vertical else hevc.LF_H_LUMA) + (2 if chroma else 0)
                e["beta"] = rng.integers(0, 65)
                e["tc"] = rng.integers(0, 25, 2)
                e["no_p"] = rng.integers(0, 2, 2) * (rng.random() < .2)
                e["no_q"] = rng.integers(0, 2, 2) * (rng.random() < .2)
                recs.append(e)
        recs = np.concatenate(recs)
        for e in recs:
            tc = e["tc"].astype(np.int32)
            O.ffo_hevc_loop_filter_bd(bd, chroma, vertical, at(want, e["offset"]), 2 * W, int(e["beta"]), ptr(tc, i32p), ptr(e["no_p"].copy()),
                                      ptr(e["no_q"].copy()))
        d_e = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 16).copy()).cuda()
        hevc.loop_filter_batch(d_p, 2 * W, d_e, len(recs), bit_depth=bd)
    torch.cuda.synchronize()
    got = back(d_p, pic)
    assert (want != pic).sum() > 500
    assert np.array_equal(got, want), "%d mismatches" % (got != want).sum()


@pytest.mark.parametrize("bd", DEPTHS)
def test_hevc_sao_batch_hbd(bd):
    from ffmpeg_amd import hevc
    torch = _torch()
    rng = np.random.default_rng(60 + bd)
    sc = 1 << (bd - 8)
    W, H = 400, 200
    src = pix(rng, (H, W), bd)
    src[64:128] = np.clip(rng.integers(-2, 3, (64, W)) + 130 * sc, 0, (1 << bd) - 1)          # flat: equal neighbours
    dst0 = pix(rng, (H, W), bd)
    recs, want = [], dst0.copy()
    O = ffi.oracle()
    y = 1
    while y + 66 < H:
        x = 1
        while x + 66 < W:
            w, h = int(rng.choice([8, 16, 24, 33, 48, 64])), int(rng.choice([4, 8, 17, 32, 64]))
            r = np.zeros(1, hevc.SAO_DTYPE)
            r["dst_offset"] = r["src_offset"] = 2 * (y * W + x)
            off = (rng.integers(-31, 32, 5) * sc).astype(np.int16)
            off[0] = 0
            r["offset_val"] = off
            r["edge"], r["cls"] = int(rng.integers(0, 2)), int(rng.integers(0, 32))
            r["width"], r["height"] = w, h
            if r["edge"][0]:
                r["cls"] = int(rng.integers(0, 4))
                O.ffo_hevc_sao_edge_bd(bd, at(want, r["dst_offset"][0]), at(src, r["src_offset"][0]), 2 * W, 2 * W, ptr(off, i16p), int(r["cls"][0]), w, h)
            else:
                O.ffo_hevc_sao_band_bd(bd, at(want, r["dst_offset"][0]), at(src, r["src_offset"][0]), 2 * W, 2 * W, ptr(off, i16p), int(r["cls"][0]), w, h)
            recs.append(r)
            x += 66
        y += 66
    recs = np.concatenate(recs)
    d_d, d_s = dev(torch, dst0), dev(torch, src)
    d_r = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 24).copy()).cuda()
    hevc.sao_batch(d_d, 2 * W, d_s, 2 * W, d_r, len(recs), bit_depth=bd)
    torch.cuda.synchronize()
    got = back(d_d, dst0)
    assert (want != dst0).sum() > 1000
    assert np.array_equal(got, want), "%d mismatches" % (got != want).sum()


@pytest.mark.parametrize("bd", DEPTHS)
def test_hevc_sao_edge_restore_hbd(bd):
    from ffmpeg_amd import hevc
    from test_oracle_vs_ref import hevc_restore_case
    torch = _torch()
    rng = np.random.default_rng(66 + bd)
    sc = 1 << (bd - 8)
    W, H = 70 * 12, 70 * 6
    src, dst0 = pix(rng, (H, W), bd), pix(rng, (H, W), bd)
    want, recs = dst0.copy(), []
    O = ffi.oracle()
    rep = 0
    for by in range(6):
        for bx in range(12):
            var, eo, off0, borders, w, h, ve, he, de = hevc_restore_case(rng, rep)
            rep += 1
            o = 2 * ((by * 70 + 2) * W + bx * 70 + 2)
            O.ffo_hevc_sao_edge_restore_bd(bd, var, at(want, o), at(src, o), 2 * W, 2 * W, eo, off0 * sc, ptr(borders, i32p), w, h, ptr(ve), ptr(he),
                                           ptr(de))
            r = np.zeros(1, hevc.RESTORE_DTYPE)
            r["dst_offset"] = r["src_offset"] = o
            r["offset0"], r["width"], r["height"], r["eo"], r["variant"] = off0 * sc, w, h, eo, var
            r["borders"] = sum(int(b != 0) << i for i, b in enumerate(borders))
            r["vert_edge"] = int(ve[0]) | int(ve[1]) << 1
            r["horiz_edge"] = int(he[0]) | int(he[1]) << 1
            r["diag_edge"] = sum(int(b) << i for i, b in enumerate(de))
            recs.append(r)
    recs = np.concatenate(recs)
    d_d, d_s = dev(torch, dst0), dev(torch, src)
    d_r = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 20).copy()).cuda()
    hevc.sao_restore_batch(d_d, 2 * W, d_s, 2 * W, d_r, len(recs), bit_depth=bd)
    torch.cuda.synchronize()
    got = back(d_d, dst0)
    assert (want != dst0).any()
    assert np.array_equal(got, want), "%d mismatches" % (got != want).sum()


def _mc_blocks(rng, chroma, W, H, dtype):
    """one block per (width class, mx, my) laid out on a grid of 72 x 72-sample cells; returns records + per-block geometry"""
    nfrac = 8 if chroma else 4
    geo = []
    for w in WIDTHS:
        for mx in range(nfrac):
            for my in range(nfrac):
                geo.append((w, int(rng.choice([2, 4, 8, 16, 64])) if w > 2 else 2, mx, my))
    per_row = W // 72
    assert (len(geo) + per_row - 1) // per_row * 72 <= H
    recs = np.zeros(len(geo), dtype)
    cells = []
    for i, (w, h, mx, my) in enumerate(geo):
        cy, cx = (i // per_row) * 72 + 4, (i % per_row) * 72 + 4
        recs[i]["src_offset"] = 2 * (cy * W + cx)
        recs[i]["width"], recs[i]["height"], recs[i]["mx"], recs[i]["my"] = w, h, mx, my
        cells.append((cy, cx))
    return recs, geo, cells


@pytest.mark.parametrize("bd", DEPTHS)
@pytest.mark.parametrize("uni", [0, 1])
@pytest.mark.parametrize("chroma", [0, 1])
def test_hevc_mc_batch_hbd(chroma, uni, bd):
    from ffmpeg_amd import hevc
    torch = _torch()
    rng = np.random.default_rng(80 + chroma * 2 + uni + bd)
    W = 72 * 12
    nb = len(WIDTHS) * (64 if chroma else 16)
    H = (nb + 11) // 12 * 72
    src = pix(rng, (H, W), bd, extremes=True)
    recs, geo, cells = _mc_blocks(rng, chroma, W, H, hevc.MC_DTYPE)
    O = ffi.oracle()
    if uni:
        dst0 = pix(rng, (H, W), bd)
        want = dst0.copy()
        for i, ((w, h, mx, my), (cy, cx)) in enumerate(zip(geo, cells)):
            recs[i]["dst_offset"] = 2 * (cy * W + cx)
            O.ffo_hevc_mc_bd(bd, chroma, 1, want.ctypes.data + int(recs[i]["dst_offset"]), 2 * W, at(src, recs[i]["src_offset"]), 2 * W, h, mx, my, w)
        d_d = dev(torch, dst0)
    else:
        dst0 = np.full((len(geo), 64, 64), 77, np.int16)
        want = dst0.copy()
        for i, (w, h, mx, my) in enumerate(geo):
            recs[i]["dst_offset"] = i * 4096
            O.ffo_hevc_mc_bd(bd, chroma, 0, want[i].ctypes.data, 0, at(src, recs[i]["src_offset"]), 2 * W, h, mx, my, w)
        d_d = torch.from_numpy(dst0.copy()).cuda()
    d_s = dev(torch, src)
    d_r = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 12).copy()).cuda()
    hevc.mc_batch(chroma, uni, d_d, 2 * W, d_s, 2 * W, d_r, len(recs), bit_depth=bd)
    torch.cuda.synchronize()
    got = back(d_d, dst0) if uni else d_d.cpu().numpy()
    assert (want != dst0).sum() > 1000
    assert np.array_equal(got, want), "%d mismatches" % (got != want).sum()


@pytest.mark.parametrize("bd", DEPTHS)
@pytest.mark.parametrize("mode", [2, 3, 4])
@pytest.mark.parametrize("chroma", [0, 1])
def test_hevc_mc_weighted_batch_hbd(chroma, mode, bd):
    from ffmpeg_amd import hevc
    from test_oracle_vs_ref import hevc_weight_case
    torch = _torch()
    rng = np.random.default_rng(90 + chroma * 3 + mode + bd)
    W = 72 * 12
    nb = len(WIDTHS) * (64 if chroma else 16)
    H = (nb + 11) // 12 * 72
    src = pix(rng, (H, W), bd, extremes=True)
    recs, geo, cells = _mc_blocks(rng, chroma, W, H, hevc.MCW_DTYPE)
    src2 = rng.integers(-8192, 16384, (len(geo), 64, 64)).astype(np.int16)
    src2[::5] = 16383
    src2[1::5] = -8192
    dst0 = pix(rng, (H, W), bd)
    want = dst0.copy()
    O = ffi.oracle()
    for i, ((w, h, mx, my), (cy, cx)) in enumerate(zip(geo, cells)):
        d, wx0, wx1, ox = hevc_weight_case(rng, i)
        recs[i]["dst_offset"] = 2 * (cy * W + cx)
        recs[i]["src2_offset"] = i * 4096
        recs[i]["wx0"], recs[i]["wx1"], recs[i]["ox"], recs[i]["denom"] = wx0, wx1, ox, d
        O.ffo_hevc_mc_w_bd(bd, chroma, mode, at(want, recs[i]["dst_offset"]), 2 * W, at(src, recs[i]["src_offset"]), 2 * W, ptr(src2[i], i16p), h, d,
                           wx0, wx1, ox, mx, my, w)
    d_d, d_s = dev(torch, dst0), dev(torch, src)
    d_2 = torch.from_numpy(src2.copy()).cuda()
    d_r = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 24).copy()).cuda()
    hevc.mc_w_batch(chroma, mode, d_d, 2 * W, d_s, 2 * W, d_2, d_r, len(recs), bit_depth=bd)
    torch.cuda.synchronize()
    got = back(d_d, dst0)
    assert (want != dst0).sum() > 1000
    assert np.array_equal(got, want), "%d mismatches" % (got != want).sum()


@pytest.mark.parametrize("bd", DEPTHS)
def test_hevc_host_faces_hbd(bd):
    """ff_hevc_dsp_init_hip(c, 10 / 12): the reference's signatures with host pointers on uint16 planes — what checkasm's
    hevc_idct / hevc_add_res / hevc_deblock / hevc_sao / hevc_pel drive at these depths"""
    from ffmpeg_amd import hevc
    from test_oracle_vs_ref import hevc_weight_case
    from test_oracle_vs_ref_hbd import lf_case
    _torch()
    c = hevc.dsp_init(bd)
    O = ffi.oracle()
    rng = np.random.default_rng(500 + bd)
    sc = 1 << (bd - 8)
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        for rep in range(4):
            col_limit = int(rng.integers(0, 2 * n + 4))
            blk = _coeffs(rng, n, rep % 4)
            a, b = blk.copy(), blk.copy()
            c.idct[lg - 2](a.ctypes.data, col_limit)
            O.ffo_hevc_idct_bd(bd, lg, ptr(b, i16p), col_limit)
            assert np.array_equal(a, b), (n, col_limit)
            a, b = blk.copy(), blk.copy()
            c.idct_dc[lg - 2](a.ctypes.data)
            O.ffo_hevc_idct_dc_bd(bd, lg, ptr(b, i16p))
            assert np.array_equal(a, b)
            a, b = blk.copy(), blk.copy()
            c.dequant(a.ctypes.data, lg)
            O.ffo_hevc_dequant_bd(bd, ptr(b, i16p), lg)
            assert np.array_equal(a, b)
            res = _coeffs(rng, n, rep % 4)
            pic = pix(rng, (n + 4, 50), bd, rep % 2 == 0)
            pa, pb = pic.copy(), pic.copy()
            c.add_residual[lg - 2](pa.ctypes.data + 2 * (2 * 50 + 3), res.ctypes.data, 100)
            O.ffo_hevc_add_residual_bd(bd, lg, at(pb, 2 * (2 * 50 + 3)), ptr(res, i16p), 100)
            assert np.array_equal(pa, pb)
    blk = _coeffs(rng, 4, 0)
    a, b = blk.copy(), blk.copy()
    c.transform_4x4_luma(a.ctypes.data)
    O.ffo_hevc_transform_4x4_luma_bd(bd, ptr(b, i16p))
    assert np.array_equal(a, b)
    # deblocking
    members = [c.hevc_h_loop_filter_luma, c.hevc_v_loop_filter_luma, c.hevc_h_loop_filter_chroma, c.hevc_v_loop_filter_chroma]
    changed = 0
    for rep in range(60):
        buf, beta, tc, no_p, no_q = lf_case(rng, rep % 4 != 0, bd)
        which = rep % 4
        chroma, vertical = (which >> 1) & 1, which & 1
        a, b = buf.copy(), buf.copy()
        off = 2 * ((4 * 16 + 8) if vertical else (8 * 16 + 4))
        if chroma:
            members[which](a.ctypes.data + off, 32, tc.ctypes.data, no_p.ctypes.data, no_q.ctypes.data)
        else:
            members[which](a.ctypes.data + off, 32, beta, tc.ctypes.data, no_p.ctypes.data, no_q.ctypes.data)
        O.ffo_hevc_loop_filter_bd(bd, chroma, vertical, at(b, off), 32, beta, ptr(tc, i32p), ptr(no_p), ptr(no_q))
        assert np.array_equal(a, b), (rep, which)
        changed += int((a != buf).any())
    assert changed > 10
    # SAO (the edge filter's source stride is the reference's fixed 192 bytes)
    for rep in range(12):
        w, h = int(rng.choice([8, 16, 33, 64])), int(rng.choice([4, 17, 64]))
        idx = [0, 1, 2, 2, 3, 3, 4, 4][((w + 7) >> 3) - 1]
        off = (rng.integers(-31, 32, 5) * sc).astype(np.int16)
        off[0] = 0
        src = pix(rng, (h + 2, 96), bd)
        d0 = pix(rng, (h, 80), bd)
        a, b = d0.copy(), d0.copy()
        lc = int(rng.integers(0, 32))
        c.sao_band_filter[idx](a.ctypes.data, src.ctypes.data + 2 * 97, 160, 192, off.ctypes.data, lc, w, h)
        O.ffo_hevc_sao_band_bd(bd, ptr(b), at(src, 2 * 97), 160, 192, ptr(off, i16p), lc, w, h)
        assert np.array_equal(a, b), ("band", rep)
        eo = rep % 4
        a, b = d0.copy(), d0.copy()
        c.sao_edge_filter[idx](a.ctypes.data, src.ctypes.data + 2 * 97, 160, off.ctypes.data, eo, w, h)
        O.ffo_hevc_sao_edge_bd(bd, ptr(b), at(src, 2 * 97), 160, 192, ptr(off, i16p), eo, w, h)
        assert np.array_equal(a, b), ("edge", rep)
    # motion compensation: every output stage
    src = pix(rng, (90, 100), bd, extremes=True)
    for rep in range(16):
        chroma = rep & 1
        idx = int(rng.integers(0, 10)); w = WIDTHS[idx]; h = int(rng.choice([2, 8, 64]))
        mx, my = (int(v) for v in rng.integers(0, 8 if chroma else 4, 2))
        sp = src.ctypes.data + 2 * (10 * 100 + 12)
        slot = (int(bool(my)), int(bool(mx)))
        a16, b16 = np.zeros((64, 64), np.int16), np.zeros((64, 64), np.int16)
        (c.put_hevc_epel if chroma else c.put_hevc_qpel)[idx][slot[0]][slot[1]](a16.ctypes.data, sp, 200, h, mx, my, w)
        O.ffo_hevc_mc_bd(bd, chroma, 0, b16.ctypes.data, 0, C.cast(sp, u8p), 200, h, mx, my, w)
        assert np.array_equal(a16, b16), (chroma, w, h, mx, my)
        a, b = np.full((64, 72), 9, np.uint16), np.full((64, 72), 9, np.uint16)
        (c.put_hevc_epel_uni if chroma else c.put_hevc_qpel_uni)[idx][slot[0]][slot[1]](a.ctypes.data, 144, sp, 200, h, mx, my, w)
        O.ffo_hevc_mc_bd(bd, chroma, 1, b.ctypes.data, 144, C.cast(sp, u8p), 200, h, mx, my, w)
        assert np.array_equal(a, b), (chroma, w, h, mx, my, "uni")
        src2 = rng.integers(-8192, 16384, (64, 64)).astype(np.int16)
        d, wx0, wx1, ox = hevc_weight_case(rng, rep)
        a, b = np.full((64, 72), 9, np.uint16), np.full((64, 72), 9, np.uint16)
        (c.put_hevc_epel_bi_w if chroma else c.put_hevc_qpel_bi_w)[idx][slot[0]][slot[1]](a.ctypes.data, 144, sp, 200, src2.ctypes.data, h, d, wx0, wx1,
                                                                                         ox, mx, my, w)
        O.ffo_hevc_mc_w_bd(bd, chroma, 4, ptr(b), 144, C.cast(sp, u8p), 200, ptr(src2, i16p), h, d, wx0, wx1, ox, mx, my, w)
        assert np.array_equal(a, b), (chroma, w, h, mx, my, "bi_w")
    assert hevc.dsp_init(8) is not None
    with pytest.raises(RuntimeError):
        hevc.dsp_init(9)
