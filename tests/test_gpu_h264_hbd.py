"""GPU parity of the H.264 tables above 8 bits and of the MBAFF / 4:2:2 members: the `_hbd` batch faces (C ABI) == the oracle's *_bd
restatement (oracle/ffo_h264_hbd.c, pinned to the reference's instantiations at 8 / 9 / 10 / 12 / 14 bits by
tests/test_oracle_vs_ref_h264_hbd.py), bit for bit, many blocks per launch.  The host-pointer faces of the same kernels are
exercised by the reference's own checkasm (tests/test_gpu_checkasm.py)."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p, i16p
from test_oracle_vs_ref_h264_hbd import DEPTHS, SCAN8, pixels, coefs, at, bptr, _sigs, ALPHA, BETA

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _lib():
    from ffmpeg_amd import _lib
    return _lib.lib()


def dev(a):
    return _torch().from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()


def back(t, like):
    return t.cpu().numpy().view(like.dtype).reshape(like.shape)


@pytest.mark.parametrize("depth", DEPTHS)
def test_idct_add_batch_hbd(depth):
    torch, L, O = _torch(), _lib(), ffi.oracle()
    _sigs(ffi.ref() if ffi.have_ref() else O, O) if False else None
    O.ffo_h264_idct_bd.argtypes = [C.c_int, C.c_int, u8p, i16p, C.c_ssize_t]
    rng = np.random.default_rng(40 + depth)
    px = 2 if depth > 8 else 1
    for kind in range(6):
        n = 4 if kind in (0, 2, 4) else 8
        bw, bh = 37, 21                       # blocks per row / rows of blocks
        W = bw * n + 6
        plane = pixels(rng, (bh * n + 3, W), depth, True)
        nb = bw * bh
        co = np.stack([coefs(rng, n * n, depth, big=i % 7 == 0) for i in range(nb)])
        if kind in (2, 3):
            co[:, 1:] = 0
        off = np.array([((1 + (i // bw) * n) * W + 3 + (i % bw) * n) * px for i in range(nb)], np.int32)
        want, wc = plane.copy(), co.copy()
        for i in range(nb):
            O.ffo_h264_idct_bd(depth, kind, C.cast(want.ctypes.data + int(off[i]), u8p), C.cast(wc[i].ctypes.data, i16p), W * px)
        dp, dc, do = dev(plane), dev(co), torch.from_numpy(off).cuda()
        assert L.ffhip_h264_idct_add_batch_dev_hbd(depth, kind, dp.data_ptr(), W * px, do.data_ptr(), dc.data_ptr(), nb, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(back(dp, plane), want) and np.array_equal(back(dc, co), wc), kind


@pytest.mark.parametrize("depth", DEPTHS)
def test_idct_mb_and_dc_batch_hbd(depth):
    torch, L, O = _torch(), _lib(), ffi.oracle()
    O.ffo_h264_idct_mb_bd.argtypes = [C.c_int, C.c_int, u8p, C.POINTER(C.c_int), i16p, C.c_ssize_t, u8p]
    O.ffo_h264_idct_add8_bd.argtypes = [C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(C.c_int), i16p, C.c_ssize_t, u8p]
    O.ffo_h264_luma_dc_dequant_bd.argtypes = [C.c_int, i16p, i16p, C.c_int]
    O.ffo_h264_chroma_dc_dequant_bd.argtypes = [C.c_int, C.c_int, i16p, C.c_int]
    rng = np.random.default_rng(140 + depth)
    px = 2 if depth > 8 else 1
    mbw, mbh = 9, 5
    nmb = mbw * mbh
    W = mbw * 16 + 8
    for which in range(5):
        chroma = which >= 3
        ncoef, nnzs = (768, 120) if chroma else (256, 40)
        H = mbh * 16 + 4
        planes = [pixels(rng, (H, W), depth, True) for _ in range(2 if chroma else 1)]
        bo = np.zeros(48, np.int32)
        if which == 1:
            for i in range(0, 16, 4):
                bo[i] = ((i >> 3) * 8 * W + ((i >> 2) & 1) * 8) * px
        elif not chroma:
            for i in range(16):
                bx, by = (i & 1) + 2 * ((i >> 2) & 1), ((i >> 1) & 1) + 2 * (i >> 3)
                bo[i] = (4 * by * W + 4 * bx) * px
        else:
            for j in (1, 2):
                for k in range(8 if which == 4 else 4):
                    bo[16 * j + k + (4 if k >= 4 else 0)] = ((k >> 1) * 4 * W + (k & 1) * 4) * px
        mbo = np.array([((2 + (m // mbw) * 16) * W + 4 + (m % mbw) * 16) * px for m in range(nmb)], np.int32)
        co = np.stack([coefs(rng, ncoef, depth) for _ in range(nmb)])
        nn = rng.choice(np.array([0, 0, 1, 1, 4], np.uint8), (nmb, nnzs))
        kill = rng.random((nmb, ncoef // 16)) < .35
        for m in range(nmb):
            for b in range(ncoef // 16):
                if kill[m, b]:
                    co[m, 16 * b] = 0
        want, wc = [p.copy() for p in planes], co.copy()
        for m in range(nmb):
            if chroma:
                dd = (u8p * 2)(C.cast(want[0].ctypes.data + int(mbo[m]), u8p), C.cast(want[1].ctypes.data + int(mbo[m]), u8p))
                O.ffo_h264_idct_add8_bd(depth, int(which == 4), dd, bo.ctypes.data_as(C.POINTER(C.c_int)), C.cast(wc[m].ctypes.data, i16p), W * px, ptr(nn[m]))
            else:
                O.ffo_h264_idct_mb_bd(depth, which, C.cast(want[0].ctypes.data + int(mbo[m]), u8p), bo.ctypes.data_as(C.POINTER(C.c_int)),
                                      C.cast(wc[m].ctypes.data, i16p), W * px, ptr(nn[m]))
        dps = [dev(p) for p in planes]
        dc, dbo, dmbo, dnn = dev(co), torch.from_numpy(bo).cuda(), torch.from_numpy(mbo).cuda(), torch.from_numpy(nn.reshape(-1)).cuda()
        assert L.ffhip_h264_idct_mb_batch_dev_hbd(depth, which, dps[0].data_ptr(), dps[-1].data_ptr(), W * px, dmbo.data_ptr(), dbo.data_ptr(),
                                                  dc.data_ptr(), dnn.data_ptr(), nmb, None) == 0, L.ffhip_last_error()
        torch.cuda.synchronize()
        assert all(np.array_equal(back(d, p), w) for d, p, w in zip(dps, planes, want)) and np.array_equal(back(dc, co), wc), which
    # DC transforms
    n = 300
    q = rng.integers(1, 1 << 12, n).astype(np.int32)
    inp = np.stack([coefs(rng, 16, depth, big=True) for _ in range(n)])
    out0 = np.stack([coefs(rng, 256, depth) for _ in range(n)])
    want = out0.copy()
    for m in range(n):
        O.ffo_h264_luma_dc_dequant_bd(depth, C.cast(want[m].ctypes.data, i16p), C.cast(inp[m].copy().ctypes.data, i16p), int(q[m]))
    do, di, dq = dev(out0), dev(inp), torch.from_numpy(q).cuda()
    assert L.ffhip_h264_dc_dequant_batch_dev_hbd(depth, 0, do.data_ptr(), 256, di.data_ptr(), 16, None, dq.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(back(do, out0), want)
    for which in (1, 2):
        blk = np.stack([coefs(rng, 256, depth, big=True) for _ in range(n)])
        want = blk.copy()
        for m in range(n):
            O.ffo_h264_chroma_dc_dequant_bd(depth, which - 1, C.cast(want[m].ctypes.data, i16p), int(q[m]))
        db, dof = dev(blk), torch.arange(n, dtype=torch.int32, device="cuda") * 256
        assert L.ffhip_h264_dc_dequant_batch_dev_hbd(depth, which, db.data_ptr(), 0, None, 0, dof.data_ptr(), dq.data_ptr(), n, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(back(db, blk), want), which


@pytest.mark.parametrize("depth", DEPTHS)
def test_loop_filter_batch_hbd(depth):
    """all 16 members of the family (plain, MBAFF, 4:2:2) in one launch each, a 32 x 32 tile per edge"""
    torch, L, O = _torch(), _lib(), ffi.oracle()
    O.ffo_h264_loop_filter_bd.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.POINTER(C.c_int8)]
    rng = np.random.default_rng(240 + depth)
    px = 2 if depth > 8 else 1
    fam = [(0, 4), (1, 4), (1, 2), (4, 4), (5, 4), (5, 2), (2, 2), (3, 2), (3, 1), (3, 4), (6, 2), (7, 2), (7, 1), (7, 4)]
    tiles_x, tiles_y = 12, 10
    W = tiles_x * 32
    for kind, inner in fam:
        base = int(rng.integers(20, (1 << depth) - 20))
        plane = np.clip(base + rng.integers(-(9 << (depth - 8)), (9 << (depth - 8)) + 1, (tiles_y * 32, W)), 0, (1 << depth) - 1)
        plane = plane.astype(np.uint16 if depth > 8 else np.uint8)
        n = tiles_x * tiles_y
        ed = np.zeros(n, ffi.EDGE_DTYPE)
        ed["offset"] = [((16 + (i // tiles_x) * 32) * W + 16 + (i % tiles_x) * 32) * px for i in range(n)]
        ed["kind"], ed["pad"] = kind, inner
        ed["alpha"] = [ALPHA[i % 8] for i in range(n)]
        ed["beta"] = [BETA[(i // 3) % 8] for i in range(n)]
        ed["tc0"] = rng.integers(-1, 6, (n, 4))
        want = plane.copy()
        for i in range(n):
            tc = ed["tc0"][i].copy()
            O.ffo_h264_loop_filter_bd(depth, kind, inner, C.cast(want.ctypes.data + int(ed["offset"][i]), u8p), W * px, int(ed["alpha"][i]),
                                      int(ed["beta"][i]), tc.ctypes.data_as(C.POINTER(C.c_int8)))
        dp, de = dev(plane), torch.from_numpy(ed.view(np.uint8).reshape(-1).copy()).cuda()
        assert L.ffhip_h264_loop_filter_batch_dev_hbd(depth, dp.data_ptr(), W * px, de.data_ptr(), n, None) == 0
        torch.cuda.synchronize()
        assert (want != plane).any()
        assert np.array_equal(back(dp, plane), want), (kind, inner)


@pytest.mark.parametrize("depth", DEPTHS)
def test_mc_and_weight_batch_hbd(depth):
    torch, L, O = _torch(), _lib(), ffi.oracle()
    O.ffo_h264_qpel_bd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t]
    O.ffo_h264_chroma_mc_bd.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
    O.ffo_h264_weight_bd.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
    O.ffo_h264_biweight_bd.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(340 + depth)
    px = 2 if depth > 8 else 1
    bx, by, P = 14, 9, 8
    W, H = bx * 16 + 2 * P, by * 16 + 2 * P
    src = pixels(rng, (H, W), depth, True)
    # luma qpel: one block per 16 x 16 tile, every (avg, size, position) in the batch
    n = bx * by
    blk = np.zeros(n, ffi.QPEL_DTYPE)
    dst0 = pixels(rng, (H, W), depth)
    for i in range(n):
        o = ((P + (i // bx) * 16) * W + P + (i % bx) * 16) * px
        blk["dst_offset"][i], blk["src_offset"][i] = o, o + (int(rng.integers(-3, 4)) * W + int(rng.integers(-3, 4))) * px
        blk["mcxy"][i], blk["size_idx"][i], blk["avg"][i] = i % 16, (i // 16) % 3, (i // 48) % 2
    want = dst0.copy()
    for i in range(n):
        O.ffo_h264_qpel_bd(depth, int(blk["avg"][i]), int(blk["size_idx"][i]), int(blk["mcxy"][i]),
                           C.cast(want.ctypes.data + int(blk["dst_offset"][i]), u8p), C.cast(src.ctypes.data + int(blk["src_offset"][i]), u8p), W * px)
    dd, ds, db = dev(dst0), dev(src), torch.from_numpy(blk.view(np.uint8).reshape(-1).copy()).cuda()
    assert L.ffhip_h264_qpel_batch_dev_hbd(depth, dd.data_ptr(), ds.data_ptr(), W * px, db.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(back(dd, dst0), want)
    # chroma MC
    cdt = ffi.CHROMA_DTYPE
    cb = np.zeros(n, cdt)
    for i in range(n):
        o = ((P + (i // bx) * 16) * W + P + (i % bx) * 16) * px
        cb["dst_offset"][i], cb["src_offset"][i] = o, o + (int(rng.integers(-2, 3)) * W + int(rng.integers(-2, 3))) * px
        cb["w_idx"][i], cb["h"][i], cb["x"][i], cb["y"][i], cb["avg"][i] = i % 3, [2, 4, 8, 16][i % 4], i % 8, (i // 8) % 8, (i // 5) % 2
    want = dst0.copy()
    for i in range(n):
        O.ffo_h264_chroma_mc_bd(depth, int(cb["avg"][i]), 8 >> int(cb["w_idx"][i]), C.cast(want.ctypes.data + int(cb["dst_offset"][i]), u8p),
                                C.cast(src.ctypes.data + int(cb["src_offset"][i]), u8p), W * px, int(cb["h"][i]), int(cb["x"][i]), int(cb["y"][i]))
    dd, dcb = dev(dst0), torch.from_numpy(cb.view(np.uint8).reshape(-1).copy()).cuda()
    assert L.ffhip_h264_chroma_mc_batch_dev_hbd(depth, dd.data_ptr(), ds.data_ptr(), W * px, dcb.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(back(dd, dst0), want)
    # explicit weighting
    wdt = np.dtype([("dst_offset", "<i4"), ("src_offset", "<i4"), ("w_idx", "u1"), ("height", "u1"), ("log2_denom", "u1"), ("bi", "u1"),
                    ("weightd", "<i2"), ("weights", "<i2"), ("offset", "<i2"), ("pad", "<i2")])
    wb = np.zeros(n, wdt)
    for i in range(n):
        o = ((P + (i // bx) * 16) * W + P + (i % bx) * 16) * px
        wb["dst_offset"][i], wb["src_offset"][i] = o, o + (W + 1) * px
        wb["w_idx"][i], wb["height"][i], wb["log2_denom"][i], wb["bi"][i] = i % 4, [2, 4, 8, 16][(i // 4) % 4], i % 8, (i // 3) % 2
        wb["weightd"][i], wb["weights"][i], wb["offset"][i] = rng.integers(-128, 128), rng.integers(-128, 128), rng.integers(-128, 128)
    want = dst0.copy()
    for i in range(n):
        d = C.cast(want.ctypes.data + int(wb["dst_offset"][i]), u8p)
        w = 16 >> int(wb["w_idx"][i])
        if wb["bi"][i]:
            O.ffo_h264_biweight_bd(depth, w, d, C.cast(src.ctypes.data + int(wb["src_offset"][i]), u8p), W * px, int(wb["height"][i]),
                                   int(wb["log2_denom"][i]), int(wb["weightd"][i]), int(wb["weights"][i]), int(wb["offset"][i]))
        else:
            O.ffo_h264_weight_bd(depth, w, d, W * px, int(wb["height"][i]), int(wb["log2_denom"][i]), int(wb["weightd"][i]), int(wb["offset"][i]))
    dd, dwb = dev(dst0), torch.from_numpy(wb.view(np.uint8).reshape(-1).copy()).cuda()
    assert L.ffhip_h264_weight_batch_dev_hbd(depth, dd.data_ptr(), ds.data_ptr(), W * px, dwb.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(back(dd, dst0), want)


def test_depths_the_standard_does_not_define_are_refused():
    L = _lib()
    p = C.c_void_p(16)
    for bd in (7, 11, 13, 16):
        assert L.ffhip_h264_idct_add_batch_dev_hbd(bd, 0, p, 64, p, p, 1, None) == -22
        assert L.ffhip_h264_qpel_batch_dev_hbd(bd, p, p, 64, p, 1, None) == -22


@pytest.mark.parametrize("chroma", [False, True], ids=["luma", "chroma"])
@pytest.mark.parametrize("depth", [9, 10, 12, 14])
@pytest.mark.parametrize("mb_w,mb_h,nf,pad", [(1, 1, 1, 0), (5, 3, 1, 16), (45, 30, 2, 0), (2, 41, 1, 32), (120, 68, 1, 0), (7, 9, 9, 16)])
def test_deblock_frames_hbd(mb_w, mb_h, nf, pad, depth, chroma):
    """frame-order (decoder-order wavefront) deblocking at 9 .. 14 bits == the serial order with the depth's filters
    (h264dsp_template.c:104-330 at BIT_DEPTH > 8: alpha, beta and tc0 scaled inside), luma and one 4:2:0 chroma plane, several
    pictures per launch, hand-offs through LDS inside a workgroup and through memory between workgroups"""
    import torch
    from ffmpeg_amd import h264
    from test_gpu_h264 import EDGE_DT, LADDER
    assert torch.cuda.is_available()
    O = ffi.oracle()
    O.ffo_h264_deblock_frame_bd.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(mb_w * 100 + mb_h + depth + (7 if chroma else 0))
    n_s = 8 if chroma else 16
    h, w = mb_h * n_s, mb_w * n_s
    stride = (w + pad // 2) * 2                      # bytes; 16-byte aligned
    base = rng.integers(0, 1 << depth, (nf, h // 8 + 1, w // 8 + 1)).astype(np.int64)
    planes = np.zeros((nf, h, stride // 2), np.uint16)
    for f in range(nf):
        p = np.kron(base[f], np.ones((8, 8), np.int64))[:h, :w] + rng.integers(-6 << (depth - 8), (6 << (depth - 8)) + 1, (h, w))
        planes[f, :, :w] = np.clip(p, 0, (1 << depth) - 1)
    ne = 4 if chroma else 8
    n = mb_w * mb_h * ne
    ed = np.zeros(nf * n, EDGE_DT)
    lad = np.array(LADDER)
    sel = rng.integers(0, len(LADDER), nf * n)
    ed["alpha"], ed["beta"] = lad[sel, 0], lad[sel, 1]
    ed["kind"] = np.where(rng.random(nf * n) < .25, 6 if chroma else 4, 2 if chroma else 0)
    ed["tc0"] = rng.integers(-1, 5, (nf * n, 4))
    ed["alpha"][rng.random(nf * n) < .15] = 0        # skipped edges
    want = planes.copy()
    for f in range(nf):
        O.ffo_h264_deblock_frame_bd(depth, int(chroma), C.cast(want[f].ctypes.data, u8p), stride, mb_w, mb_h, C.c_void_p(ed[f * n:].ctypes.data))
    d = torch.from_numpy(planes.view(np.uint8).reshape(nf, h, stride)).cuda()
    h264.deblock_frames_hbd(depth, d, h * stride, nf, stride, mb_w, mb_h, torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).cuda(), chroma=chroma)
    torch.cuda.synchronize()
    from ffmpeg_amd import _lib
    assert _lib.lib().ffhip_stream_synchronize(None) == 0
    got = d.cpu().numpy().view(np.uint16).reshape(planes.shape)
    assert (want != planes).sum() > (10 if mb_w > 1 else 0)
    assert np.array_equal(got, want), "%d mismatches, first %s" % ((got != want).sum(), np.argwhere(got != want)[:4])


def test_deblock_frames_hbd_rejects():
    import torch
    from ffmpeg_amd import h264
    d = torch.zeros((16, 40), dtype=torch.uint8, device="cuda:0")
    ed = torch.zeros((8, 12), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(RuntimeError):
        h264.deblock_frames_hbd(11, d, 0, 1, 32, 1, 1, ed)          # not a depth of H264DSPContext
    with pytest.raises(RuntimeError, match="aligned"):
        h264.deblock_frames_hbd(10, d, 0, 1, 40, 1, 1, ed)          # stride not 16-byte aligned
