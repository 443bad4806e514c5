"""GPU parity: HIP h264dsp / h264qpel kernels vs the oracle, bit-exact (incl. the cleared coefficients)."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, i16p, i32p, u8p

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _coefs(rng, n, size):
    c = rng.integers(-2048, 2048, (n, size * size)).astype(np.int16)
    c[::7] = rng.integers(-32768, 32768, c[::7].shape).astype(np.int16)
    c[1::5, 1:] = 0
    c[2::11] = 0
    return c


@pytest.mark.parametrize("kind,size", [(0, 4), (1, 8), (2, 4), (3, 8)])
@pytest.mark.parametrize("w,h,stride", [(64, 32, 64), (3840, 2160, 3840), (200, 56, 211)])
def test_idct_add_batch(kind, size, w, h, stride):
    from ffmpeg_amd import h264
    torch = _torch()
    O = ffi.oracle()
    ofn = [O.ffo_h264_idct_add, O.ffo_h264_idct8_add, O.ffo_h264_idct_dc_add, O.ffo_h264_idct8_dc_add][kind]
    rng = np.random.default_rng(kind * 100 + w)
    bw, bh = w // size, h // size
    n = bw * bh
    plane = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    offs = (np.arange(bh)[:, None] * size * stride + np.arange(bw)[None, :] * size).astype(np.int32).ravel()
    coefs = _coefs(rng, n, size)
    want, wc = plane.copy(), coefs.copy()
    if n <= 40000:
        for i in range(n):
            ofn(C.cast(want.ctypes.data + int(offs[i]), u8p), ptr(wc[i], i16p), stride)
    else:  # full 4K plane: oracle on a strided sample of blocks, the rest must stay as computed by neighbours' independence
        idx = rng.choice(n, 20000, replace=False)
        for i in idx:
            ofn(C.cast(want.ctypes.data + int(offs[i]), u8p), ptr(wc[i], i16p), stride)
    d_plane = torch.from_numpy(plane).cuda(); d_offs = torch.from_numpy(offs).cuda(); d_c = torch.from_numpy(coefs).cuda()
    h264.idct_add_batch(kind, d_plane, stride, d_offs, d_c)
    torch.cuda.synchronize()
    got, gc = d_plane.cpu().numpy(), d_c.cpu().numpy()
    if n <= 40000:
        assert np.array_equal(got, want) and np.array_equal(gc, wc)
    else:
        for i in idx[:5000]:
            y, x = divmod(int(offs[i]), stride)
            assert np.array_equal(got[y:y + size, x:x + size], want[y:y + size, x:x + size])
        assert np.array_equal(gc[idx], wc[idx])
        if kind < 2:
            assert not gc.any()


@pytest.mark.parametrize("which", [0, 1, 2])
def test_idct_add_mb_batch(which):
    from ffmpeg_amd import h264
    torch = _torch()
    O = ffi.oracle()
    ofn = [O.ffo_h264_idct_add16, O.ffo_h264_idct8_add4, O.ffo_h264_idct_add16intra][which]
    rng = np.random.default_rng(10 + which)
    mbw, mbh, stride = 12, 7, 12 * 16 + 16
    nmb = mbw * mbh
    bo = np.array([(i & 1) * 4 + ((i >> 1) & 1) * 4 * stride + ((i >> 2) & 1) * 8 + (i >> 3) * 8 * stride
                   for i in range(16)], np.int32)
    plane = rng.integers(0, 256, (mbh * 16, stride), dtype=np.uint8)
    mb_off = (np.arange(mbh)[:, None] * 16 * stride + np.arange(mbw)[None, :] * 16).astype(np.int32).ravel()
    blocks = rng.integers(-512, 512, (nmb, 256)).astype(np.int16)
    nnzc = rng.integers(0, 3, (nmb, 40), dtype=np.uint8)
    for m in range(nmb):
        for i in range(16):
            r = rng.random()
            if r < .3:
                blocks[m, i * 16 + 1:(i + 1) * 16] = 0
            elif r < .45:
                blocks[m, i * 16] = 0
    want, wb = plane.copy(), blocks.copy()
    for m in range(nmb):
        ofn(C.cast(want.ctypes.data + int(mb_off[m]), u8p), ptr(bo, i32p), ptr(wb[m], i16p), stride, ptr(nnzc[m]))
    d_plane = torch.from_numpy(plane).cuda(); d_b = torch.from_numpy(blocks).cuda()
    h264.idct_add_mb_batch(which, d_plane, stride, torch.from_numpy(mb_off).cuda(), torch.from_numpy(bo).cuda(), d_b,
                           torch.from_numpy(nnzc).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(d_plane.cpu().numpy(), want) and np.array_equal(d_b.cpu().numpy(), wb)


# ---------------------------------------------------------------------------------------------
# loop filters, frame-order deblocking, luma qpel
# ---------------------------------------------------------------------------------------------
EDGE_DT = np.dtype([("offset", np.int32), ("kind", np.uint8), ("alpha", np.uint8), ("beta", np.uint8), ("pad", np.uint8),
                    ("tc0", np.int8, 4)])
# the (alpha, beta, tc0) ladder of tests/checkasm/h264dsp.c:394-402 (indexA/indexB driven), plus extremes
LADDER = [(a, b, t) for a, b, t in zip([4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71,
                                        80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255],
                                       [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13,
                                        14, 14, 15, 15, 16, 16, 17, 17, 18],
                                       [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 6,
                                        6, 7, 8, 9, 10, 11, 13])]


def _smooth_plane(rng, h, w):
    """pixels with small steps so that the filters' |p0-q0| < alpha conditions fire often"""
    base = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1)).astype(np.int32)
    p = np.kron(base, np.ones((8, 8), np.int32))[:h, :w] + rng.integers(-6, 7, (h, w))
    return np.clip(p, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("kind", range(8))
def test_loop_filter_batch(kind):
    from ffmpeg_amd import h264
    torch = _torch()
    rng = np.random.default_rng(40 + kind)
    tiles_x, tiles_y, stride = 24, 18, 24 * 32 + 32
    plane = _smooth_plane(rng, tiles_y * 16, stride)
    plane[::3] = rng.integers(0, 256, plane[::3].shape, dtype=np.uint8)
    n = tiles_x * tiles_y
    ed = np.zeros(n, EDGE_DT)
    for i in range(n):
        ty, tx = divmod(i, tiles_x)
        a, b, t = LADDER[rng.integers(len(LADDER))]
        # checkasm layout: a 32x16 tile, edge in its middle (tests/checkasm/h264dsp.c:375-440)
        if kind & 1:    # h_: vertical edge at column 16 of the tile, 16 (chroma: 8) rows
            off = (ty * 16) * stride + tx * 32 + 16
        else:           # v_: horizontal edge at row 8 of the tile, 16 (chroma: 8) columns
            off = (ty * 16 + 8) * stride + tx * 32 + 8
        ed[i] = (off, kind, a, b, 0, [rng.integers(-1, t + 2) for _ in range(4)])
    want = plane.copy()
    for i in range(n):
        ffi.oracle().ffo_h264_loop_filter(kind, C.cast(want.ctypes.data + int(ed["offset"][i]), u8p), stride,
                                          int(ed["alpha"][i]), int(ed["beta"][i]), ptr(ed["tc0"][i].copy(), ffi.i8p))
    assert (want != plane).sum() > 100
    d_plane = torch.from_numpy(plane).cuda()
    d_ed = torch.from_numpy(ed.view(np.uint8).reshape(n, 12)).cuda()
    h264.loop_filter_batch(d_plane, stride, d_ed, n)
    torch.cuda.synchronize()
    assert np.array_equal(d_plane.cpu().numpy(), want)


@pytest.mark.parametrize("mb_w,mb_h,pad", [(1, 1, 0), (5, 3, 16), (7, 4, 3), (45, 30, 0), (240, 135, 0), (2, 9, 0), (3, 5, 32), (1, 13, 0),
                                           (9, 8, 4), (120, 68, 0), (17, 6, 16)])
def test_deblock_frame(mb_w, mb_h, pad):
    """frame order (wavefront) == serial order, bit for bit; 240x135 MBs = one 4K luma plane.  pad 0 / 16 / 32: 16-byte aligned
    rows take the skewed-rows kernel (4 macroblock rows per wave; mb_h around and across multiples of 4: partial last bands,
    one-macroblock-wide pictures), pad 4 the band kernel (dword rows), pad 3 (a stride that is not a multiple of 4) the byte path
    with full release/acquire hand-offs"""
    from ffmpeg_amd import h264
    torch = _torch()
    rng = np.random.default_rng(mb_w * 100 + mb_h)
    stride = mb_w * 16 + pad
    plane = _smooth_plane(rng, mb_h * 16, stride)
    n = mb_w * mb_h * 8
    ed = np.zeros(n, EDGE_DT)
    lad = np.array(LADDER)
    sel = rng.integers(0, len(LADDER), n)
    ed["alpha"], ed["beta"] = lad[sel, 0], lad[sel, 1]
    ed["kind"] = np.where(rng.random(n) < .25, 4, 0)
    ed["tc0"] = rng.integers(-1, 5, (n, 4))
    ed["alpha"][rng.random(n) < .15] = 0            # skipped edges
    want = plane.copy()
    ffi.oracle().ffo_h264_deblock_frame(ptr(want), stride, mb_w, mb_h, C.c_void_p(ed.ctypes.data))
    d_plane = torch.from_numpy(plane).cuda()
    d_ed = torch.from_numpy(ed.view(np.uint8).reshape(n, 12)).cuda()
    h264.deblock_frame(d_plane, stride, mb_w, mb_h, d_ed)
    torch.cuda.synchronize()
    got = d_plane.cpu().numpy()
    assert (want != plane).sum() > (10 if mb_w > 1 else 0)
    assert np.array_equal(got, want), "%d mismatches" % (got != want).sum()


QPEL_DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("mcxy", np.uint8), ("size_idx", np.uint8), ("avg", np.uint8),
                    ("flags", np.uint8), ("src_x", np.int16), ("src_y", np.int16)])


@pytest.mark.parametrize("old", ["default", "0", "1", "w0"], ids=["product", "lds", "regs", "nowindow"])
@pytest.mark.parametrize("w,h,pad,mvr", [(64, 48, 0, 24), (3840, 2160, 0, 24), (208, 96, 5, 24), (208, 96, 12, 24), (1280, 720, 0, 6), (1264, 720, 4, 14),
                                         (1920, 1088, 0, 100)])
def test_qpel_batch(w, h, pad, mvr, old, monkeypatch):
    """put/avg x 16 mcXY x 3 sizes mixed in one batch; unaligned reference positions; every kernel: the workgroup-window one (1024
    blocks and up; motion within +-6 / +-14 keeps the 16 blocks of a workgroup inside one window, +-24 mixes window and per-block
    workgroups, +-100 leaves almost none; 1264 wide = 79 macroblocks per row: workgroups that straddle a row's end), the LDS-sharing
    one (stride % 4 == 0), the register-only one (pad 5)"""
    from ffmpeg_amd import h264
    torch = _torch()
    if old == "w0":
        monkeypatch.setenv("FFHIP_QPEL_W", "0")
    elif old != "default":   # a knob selects libffhip_measure.so (conftest.py); "default" is the product library
        monkeypatch.setenv("FFHIP_QPEL_OLD", old)
    rng = np.random.default_rng(w + pad)
    P = 32 if mvr <= 24 else 128                            # reference padding so that MVs may point outside the picture
    stride = w + 2 * P + pad
    ref = rng.integers(0, 256, (h + 2 * P, stride), dtype=np.uint8)
    dst = rng.integers(0, 256, (h + 2 * P, stride), dtype=np.uint8)
    blocks = []
    for my in range(h // 16):
        for mx in range(w // 16):
            size_idx = int(rng.integers(0, 3)) if w < 1000 else (0 if rng.random() < .9 else int(rng.integers(1, 3)))
            n = 16 >> size_idx
            for sy in range(0, 16, n):
                for sx in range(0, 16, n):
                    dy, dx = rng.integers(-mvr, mvr + 1, 2)
                    y, x = P + my * 16 + sy, P + mx * 16 + sx
                    blocks.append((y * stride + x, (y + dy) * stride + x + dx, rng.integers(0, 16), size_idx,
                                   rng.integers(0, 2), 0, 0, 0))
    bl = np.array(blocks, QPEL_DT)
    n = len(bl)
    want = dst.copy()
    O = ffi.oracle()
    chk = np.arange(n) if n <= 30000 else rng.choice(n, 30000, replace=False)
    for i in chk:
        b = bl[i]
        O.ffo_h264_qpel(int(b["avg"]), int(b["size_idx"]), int(b["mcxy"]), C.cast(want.ctypes.data + int(b["dst_offset"]), u8p),
                        C.cast(ref.ctypes.data + int(b["src_offset"]), u8p), stride)
    d_dst, d_ref = torch.from_numpy(dst).cuda(), torch.from_numpy(ref).cuda()
    d_bl = torch.from_numpy(bl.view(np.uint8).reshape(n, 16)).cuda()
    h264.qpel_batch(d_dst, d_ref, stride, d_bl, n)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    if n <= 30000:
        assert np.array_equal(got, want)
    else:
        for i in chk:
            b = bl[i]
            y, x = divmod(int(b["dst_offset"]), stride)
            s = 16 >> int(b["size_idx"])
            assert np.array_equal(got[y:y + s, x:x + s], want[y:y + s, x:x + s]), "block %d mc %d" % (i, b["mcxy"])


def test_idct_add8_dc_dequant_add_pixels_batches():
    """the batch faces of idct_add8 / luma + chroma dc_dequant_idct / add_pixels{4,8}_clear: many macroblocks per launch == the
    oracle one call at a time (h264idct_template.c:216-345, h264addpx_template.c)"""
    from ffmpeg_amd import h264
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(31)
    mbw, mbh, stride = 12, 7, 128
    nmb = mbw * mbh
    bo = np.array([(i & 1) * 4 + ((i >> 1) & 1) * 4 * stride for i in range(48)], np.int32)
    planes = [rng.integers(0, 256, (mbh * 8, stride), dtype=np.uint8) for _ in range(2)]
    blocks = rng.integers(-400, 400, (nmb, 768)).astype(np.int16)
    blocks[rng.random((nmb, 768)) < .5] = 0
    nnzc = rng.integers(0, 2, (nmb, 120), dtype=np.uint8)
    mboff = np.array([(m // mbw) * 8 * stride + (m % mbw) * 8 for m in range(nmb)], np.int32)
    want, wb = [a.copy() for a in planes], blocks.copy()
    for m in range(nmb):
        dp = (ffi.u8p * 2)(*[C.cast(a.ctypes.data + int(mboff[m]), ffi.u8p) for a in want])
        O.ffo_h264_idct_add8(dp, ptr(bo, i32p), ptr(wb[m], i16p), stride, ptr(nnzc[m]))
    d = [torch.from_numpy(a).cuda() for a in planes]
    db = torch.from_numpy(blocks).cuda()
    h264.idct_add8_batch(d[0], d[1], stride, torch.from_numpy(mboff).cuda(), torch.from_numpy(bo).cuda(), db, torch.from_numpy(nnzc).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(d[0].cpu().numpy(), want[0]) and np.array_equal(d[1].cpu().numpy(), want[1])
    assert np.array_equal(db.cpu().numpy(), wb) and (want[0] != planes[0]).any()
    # DC transforms
    n = 5000
    qmul = rng.choice([16, 208, 1024, 14000, 65535, -9], n).astype(np.int32)
    inp = rng.integers(-4000, 4000, (n, 16)).astype(np.int16)
    out = rng.integers(-50, 50, (n, 256)).astype(np.int16)
    wo = out.copy()
    for m in range(n):
        O.ffo_h264_luma_dc_dequant_idct(ptr(wo[m], i16p), ptr(inp[m].copy(), i16p), int(qmul[m]))
    do = torch.from_numpy(out).cuda()
    h264.luma_dc_dequant_batch(do, torch.from_numpy(inp).cuda(), torch.from_numpy(qmul).cuda())
    cblk = rng.integers(-4000, 4000, (n, 64)).astype(np.int16)
    wc = cblk.copy()
    for m in range(n):
        O.ffo_h264_chroma_dc_dequant_idct(ptr(wc[m], i16p), int(qmul[m]))
    dcb = torch.from_numpy(cblk).cuda()
    h264.chroma_dc_dequant_batch(dcb, torch.arange(n, dtype=torch.int32, device="cuda") * 64, torch.from_numpy(qmul).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(do.cpu().numpy(), wo) and np.array_equal(dcb.cpu().numpy(), wc)
    # the lossless bypass
    for kind, nn in ((h264.ADD_PIXELS4_CLEAR, 4), (h264.ADD_PIXELS8_CLEAR, 8)):
        plane = rng.integers(0, 256, (64, 256), dtype=np.uint8)
        nb = (64 // nn) * (256 // nn)
        offs = np.array([(b // (256 // nn)) * nn * 256 + (b % (256 // nn)) * nn for b in range(nb)], np.int32)
        res = rng.integers(-300, 300, (nb, nn * nn)).astype(np.int16)
        wp, wr = plane.copy(), res.copy()
        for b in range(nb):
            O.ffo_h264_add_pixels_clear(nn, C.cast(wp.ctypes.data + int(offs[b]), ffi.u8p), ptr(wr[b], i16p), 256)
        dp_, dr = torch.from_numpy(plane).cuda(), torch.from_numpy(res).cuda()
        h264.idct_add_batch(kind, dp_, 256, torch.from_numpy(offs).cuda(), dr)
        torch.cuda.synchronize()
        assert np.array_equal(dp_.cpu().numpy(), wp) and not dr.cpu().numpy().any()


@pytest.mark.parametrize("old", ["1", "2"], ids=["row-kernel", "band-kernel"])
def test_deblock_frame_row_kernel_agrees(old, monkeypatch):
    """FFHIP_DEBLOCK_OLD=1: the one-workgroup-per-row kernel (the byte path of unaligned strides), =2: the band kernel (dword rows),
    both on a 16-byte aligned picture that the skewed-rows kernel would otherwise take"""
    monkeypatch.setenv("FFHIP_DEBLOCK_OLD", old)
    test_deblock_frame(40, 37, 0)


@pytest.mark.parametrize("wpb,waves", [("1", "0"), ("2", "0"), ("4", "4"), ("4", "8"), ("3", "3"), ("1", "2")])
def test_deblock_frame_workgroup_shapes(wpb, waves, monkeypatch, measure_build):
    """the skewed-rows kernel with 1 .. 4 cooperating waves per workgroup (hand-offs through LDS inside a workgroup, through memory
    between workgroups) and with fewer workgroups than super-bands (a workgroup walks several, its strip reused) == serial order;
    luma and chroma"""
    monkeypatch.setenv("FFHIP_DEBLOCK_WPB", wpb)
    monkeypatch.setenv("FFHIP_DEBLOCK_WAVES", waves)
    test_deblock_frame(45, 30, 0)
    test_deblock_frame(9, 70, 16)
    test_deblock_frame_chroma(30, 35, 3, 0)


def test_deblock_lost_handoff_is_reported(monkeypatch, measure_build):
    """a wavefront that never receives a hand-off must time out and be REPORTED at the next synchronisation point, not leave a
    partly filtered picture behind silently (FFHIP_DEBLOCK_FAULT=1: rows do not publish their progress)"""
    from ffmpeg_amd import h264, _lib
    torch = _torch()
    L = _lib.lib()
    assert L.ffhip_stream_synchronize(None) == 0
    mb_w, mb_h = 4, 36         # nine bands of the skewed-rows kernel = three workgroups: hand-offs through LDS inside, through memory between
    plane = torch.zeros((mb_h * 16, mb_w * 16), dtype=torch.uint8, device="cuda:0")
    ed = torch.zeros((mb_w * mb_h * 8, 12), dtype=torch.uint8, device="cuda:0")
    monkeypatch.setenv("FFHIP_DEBLOCK_FAULT", "1")
    h264.deblock_frame(plane, mb_w * 16, mb_w, mb_h, ed)
    monkeypatch.delenv("FFHIP_DEBLOCK_FAULT")
    assert L.ffhip_stream_synchronize(None) == -5                       # FFHIP_EIO
    assert b"hand-off" in L.ffhip_last_error()
    assert L.ffhip_stream_synchronize(None) == 0                       # reported once
    test_deblock_frame(5, 4, 0)                                          # and the pool keeps working


def test_deblock_frames_batch():
    """several pictures in one launch == each picture alone"""
    from ffmpeg_amd import h264
    torch = _torch()
    rng = np.random.default_rng(9)
    nf, mb_w, mb_h = 24, 60, 34
    stride = mb_w * 16
    planes = np.stack([_smooth_plane(rng, mb_h * 16, stride) for _ in range(nf)])
    n = mb_w * mb_h * 8
    ed = np.zeros(nf * n, EDGE_DT)
    lad = np.array(LADDER)
    sel = rng.integers(0, len(LADDER), nf * n)
    ed["alpha"], ed["beta"] = lad[sel, 0], lad[sel, 1]
    ed["kind"] = np.where(rng.random(nf * n) < .25, 4, 0)
    ed["tc0"] = rng.integers(-1, 5, (nf * n, 4))
    want = planes.copy()
    for f in range(nf):
        ffi.oracle().ffo_h264_deblock_frame(ptr(want[f]), stride, mb_w, mb_h, C.c_void_p(ed[f * n:].ctypes.data))
    d = torch.from_numpy(planes).cuda()
    h264.deblock_frames(d, mb_h * 16 * stride, nf, stride, mb_w, mb_h, torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), want)


@pytest.mark.parametrize("mb_w,mb_h,nf,pad", [(1, 1, 1, 0), (5, 1, 1, 4), (1, 7, 2, 0), (40, 25, 3, 8), (240, 135, 2, 0), (3, 9, 9, 0), (2, 17, 1, 8),
                                              (120, 68, 2, 0), (7, 8, 17, 16), (6, 16, 1, 4)])
def test_deblock_frame_chroma(mb_w, mb_h, nf, pad):
    """one 4:2:0 chroma plane per frame in decoder order (wavefront) == the serial order, bit for bit; 240x135 MBs = the chroma
    plane of a 4K picture; several pictures per launch"""
    from ffmpeg_amd import h264
    torch = _torch()
    rng = np.random.default_rng(mb_w * 100 + mb_h + 7)
    stride = mb_w * 8 + pad
    planes = np.stack([_smooth_plane(rng, mb_h * 8, stride) for _ in range(nf)])
    n = mb_w * mb_h * 4
    ed = np.zeros(nf * n, EDGE_DT)
    lad = np.array(LADDER)
    sel = rng.integers(0, len(LADDER), nf * n)
    ed["alpha"], ed["beta"] = lad[sel, 0], lad[sel, 1]
    ed["kind"] = np.where(rng.random(nf * n) < .25, 6, 2)       # chroma intra / chroma normal
    ed["tc0"] = rng.integers(-1, 5, (nf * n, 4))
    ed["alpha"][rng.random(nf * n) < .15] = 0                     # skipped edges
    want = planes.copy()
    for f in range(nf):
        ffi.oracle().ffo_h264_deblock_frame_chroma(ptr(want[f]), stride, mb_w, mb_h, C.c_void_p(ed[f * n:].ctypes.data))
    d = torch.from_numpy(planes).cuda()
    h264.deblock_frames_chroma(d, mb_h * 8 * stride, nf, stride, mb_w, mb_h, torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).cuda())
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    assert (want != planes).sum() > (10 if mb_w > 1 else 0)
    assert np.array_equal(got, want), "%d mismatches" % (got != want).sum()


def test_deblock_frame_chroma_rejects_unaligned():
    from ffmpeg_amd import h264
    torch = _torch()
    d = torch.zeros((16, 19), dtype=torch.uint8, device="cuda:0")
    ed = torch.zeros((2 * 2 * 4, 12), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(RuntimeError, match="aligned"):
        h264.deblock_frames_chroma(d, 0, 1, 19, 2, 2, ed)


# ---------------------------------------------------------------------------------------------
# chroma 1/8-pel MC and explicit weighted prediction
# ---------------------------------------------------------------------------------------------
CHROMA_DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("w_idx", np.uint8), ("h", np.uint8), ("x", np.uint8),
                      ("y", np.uint8), ("avg", np.uint8), ("flags", np.uint8), ("src_x", np.int16), ("src_y", np.int16), ("pad", np.int16)])
WEIGHT_DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("w_idx", np.uint8), ("height", np.uint8),
                      ("log2_denom", np.uint8), ("bi", np.uint8), ("weightd", np.int16), ("weights", np.int16),
                      ("offset", np.int16), ("pad", np.int16)])


@pytest.mark.parametrize("w,h,pad", [(64, 48, 0), (1920, 1080, 0), (200, 96, 3)])
def test_chroma_mc_batch(w, h, pad):
    """every 8x8 chroma block of a plane: mixed widths 8/4/2, heights, all (x,y) fractions, put/avg"""
    from ffmpeg_amd import h264
    torch = _torch()
    assert CHROMA_DT.itemsize == 20
    rng = np.random.default_rng(w + pad)
    P = 16
    stride = w + 2 * P + pad
    ref = rng.integers(0, 256, (h + 2 * P, stride), dtype=np.uint8)
    dst = rng.integers(0, 256, (h + 2 * P, stride), dtype=np.uint8)
    blocks = []
    for by in range(h // 8):
        for bx in range(w // 8):
            w_idx = int(rng.integers(0, 3))
            bw = 8 >> w_idx
            bh = int(rng.choice([2, 4, 8]))
            for sy in range(0, 8, bh):
                for sx in range(0, 8, bw):
                    dy, dx = rng.integers(-8, 9, 2)
                    y, x = P + by * 8 + sy, P + bx * 8 + sx
                    blocks.append((y * stride + x, (y + dy) * stride + x + dx, w_idx, bh, rng.integers(0, 8), rng.integers(0, 8),
                                   rng.integers(0, 2), 0))
    bl = np.zeros(len(blocks), CHROMA_DT)
    for i, b in enumerate(blocks):
        bl[i] = b[:7] + (0, 0, 0, 0)
    n = len(bl)
    chk = np.arange(n) if n <= 30000 else rng.choice(n, 30000, replace=False)
    want = dst.copy()
    for i in chk:
        b = bl[i]
        ffi.oracle().ffo_h264_chroma_mc(int(b["avg"]), 8 >> int(b["w_idx"]), C.cast(want.ctypes.data + int(b["dst_offset"]), u8p),
                                        C.cast(ref.ctypes.data + int(b["src_offset"]), u8p), stride, int(b["h"]), int(b["x"]), int(b["y"]))
    d_dst, d_ref = torch.from_numpy(dst).cuda(), torch.from_numpy(ref).cuda()
    h264.chroma_mc_batch(d_dst, d_ref, stride, torch.from_numpy(bl.view(np.uint8).reshape(n, 20)).cuda(), n)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    if n <= 30000:
        assert np.array_equal(got, want)
    else:
        for i in chk:
            b = bl[i]
            y, x = divmod(int(b["dst_offset"]), stride)
            assert np.array_equal(got[y:y + int(b["h"]), x:x + (8 >> int(b["w_idx"]))], want[y:y + int(b["h"]), x:x + (8 >> int(b["w_idx"]))])


def test_weight_batch():
    from ffmpeg_amd import h264
    torch = _torch()
    assert WEIGHT_DT.itemsize == 20
    rng = np.random.default_rng(12)
    w, h, stride = 256, 128, 272
    dst = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    src = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    bl = np.zeros((h // 16) * (w // 16), WEIGHT_DT)
    i = 0
    for by in range(h // 16):
        for bx in range(w // 16):
            bl[i] = (by * 16 * stride + bx * 16, by * 16 * stride + bx * 16, rng.integers(0, 4), rng.choice([2, 4, 8, 16]),
                     rng.integers(0, 8), rng.integers(0, 2), rng.integers(-128, 128), rng.integers(-128, 128),
                     rng.integers(-128, 128), 0)
            i += 1
    want = dst.copy()
    for b in bl:
        pd = C.cast(want.ctypes.data + int(b["dst_offset"]), u8p)
        if b["bi"]:
            ffi.oracle().ffo_h264_biweight(16 >> int(b["w_idx"]), pd, C.cast(src.ctypes.data + int(b["src_offset"]), u8p), stride,
                                           int(b["height"]), int(b["log2_denom"]), int(b["weightd"]), int(b["weights"]), int(b["offset"]))
        else:
            ffi.oracle().ffo_h264_weight(16 >> int(b["w_idx"]), pd, stride, int(b["height"]), int(b["log2_denom"]), int(b["weightd"]),
                                         int(b["offset"]))
    d_dst, d_src = torch.from_numpy(dst).cuda(), torch.from_numpy(src).cuda()
    h264.weight_batch(d_dst, d_src, stride, torch.from_numpy(bl.view(np.uint8).reshape(-1, 20)).cuda(), bl.size)
    torch.cuda.synchronize()
    assert np.array_equal(d_dst.cpu().numpy(), want)


# ---------------------------------------------------------------------------------------------
# FFHIP_MC_EMU: blocks whose footprint leaves an UNPADDED reference picture (h264_mb.c:229-247, 297-317)
# ---------------------------------------------------------------------------------------------
def _emu(depth=8):
    R = ffi.ref()
    R.ffref_emulated_edge_mc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t] + [C.c_int] * 6
    R.ffref_emulated_edge_mc.restype = None
    return lambda buf, src_addr, stride, bw, bh, x, y, w, h: R.ffref_emulated_edge_mc(depth, buf.ctypes.data, src_addr, stride, stride, bw, bh, x, y, w, h)


@pytest.mark.parametrize("kern", ["default", "0", "1"], ids=["product", "lds", "regs"])
@pytest.mark.parametrize("w,h,pad,n", [(64, 48, 0, 600), (64, 48, 5, 600), (208, 96, 0, 6000), (1920, 1088, 0, 40000)])
def test_qpel_edge_emulation(w, h, pad, n, kern, monkeypatch):
    """n blocks of every size / position / op at random places around and far outside a w x h reference picture that has NO border:
    records flagged FFHIP_MC_EMU == the reference's emulated_edge_mc() (videodsp_template.c:24, compiled in place) into a 21 x 21
    buffer, then the qpel function on that buffer — what mc_dir_part() does; unflagged interior blocks ride in the same batch"""
    if not ffi.have_ref():
        pytest.skip("oracle/_ref not built")
    from ffmpeg_amd import h264
    torch = _torch()
    if kern != "default":
        monkeypatch.setenv("FFHIP_QPEL_OLD", kern)
    rng = np.random.default_rng(w + pad + n)
    stride = max(w + 16, 1024) + pad                          # one stride for both operands; 64 destination blocks per row
    two = 2                                                   # two reference pictures in the allocation, back to back: no border
    ref = rng.integers(0, 256, (two * h, stride), dtype=np.uint8)
    dst = rng.integers(0, 256, (16 * ((n + 63) // 64), stride), dtype=np.uint8)
    emu = _emu()
    bl = np.zeros(n, QPEL_DT)
    want = dst.copy()
    O = ffi.oracle()
    buf = np.zeros((21, stride), np.uint8)
    for i in range(n):
        size_idx = int(rng.integers(0, 3))
        far = rng.random() < .3
        x = int(rng.integers(-4000, 4000)) if far else int(rng.integers(-40, w + 24))
        y = int(rng.integers(-4000, 4000)) if far else int(rng.integers(-40, h + 24))
        pic = int(rng.integers(0, two))
        mc, avg = int(rng.integers(0, 16)), int(rng.integers(0, 2))
        do = (i // 64) * 16 * stride + (i % 64) * 16
        inside = x >= 2 and y >= 2 and x + 16 + 3 <= w and y + 16 + 3 <= h
        org = pic * h * stride
        if inside and rng.random() < .5:
            bl[i] = (do, org + y * stride + x, mc, size_idx, avg, 0, 0, 0)
            src_at = ref.ctypes.data + org + y * stride + x
        else:
            bl[i] = (do, org, mc, size_idx, avg, h264.MC_EMU, x, y)
            emu(buf, ref.ctypes.data + org + (y - 2) * stride + (x - 2), stride, 21, 21, x - 2, y - 2, w, h)
            src_at = buf.ctypes.data + 2 * stride + 2
        O.ffo_h264_qpel(avg, size_idx, mc, C.cast(want.ctypes.data + do, u8p), C.cast(src_at, u8p), stride)
    d_dst, d_ref = torch.from_numpy(dst).cuda(), torch.from_numpy(ref).cuda()
    d_bl = torch.from_numpy(bl.view(np.uint8).reshape(n, 16)).cuda()
    h264.qpel_batch(d_dst, d_ref, stride, d_bl, n, pic=(w, h))
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    bad = got != want
    assert not bad.any(), "%d mismatches, first block %d" % (bad.sum(), (np.argwhere(bad)[0][0] // 16) * 64 + np.argwhere(bad)[0][1] // 16)


@pytest.mark.parametrize("w,h,n", [(32, 24, 500), (104, 48, 4000), (960, 544, 20000)])
def test_chroma_mc_edge_emulation(w, h, n):
    """chroma blocks around and far outside an unpadded w x h chroma plane: FFHIP_MC_EMU == emulated_edge_mc(buf, src, …, 9, 9, x, y, w, h)
    then the chroma function on the buffer (h264_mb.c:297-317)"""
    if not ffi.have_ref():
        pytest.skip("oracle/_ref not built")
    from ffmpeg_amd import h264
    torch = _torch()
    rng = np.random.default_rng(w + n)
    stride = max(w + 8, 64 * 8)
    ref = rng.integers(0, 256, (2 * h, stride), dtype=np.uint8)
    dst = rng.integers(0, 256, (16 * ((n + 63) // 64), stride), dtype=np.uint8)
    emu = _emu()
    bl = np.zeros(n, CHROMA_DT)
    want = dst.copy()
    O = ffi.oracle()
    buf = np.zeros((17, stride), np.uint8)
    for i in range(n):
        w_idx = int(rng.integers(0, 3))
        bh = int(rng.choice([2, 4, 8, 16] if w_idx == 0 else [2, 4, 8]))
        far = rng.random() < .3
        x = int(rng.integers(-3000, 3000)) if far else int(rng.integers(-20, w + 12))
        y = int(rng.integers(-3000, 3000)) if far else int(rng.integers(-20, h + 12))
        pic = int(rng.integers(0, 2))
        fx, fy, avg = int(rng.integers(0, 8)), int(rng.integers(0, 8)), int(rng.integers(0, 2))
        do = (i // 64) * 16 * stride + (i % 64) * 8
        org = pic * h * stride
        bl[i] = (do, org, w_idx, bh, fx, fy, avg, h264.MC_EMU, x, y, 0)
        emu(buf, ref.ctypes.data + org + y * stride + x, stride, 9, 17, x, y, w, h)
        O.ffo_h264_chroma_mc(avg, 8 >> w_idx, C.cast(want.ctypes.data + do, u8p), C.cast(buf.ctypes.data, u8p), stride, bh, fx, fy)
    d_dst, d_ref = torch.from_numpy(dst).cuda(), torch.from_numpy(ref).cuda()
    h264.chroma_mc_batch(d_dst, d_ref, stride, torch.from_numpy(bl.view(np.uint8).reshape(n, 20)).cuda(), n, pic=(w, h))
    torch.cuda.synchronize()
    assert np.array_equal(d_dst.cpu().numpy(), want)
