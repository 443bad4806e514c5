"""GPU parity: HIP hevcdsp inverse transforms vs the oracle, bit-exact (residuals left in coeffs AND the picture)."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _coeffs(rng, n, kind):
    if kind == 0:
        c = rng.integers(-32768, 32768, (n, n))
    elif kind == 1:
        c = rng.integers(-512, 512, (n, n))
    elif kind == 2:
        c = np.zeros((n, n), np.int64)
        k = int(rng.integers(1, n + 1))
        c[:k, :k] = rng.integers(-2048, 2048, (k, k))
    else:
        c = rng.choice(np.array([-32768, 32767, 0, 1, -1]), (n, n))
    return c.astype(np.int16)


@pytest.mark.parametrize("with_dst", [True, False])
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("lg", [2, 3, 4, 5])
def test_hevc_idct_batch(lg, kind, with_dst):
    """a frame's worth of TUs of one size: every col_limit (incl. odd / out of range), coefficient blocks that are dense,
    sparse and saturating, units that skip the picture, a ragged last wave"""
    _run_idct(lg, kind, with_dst)


@pytest.mark.parametrize("with_dst", [True, False])
@pytest.mark.parametrize("lg,env", [(4, "FFHIP_HEVC_IDCT16_MFMA"), (5, "FFHIP_HEVC_IDCT32_VALU")])
def test_hevc_idct_other_kernel(lg, env, with_dst, monkeypatch):
    """the kernels that are not the default for their size stay correct: 16x16 as two units per MFMA, 32x32 on the dot2 kernel"""
    monkeypatch.setenv(env, "1")
    _run_idct(lg, 0, with_dst)


def _run_idct(lg, kind, with_dst):
    from ffmpeg_amd import hevc
    torch = _torch()
    if kind == hevc.DST_4X4 and lg != 2:
        pytest.skip("transform_4x4_luma is 4x4 only")
    if kind == hevc.ADD_ONLY and not with_dst:
        pytest.skip("nothing to do")
    n = 1 << lg
    rng = np.random.default_rng(lg * 10 + kind)
    W, H = 256 + 24, 128
    bw, bh = 256 // n, H // n
    ntu = bw * bh - 3
    coeffs = np.stack([_coeffs(rng, n, t % 4) for t in range(ntu)])
    tus = np.zeros(ntu, hevc.TU_DTYPE)
    order = rng.permutation(bw * bh)[:ntu]
    tus["coeff_offset"] = np.arange(ntu) * n * n
    tus["dst_offset"] = (order // bw) * n * W + (order % bw) * n + 5          # unaligned picture columns
    tus["dst_offset"][::7] = -1
    tus["col_limit"] = rng.integers(0, 2 * n + 6, ntu)
    tus["col_limit"][::11] = 1000
    pic = rng.integers(0, 256, (H, W), dtype=np.uint8)
    want_c, want_p = coeffs.copy(), pic.copy()
    O = ffi.oracle()
    for t in range(ntu):
        c = np.ascontiguousarray(want_c[t])
        if kind == hevc.IDCT:
            O.ffo_hevc_idct(lg, ptr(c, ffi.i16p), int(tus["col_limit"][t]))
        elif kind == hevc.IDCT_DC:
            O.ffo_hevc_idct_dc(lg, ptr(c, ffi.i16p))
        elif kind == hevc.DST_4X4:
            O.ffo_hevc_transform_4x4_luma(ptr(c, ffi.i16p))
        want_c[t] = c
        if with_dst and tus["dst_offset"][t] >= 0:
            O.ffo_hevc_add_residual(lg, C.cast(want_p.ctypes.data + int(tus["dst_offset"][t]), u8p), ptr(c, ffi.i16p), W)
    d_c = torch.from_numpy(coeffs.copy()).cuda()
    d_p = torch.from_numpy(pic.copy()).cuda()
    d_t = torch.from_numpy(tus.view(np.uint8).reshape(ntu, 12).copy()).cuda()
    hevc.idct_batch(kind, lg, d_c, d_p if with_dst else None, W, d_t, ntu)
    torch.cuda.synchronize()
    assert np.array_equal(d_c.cpu().numpy(), want_c), "residuals"
    assert np.array_equal(d_p.cpu().numpy(), want_p if with_dst else pic), "picture"
    if kind != hevc.ADD_ONLY:
        assert (want_c != coeffs).any()


def test_hevc_host_faces():
    """ff_hevc_dsp_init_hip: the reference's signatures with host pointers (what checkasm's hevc_idct / hevc_add_res drive)"""
    from ffmpeg_amd import hevc
    _torch()
    c = hevc.dsp_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(5)
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        for rep in range(6):
            col_limit = int(rng.integers(0, 2 * n + 4))
            blk = _coeffs(rng, n, rep % 4)
            a, b = blk.copy(), blk.copy()
            c.idct[lg - 2](a.ctypes.data, col_limit)
            O.ffo_hevc_idct(lg, ptr(b, ffi.i16p), col_limit)
            assert np.array_equal(a, b), (n, col_limit)
            a, b = blk.copy(), blk.copy()
            c.idct_dc[lg - 2](a.ctypes.data)
            O.ffo_hevc_idct_dc(lg, ptr(b, ffi.i16p))
            assert np.array_equal(a, b)
            res = _coeffs(rng, n, rep % 4)
            pic = rng.integers(0, 256, (n + 4, 50), dtype=np.uint8)
            pa, pb = pic.copy(), pic.copy()
            c.add_residual[lg - 2](pa.ctypes.data + 2 * 50 + 3, res.ctypes.data, 50)
            O.ffo_hevc_add_residual(lg, C.cast(pb.ctypes.data + 2 * 50 + 3, u8p), ptr(res, ffi.i16p), 50)
            assert np.array_equal(pa, pb)
    blk = _coeffs(rng, 4, 0)
    a, b = blk.copy(), blk.copy()
    c.transform_4x4_luma(a.ctypes.data)
    O.ffo_hevc_transform_4x4_luma(ptr(b, ffi.i16p))
    assert np.array_equal(a, b)


def _lf_picture(rng, w, h):
    """smooth blocks with steps on the 8x8 grid (so that all three filter classes fire) plus noise patches"""
    pic = np.zeros((h, w), np.int64)
    for by in range(0, h, 8):
        for bx in range(0, w, 8):
            pic[by:by + 8, bx:bx + 8] = rng.integers(30, 220)
    pic = pic + rng.integers(-2, 3, (h, w))
    noisy = rng.random((h // 8, w // 8)) < .15
    pic = np.where(np.kron(noisy, np.ones((8, 8), bool)), rng.integers(0, 256, (h, w)), pic)
    small = rng.random((h // 8, w // 8)) < .5                       # neighbours a few levels apart: weak / strong filters
    pic = np.where(np.kron(small, np.ones((8, 8), bool)), 120 + (pic % 9), pic)
    return np.clip(pic, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("chroma", [0, 1])
def test_hevc_deblock_picture(chroma):
    """a picture's worth of edge segments: all vertical edges in one call, then all horizontal ones (the decoder's order),
    against the oracle applied segment by segment in the same two phases"""
    from ffmpeg_amd import hevc
    torch = _torch()
    rng = np.random.default_rng(40 + chroma)
    W, H, pad = 256, 128, 24
    stride = W + pad
    pic = np.zeros((H, stride), np.uint8)
    pic[:, :W] = _lf_picture(rng, W, H)
    want = pic.copy()
    O = ffi.oracle()
    d_pic = torch.from_numpy(pic.copy()).cuda()
    changed = 0
    for vertical in (1, 0):
        segs = []
        if vertical:
            for x in range(8, W, 8):
                for y in range(0, H, 8):
                    segs.append(y * stride + x)
        else:
            for y in range(8, H, 8):
                for x in range(0, W, 8):
                    segs.append(y * stride + x)
        n = len(segs)
        ed = np.zeros(n, hevc.EDGE_DTYPE)
        ed["offset"] = segs
        ed["kind"] = (2 if chroma else 0) + vertical
        ed["beta"] = rng.integers(0, 65, n)
        ed["tc"] = rng.integers(0, 25, (n, 2))
        ed["no_p"] = rng.random((n, 2)) < .1
        ed["no_q"] = rng.random((n, 2)) < .1
        before = want.copy()
        for i in range(n):
            O.ffo_hevc_loop_filter(chroma, vertical, C.cast(want.ctypes.data + int(ed["offset"][i]), u8p), stride, int(ed["beta"][i]),
                                   ptr(ed["tc"][i].astype(np.int32), ffi.i32p), ptr(np.ascontiguousarray(ed["no_p"][i])),
                                   ptr(np.ascontiguousarray(ed["no_q"][i])))
        changed += int((before != want).sum())
        hevc.loop_filter_batch(d_pic, stride, torch.from_numpy(ed.view(np.uint8).reshape(n, 16).copy()).cuda(), n)
    torch.cuda.synchronize()
    assert changed > 2000
    assert np.array_equal(d_pic.cpu().numpy(), want)


def test_hevc_deblock_host_faces():
    from ffmpeg_amd import hevc
    _torch()
    c = hevc.dsp_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(77)
    names = ["hevc_h_loop_filter_luma", "hevc_v_loop_filter_luma", "hevc_h_loop_filter_chroma", "hevc_v_loop_filter_chroma"]
    for rep in range(40):
        which = rep % 4
        chroma, vertical = which >> 1, which & 1
        buf = _lf_picture(rng, 16, 16)
        beta = int(rng.integers(0, 65))
        tc = rng.integers(0, 25, 2).astype(np.int32)
        no_p = (rng.random(2) < .2).astype(np.uint8)
        no_q = (rng.random(2) < .2).astype(np.uint8)
        a, b = buf.copy(), buf.copy()
        off = 4 * 16 + 8 if vertical else 8 * 16 + 4
        fn = getattr(c, names[which] + ("_c" if rep % 8 >= 4 else ""))
        if chroma:
            fn(a.ctypes.data + off, 16, tc.ctypes.data, no_p.ctypes.data, no_q.ctypes.data)
        else:
            fn(a.ctypes.data + off, 16, beta, tc.ctypes.data, no_p.ctypes.data, no_q.ctypes.data)
        O.ffo_hevc_loop_filter(chroma, vertical, C.cast(b.ctypes.data + off, u8p), 16, beta, ptr(tc, ffi.i32p), ptr(no_p), ptr(no_q))
        assert np.array_equal(a, b), (rep, which)


@pytest.mark.parametrize("aligned", [0, 1])
def test_hevc_sao_batch(aligned):
    """a picture's worth of CTB blocks, band and edge classes mixed, ragged widths/heights, source = a padded copy; destination
    rows on and off the dword grid (packed / bytewise stores), offsets at the int8 limits and beyond them (bytewise arithmetic)"""
    from ffmpeg_amd import hevc
    torch = _torch()
    rng = np.random.default_rng(50 + aligned)
    W, H, M = 320, 448, 8                                   # picture and margin of the source copy
    ss, sd = W + 2 * M + 5, W + (12 if aligned else 11)
    src = rng.integers(0, 256, (H + 2 * M, ss), dtype=np.uint8)
    src[M:M + H, M:M + W] = np.clip(128 + np.cumsum(rng.integers(-3, 4, (H, W)), axis=1) % 40 + rng.integers(-2, 3, (H, W)), 0, 255)
    dst = rng.integers(0, 256, (H, sd), dtype=np.uint8)
    blocks = []
    for by in range(0, H, 64):
        for bx in range(0, W, 64):
            w, h = min(64, W - bx), min(64, H - by)
            if rng.random() < .3:
                w, h = max(1, w - int(rng.integers(0, 9))), max(1, h - int(rng.integers(0, 9)))
            edge = int(rng.integers(0, 2))
            off = rng.integers(-7, 8, 5)
            if edge:
                off[0] = 0
            kind = len(blocks) % 5
            if kind == 3:
                off = rng.choice(np.array([-128, 127, -100, 90, 0]), 5)     # the packed path's limits
            elif kind == 4:
                off = rng.integers(-300, 301, 5)                            # beyond int8: the signature allows it
            blocks.append((by * sd + bx, (by + M) * ss + bx + M, off, edge, int(rng.integers(0, 4 if edge else 32)), w, h))
    n = len(blocks)
    rec = np.zeros(n, hevc.SAO_DTYPE)
    for i, (do, so, off, edge, cls, w, h) in enumerate(blocks):
        rec[i] = (do, so, off, edge, cls, w, h, (0, 0))
    want = dst.copy()
    O = ffi.oracle()
    for do, so, off, edge, cls, w, h in blocks:
        o16 = off.astype(np.int16)
        if edge:
            O.ffo_hevc_sao_edge(C.cast(want.ctypes.data + do, u8p), C.cast(src.ctypes.data + so, u8p), sd, ss, ptr(o16, ffi.i16p), cls, w, h)
        else:
            O.ffo_hevc_sao_band(C.cast(want.ctypes.data + do, u8p), C.cast(src.ctypes.data + so, u8p), sd, ss, ptr(o16, ffi.i16p), cls, w, h)
    d_dst, d_src = torch.from_numpy(dst.copy()).cuda(), torch.from_numpy(src).cuda()
    hevc.sao_batch(d_dst, sd, d_src, ss, torch.from_numpy(rec.view(np.uint8).reshape(n, 24).copy()).cuda(), n)
    torch.cuda.synchronize()
    assert (want != dst).sum() > 10000
    assert np.array_equal(d_dst.cpu().numpy(), want)


def test_hevc_sao_host_faces():
    from ffmpeg_amd import hevc
    _torch()
    c = hevc.dsp_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(51)
    for rep in range(12):
        w = int(rng.choice([8, 16, 32, 48, 64])); h = int(rng.choice([8, 16, 64]))
        idx = {8: 0, 16: 1, 32: 2, 48: 3, 64: 4}[w]
        src = rng.integers(0, 256, (h + 2, 192), dtype=np.uint8)
        off = rng.integers(-7, 8, 5).astype(np.int16)
        dst0 = rng.integers(0, 256, (h, 80), dtype=np.uint8)
        a, b = dst0.copy(), dst0.copy()
        left = int(rng.integers(0, 32))
        c.sao_band_filter[idx](a.ctypes.data, src.ctypes.data + 193, 80, 192, off.ctypes.data, left, w, h)
        O.ffo_hevc_sao_band(ptr(b), C.cast(src.ctypes.data + 193, u8p), 80, 192, ptr(off, ffi.i16p), left, w, h)
        assert np.array_equal(a, b)
        off[0] = 0
        eo = rep % 4
        a, b = dst0.copy(), dst0.copy()
        c.sao_edge_filter[idx](a.ctypes.data, src.ctypes.data + 193, 80, off.ctypes.data, eo, w, h)
        O.ffo_hevc_sao_edge(ptr(b), C.cast(src.ctypes.data + 193, u8p), 80, 192, ptr(off, ffi.i16p), eo, w, h)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("old", ["default", "0", "1"])
@pytest.mark.parametrize("uni", [0, 1])
@pytest.mark.parametrize("chroma", [0, 1])
def test_hevc_mc_batch(chroma, uni, old, monkeypatch):
    """prediction blocks of all 10 widths x fractional positions in one batch (tests/checkasm/hevc_pel.c shapes); both kernels"""
    from ffmpeg_amd import hevc
    torch = _torch()
    if old != "default":   # a knob selects libffhip_measure.so (conftest.py); "default" is the product library
        monkeypatch.setenv("FFHIP_HEVC_MC_OLD", old)
    rng = np.random.default_rng(60 + 2 * chroma + uni)
    W, H, P = 512, 1024, 16
    ss = W + 2 * P + 3
    ref = rng.integers(0, 256, (H + 2 * P, ss), dtype=np.uint8)
    ref[:60] = rng.choice(np.array([0, 255], np.uint8), (60, ss))
    nfrac = 8 if chroma else 4
    widths = [2, 4, 6, 8, 12, 16, 24, 32, 48, 64]
    blocks = []
    for by in range(0, H, 64):
        for bx in range(0, W, 64):
            w = int(rng.choice(widths)); h = int(rng.choice([2, 4, 8, 16, 32, 64]))
            dy, dx = rng.integers(-8, 9, 2)
            blocks.append((by, bx, (by + P + int(dy)) * ss + bx + P + int(dx), w, h, int(rng.integers(0, nfrac)), int(rng.integers(0, nfrac))))
    n = len(blocks)
    rec = np.zeros(n, hevc.MC_DTYPE)
    O = ffi.oracle()
    if uni:
        sd = W + (8 if chroma else 9)     # dword-aligned rows (packed stores) and odd ones (byte stores)
        dst = rng.integers(0, 256, (H, sd), dtype=np.uint8)
        want = dst.copy()
        for i, (by, bx, so, w, h, mx, my) in enumerate(blocks):
            rec[i] = (by * sd + bx, so, w, h, mx, my)
            O.ffo_hevc_mc(chroma, 1, want.ctypes.data + by * sd + bx, sd, C.cast(ref.ctypes.data + so, u8p), ss, h, mx, my, w)
        d_dst = torch.from_numpy(dst.copy()).cuda()
    else:
        sd = 0
        dst = np.full((n, 66, 64), -7, np.int16)
        want = dst.copy()
        for i, (by, bx, so, w, h, mx, my) in enumerate(blocks):
            do = i * 66 * 64 + (i & 1)    # every other block off the 8-byte grid
            rec[i] = (do, so, w, h, mx, my)
            O.ffo_hevc_mc(chroma, 0, want.ctypes.data + 2 * do, 0, C.cast(ref.ctypes.data + so, u8p), ss, h, mx, my, w)
        d_dst = torch.from_numpy(dst.copy()).cuda()
    hevc.mc_batch(chroma, uni, d_dst, sd, torch.from_numpy(ref).cuda(), ss, torch.from_numpy(rec.view(np.uint8).reshape(n, 12).copy()).cuda(), n)
    torch.cuda.synchronize()
    assert (want != dst).sum() > 1000
    assert np.array_equal(d_dst.cpu().numpy(), want)


@pytest.mark.parametrize("m", ["default", "0"])
@pytest.mark.parametrize("case", ["all16", "mixed_sizes", "unaligned_dst", "ragged_n", "big_blocks"])
def test_hevc_qpel_uni16_matrix_cores(case, m, monkeypatch):
    """put_hevc_qpel_uni with a 16-byte-aligned source stride: the batch's 16 x 16 blocks run on k_hevc_qpel_m (hevc_qpel_m.hip), the
    rest on k_hevc_mc in a second launch that skips them.  Every (mx, my), every source alignment modulo 16, saturating content (0 / 255
    stripes), blocks of other sizes in between, destinations off the dword grid, batch sizes that leave a wave's group of four partly
    empty — against the oracle, and against the same batch with the matrix-core kernel switched off (FFHIP_HEVC_MC_M=0)"""
    from ffmpeg_amd import hevc
    torch = _torch()
    if m != "default":
        monkeypatch.setenv("FFHIP_HEVC_MC_M", m)
    rng = np.random.default_rng({"all16": 1, "mixed_sizes": 2, "unaligned_dst": 3, "ragged_n": 4, "big_blocks": 5}[case])
    W, H, P = 512, 512, 24
    ss = W + 2 * P                       # 560 = 16 * 35
    ref = rng.integers(0, 256, (H + 2 * P, ss), dtype=np.uint8)
    ref[:120] = rng.choice(np.array([0, 255], np.uint8), (120, ss))
    ref[200:230, ::2] = 255; ref[200:230, 1::2] = 0
    sd = W + (4 if case != "unaligned_dst" else 8)
    blocks = []
    i = 0
    step = 64 if case == "big_blocks" else 16
    for by in range(0, H, step):
        for bx in range(0, W, step):
            w = h = 16
            if case == "mixed_sizes" and rng.integers(0, 3) == 0:
                w, h = int(rng.choice([4, 8, 12, 16])), int(rng.choice([4, 8, 16]))
                if w == 16 and h == 16:
                    h = 8
            if case == "big_blocks":     # every luma width of the reference's tables, tiles cut off at the block's edge (round 6)
                w, h = int(rng.choice([4, 8, 12, 16, 24, 32, 48, 64])), int(rng.choice([4, 8, 12, 16, 24, 32, 48, 64]))
            dy, dx = rng.integers(-20, 21, 2)
            doff = by * sd + bx + (int(rng.integers(0, 4)) if case == "unaligned_dst" and bx + 20 < W else 0)
            blocks.append((doff, (by + P + int(dy)) * ss + bx + P + int(dx), w, h, i & 3, (i >> 2) & 3))
            i += 1
    if case == "ragged_n":
        blocks = blocks[:1021]
    if case == "unaligned_dst":
        blocks = blocks[::2]             # shifted destinations must not overlap their neighbours
    n = len(blocks)
    rec = np.zeros(n, hevc.MC_DTYPE)
    O = ffi.oracle()
    dst = rng.integers(0, 256, (H + 1, sd), dtype=np.uint8)
    want = dst.copy()
    for j, (do, so, w, h, mx, my) in enumerate(blocks):
        rec[j] = (do, so, w, h, mx, my)
        O.ffo_hevc_mc(0, 1, want.ctypes.data + do, sd, C.cast(ref.ctypes.data + so, u8p), ss, h, mx, my, w)
    assert len({(b[1] - 3 - 3 * ss) & 15 for b in blocks}) == 16 or case == "big_blocks"
    d_dst = torch.from_numpy(dst.copy()).cuda()
    hevc.mc_batch(0, 1, d_dst, sd, torch.from_numpy(ref).cuda(), ss, torch.from_numpy(rec.view(np.uint8).reshape(n, 12).copy()).cuda(), n)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    assert (want != dst).sum() > (20000 if case == "big_blocks" else 100000)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (case, bad[:5], len(bad))


@pytest.mark.parametrize("mode", [0, 2, 3, 4])
@pytest.mark.parametrize("case", ["all16", "mixed_sizes", "ragged_unaligned", "big_blocks"])
def test_hevc_qpel16_matrix_cores_other_stages(case, mode, monkeypatch):
    """the other output stages of k_hevc_qpel_m — put (int16 rows of 64), uni_w, bi, bi_w — on batches of 16 x 16 luma blocks with a
    16-byte-aligned source stride: every (mx, my) and source alignment, saturating content, weights over the slice header's ranges and
    checkasm's ladders, the other list's block on and off the 8-byte grid, other block sizes mixed in, a ragged batch"""
    from ffmpeg_amd import hevc
    torch = _torch()
    rng = np.random.default_rng(100 + 10 * mode + len(case))
    W, H, P = 256, 512, 24
    ss = W + 2 * P + 8                   # 312 = 16 * 19.5 -> make it a multiple of 16
    ss = (ss + 15) & ~15
    ref = rng.integers(0, 256, (H + 2 * P, ss), dtype=np.uint8)
    ref[:120] = rng.choice(np.array([0, 255], np.uint8), (120, ss))
    ref[200:230, ::2] = 255; ref[200:230, 1::2] = 0
    sd = W + 4
    blocks = []
    i = 0
    step = 64 if case == "big_blocks" else 16
    for by in range(0, H, step):
        for bx in range(0, W, step):
            w = h = 16
            if case == "mixed_sizes" and rng.integers(0, 3) == 0:
                w, h = int(rng.choice([4, 8, 12, 16])), int(rng.choice([4, 8, 16]))
                if w == 16 and h == 16:
                    w = 8
            if case == "big_blocks":
                w, h = int(rng.choice([4, 8, 12, 16, 24, 32, 48, 64])), int(rng.choice([4, 8, 12, 16, 24, 32, 48, 64]))
            dy, dx = rng.integers(-20, 21, 2)
            blocks.append((by, bx, (by + P + int(dy)) * ss + bx + P + int(dx), w, h, i & 3, (i >> 2) & 3))
            i += 1
    if case == "ragged_unaligned":
        blocks = blocks[:509]
    n = len(blocks)
    O = ffi.oracle()
    d_ref = torch.from_numpy(ref).cuda()
    if mode == 0:
        rec = np.zeros(n, hevc.MC_DTYPE)
        dst = np.full((n, 65, 64), -7, np.int16)
        want = dst.copy()
        for j, (by, bx, so, w, h, mx, my) in enumerate(blocks):
            do = j * 65 * 64 + ((j & 1) if case == "ragged_unaligned" else 0)
            rec[j] = (do, so, w, h, mx, my)
            O.ffo_hevc_mc(0, 0, want.ctypes.data + 2 * do, 0, C.cast(ref.ctypes.data + so, u8p), ss, h, mx, my, w)
        d_dst = torch.from_numpy(dst.copy()).cuda()
        hevc.mc_batch(0, 0, d_dst, 0, d_ref, ss, torch.from_numpy(rec.view(np.uint8).reshape(n, 12).copy()).cuda(), n)
    else:
        rec = np.zeros(n, hevc.MCW_DTYPE)
        src2 = rng.integers(-8192, 16384, (n + 1, 64, 64)).astype(np.int16)
        src2[::5] = 16383
        src2[1::7] = -8192
        flat2 = src2.reshape(-1)
        dst = rng.integers(0, 256, (H + 1, sd), dtype=np.uint8)
        want = dst.copy()
        for j, (by, bx, so, w, h, mx, my) in enumerate(blocks):
            d, wx0, wx1, ox = _weights(rng, j)
            o2 = j * 4096 + (1 if (case == "ragged_unaligned" and j % 3 == 0) else 0)
            do = by * sd + bx
            rec[j] = (do, so, o2, w, h, mx, my, wx0, wx1, ox, d, 0)
            O.ffo_hevc_mc_w(0, mode, C.cast(want.ctypes.data + do, u8p), sd, C.cast(ref.ctypes.data + so, u8p), ss,
                            C.cast(flat2.ctypes.data + 2 * o2, ffi.i16p), h, d, wx0, wx1, ox, mx, my, w)
        d_dst = torch.from_numpy(dst.copy()).cuda()
        hevc.mc_w_batch(0, mode, d_dst, sd, d_ref, ss, torch.from_numpy(src2).cuda() if mode != 2 else None,
                        torch.from_numpy(rec.view(np.uint8).reshape(n, 24).copy()).cuda(), n)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    assert (want != dst).sum() > (8000 if case == "big_blocks" else 50000)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (case, mode, bad[:5], len(bad))


def test_hevc_mc_host_faces():
    from ffmpeg_amd import hevc
    _torch()
    c = hevc.dsp_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(61)
    src = rng.integers(0, 256, (90, 100), dtype=np.uint8)
    widths = [2, 4, 6, 8, 12, 16, 24, 32, 48, 64]
    for rep in range(16):
        chroma = rep & 1
        idx = int(rng.integers(0, 10)); w = widths[idx]; h = int(rng.choice([2, 8, 64]))
        mx, my = (int(v) for v in rng.integers(0, 8 if chroma else 4, 2))
        sp = src.ctypes.data + 10 * 100 + 12
        a16, b16 = np.zeros((64, 64), np.int16), np.zeros((64, 64), np.int16)
        (c.put_hevc_epel if chroma else c.put_hevc_qpel)[idx][int(bool(my))][int(bool(mx))](a16.ctypes.data, sp, 100, h, mx, my, w)
        O.ffo_hevc_mc(chroma, 0, b16.ctypes.data, 0, C.cast(sp, u8p), 100, h, mx, my, w)
        assert np.array_equal(a16, b16), (chroma, w, h, mx, my)
        a8, b8 = np.full((64, 72), 9, np.uint8), np.full((64, 72), 9, np.uint8)
        (c.put_hevc_epel_uni if chroma else c.put_hevc_qpel_uni)[idx][int(bool(my))][int(bool(mx))](a8.ctypes.data, 72, sp, 100, h, mx, my, w)
        O.ffo_hevc_mc(chroma, 1, b8.ctypes.data, 72, C.cast(sp, u8p), 100, h, mx, my, w)
        assert np.array_equal(a8, b8), (chroma, w, h, mx, my, "uni")


def _weights(rng, rep):
    """(denom, wx0, wx1, ox): slice-header ranges mixed with tests/checkasm/hevc_pel.c's ladders"""
    if rep % 3 == 0:
        return int(rng.choice([0, 7, 12])), int(rng.choice([0, 128, 255])), int(rng.choice([0, 128, 255])), int(rng.choice([0, 255]))
    d = int(rng.integers(0, 8))
    return d, (1 << d) + int(rng.integers(-128, 128)), (1 << d) + int(rng.integers(-128, 128)), int(rng.integers(-256, 255))


@pytest.mark.parametrize("old", ["default", "0", "1"])
@pytest.mark.parametrize("mode", [2, 3, 4])
@pytest.mark.parametrize("chroma", [0, 1])
def test_hevc_mc_weighted_batch(chroma, mode, old, monkeypatch):
    """put_hevc_{qpel,epel}_{uni_w,bi,bi_w}: all 10 widths x fractional positions x weights in one batch; both kernels"""
    from ffmpeg_amd import hevc
    torch = _torch()
    if old != "default":   # a knob selects libffhip_measure.so (conftest.py); "default" is the product library
        monkeypatch.setenv("FFHIP_HEVC_MC_OLD", old)
    rng = np.random.default_rng(160 + 8 * chroma + mode)
    W, H, P = 512, 1024, 16
    ss = W + 2 * P + 3
    ref = rng.integers(0, 256, (H + 2 * P, ss), dtype=np.uint8)
    ref[:60] = rng.choice(np.array([0, 255], np.uint8), (60, ss))
    nfrac = 8 if chroma else 4
    widths = [2, 4, 6, 8, 12, 16, 24, 32, 48, 64]
    sd = W + 8 + (mode & 1)
    dst = rng.integers(0, 256, (H, sd), dtype=np.uint8)
    want = dst.copy()
    blocks = [(by, bx) for by in range(0, H, 64) for bx in range(0, W, 64)]
    n = len(blocks)
    src2 = rng.integers(-8192, 16384, (n + 1, 64, 64)).astype(np.int16)
    src2[::5] = 16383
    src2[1::7] = -8192
    flat2 = src2.reshape(-1)
    rec = np.zeros(n, hevc.MCW_DTYPE)
    O = ffi.oracle()
    for i, (by, bx) in enumerate(blocks):
        w = int(rng.choice(widths)); h = int(rng.choice([2, 4, 8, 16, 32, 64]))
        dy, dx = (int(v) for v in rng.integers(-8, 9, 2))
        so = (by + P + dy) * ss + bx + P + dx
        mx, my = int(rng.integers(0, nfrac)), int(rng.integers(0, nfrac))
        d, wx0, wx1, ox = _weights(rng, i)
        o2 = i * 4096 + (1 if i % 3 == 0 else 0)    # a third of the blocks off the 8-byte grid
        rec[i] = (by * sd + bx, so, o2, w, h, mx, my, wx0, wx1, ox, d, 0)
        O.ffo_hevc_mc_w(chroma, mode, C.cast(want.ctypes.data + by * sd + bx, u8p), sd, C.cast(ref.ctypes.data + so, u8p), ss,
                        C.cast(flat2.ctypes.data + 2 * o2, ffi.i16p), h, d, wx0, wx1, ox, mx, my, w)
    d_dst = torch.from_numpy(dst.copy()).cuda()
    hevc.mc_w_batch(chroma, mode, d_dst, sd, torch.from_numpy(ref).cuda(), ss, torch.from_numpy(src2).cuda() if mode != 2 else None,
                    torch.from_numpy(rec.view(np.uint8).reshape(n, 24).copy()).cuda(), n)
    torch.cuda.synchronize()
    assert (want != dst).sum() > 1000
    assert np.array_equal(d_dst.cpu().numpy(), want)


def test_hevc_mc_weighted_host_faces():
    from ffmpeg_amd import hevc
    _torch()
    c = hevc.dsp_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(62)
    src = rng.integers(0, 256, (90, 100), dtype=np.uint8)
    widths = [2, 4, 6, 8, 12, 16, 24, 32, 48, 64]
    for rep in range(18):
        chroma = rep & 1
        idx = int(rng.integers(0, 10)); w = widths[idx]; h = int(rng.choice([2, 8, 64]))
        mx, my = (int(v) for v in rng.integers(0, 8 if chroma else 4, 2))
        a, b = int(bool(my)), int(bool(mx))
        sp = src.ctypes.data + 10 * 100 + 12
        # exactly as large as the reference reads: (h - 1) rows of 64 plus w elements
        src2 = rng.integers(-8192, 16384, (h - 1) * 64 + w).astype(np.int16)
        full2 = np.zeros(64 * 64, np.int16); full2[:src2.size] = src2
        d, wx0, wx1, ox = _weights(rng, rep)
        for mode in (2, 3, 4):
            a8, b8 = np.full((64, 72), 9, np.uint8), np.full((64, 72), 9, np.uint8)
            if mode == 2:
                (c.put_hevc_epel_uni_w if chroma else c.put_hevc_qpel_uni_w)[idx][a][b](a8.ctypes.data, 72, sp, 100, h, d, wx0, ox, mx, my, w)
            elif mode == 3:
                (c.put_hevc_epel_bi if chroma else c.put_hevc_qpel_bi)[idx][a][b](a8.ctypes.data, 72, sp, 100, src2.ctypes.data, h, mx, my, w)
            else:
                (c.put_hevc_epel_bi_w if chroma else c.put_hevc_qpel_bi_w)[idx][a][b](a8.ctypes.data, 72, sp, 100, src2.ctypes.data, h, d, wx0,
                                                                                      wx1, ox, mx, my, w)
            O.ffo_hevc_mc_w(chroma, mode, ptr(b8), 72, C.cast(sp, u8p), 100, ptr(full2, ffi.i16p), h, d, wx0, wx1, ox, mx, my, w)
            assert np.array_equal(a8, b8), (chroma, mode, w, h, mx, my, d, wx0, wx1, ox)


@pytest.mark.parametrize("lg", [2, 3, 4, 5])
def test_hevc_dequant_rdpcm(lg):
    """dequant and transform_rdpcm (both modes): batch face on 300 units, host faces on one"""
    from ffmpeg_amd import hevc
    torch = _torch()
    n = 1 << lg
    rng = np.random.default_rng(300 + lg)
    nu = 300
    c0 = rng.integers(-32768, 32768, (nu, n * n)).astype(np.int16)
    c0[::3] = rng.integers(-300, 301, (len(c0[::3]), n * n))
    O = ffi.oracle()
    tus = np.zeros(nu, hevc.TU_DTYPE)
    tus["coeff_offset"] = np.arange(nu) * n * n
    tus["dst_offset"] = -1
    d_tus = torch.from_numpy(tus.view(np.uint8).reshape(nu, 12).copy()).cuda()
    ctx = hevc.dsp_init(8)
    for kind, fn in ((hevc.DEQUANT, lambda p: O.ffo_hevc_dequant(p, lg)), (hevc.RDPCM_H, lambda p: O.ffo_hevc_transform_rdpcm(p, lg, 0)),
                     (hevc.RDPCM_V, lambda p: O.ffo_hevc_transform_rdpcm(p, lg, 1))):
        want = c0.copy()
        for u in range(nu):
            fn(ptr(want[u], ffi.i16p))
        d_c = torch.from_numpy(c0.copy()).cuda()
        hevc.idct_batch(kind, lg, d_c, None, 0, d_tus, nu)
        torch.cuda.synchronize()
        assert np.array_equal(d_c.cpu().numpy(), want), kind
        one = c0[7].copy()
        if kind == hevc.DEQUANT:
            ctx.dequant(one.ctypes.data, lg)
        else:
            ctx.transform_rdpcm(one.ctypes.data, lg, int(kind == hevc.RDPCM_V))
        assert np.array_equal(one, want[7]), ("host", kind)


def _restore_case(rng, rep):
    p = .5 if rep % 3 else .85
    return (rep & 1, int(rng.integers(0, 4)), int(rng.integers(-60, 61)), (rng.random(4) < p).astype(np.int32),
            int(rng.choice([2, 3, 8, 16, 33, 64])), int(rng.choice([2, 3, 8, 16, 33, 64])), (rng.random(2) < p).astype(np.uint8),
            (rng.random(2) < p).astype(np.uint8), (rng.random(4) < p).astype(np.uint8))


def test_hevc_sao_edge_restore():
    """sao_edge_restore[0] / [1]: 600 blocks with every combination of border / edge flags in one batch, then the host faces"""
    from ffmpeg_amd import hevc
    torch = _torch()
    rng = np.random.default_rng(310)
    O = ffi.oracle()
    nb = 600
    per_row = 20
    sd, ss = per_row * 64 + 7, per_row * 64 + 12
    rows = (nb + per_row - 1) // per_row
    src = rng.integers(0, 256, (rows * 64, ss), dtype=np.uint8)
    dst = rng.integers(0, 256, (rows * 64, sd), dtype=np.uint8)
    want = dst.copy()
    rec = np.zeros(nb, hevc.RESTORE_DTYPE)
    cases = []
    for i in range(nb):
        var, eo, off, borders, w, h, ve, he, de = _restore_case(rng, i)
        by, bx = (i // per_row) * 64, (i % per_row) * 64
        bits = lambda a: int(sum(int(bool(v)) << k for k, v in enumerate(a)))
        rec[i] = (by * sd + bx, by * ss + bx, off, w, h, eo, var, bits(borders), bits(ve), bits(he), bits(de), (0, 0))
        O.ffo_hevc_sao_edge_restore(var, C.cast(want.ctypes.data + by * sd + bx, u8p), C.cast(src.ctypes.data + by * ss + bx, u8p), sd, ss, eo, off,
                                    ptr(borders, ffi.i32p), w, h, ptr(ve), ptr(he), ptr(de))
        cases.append((var, eo, off, borders, w, h, ve, he, de, by, bx))
    d_dst, d_src = torch.from_numpy(dst.copy()).cuda(), torch.from_numpy(src).cuda()
    hevc.sao_restore_batch(d_dst, sd, d_src, ss, torch.from_numpy(rec.view(np.uint8).reshape(nb, 20).copy()).cuda(), nb)
    torch.cuda.synchronize()
    assert (want != dst).sum() > 2000
    got = d_dst.cpu().numpy()
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    ctx = hevc.dsp_init(8)
    for var, eo, off, borders, w, h, ve, he, de, by, bx in cases[:24]:
        sao = hevc.SAOParams()
        c_idx = int(rng.integers(0, 3))
        sao.eo_class[c_idx] = eo
        sao.offset_val[c_idx][0] = off
        a = dst[by:by + 64, bx:bx + 64].copy()
        s_ = src[by:by + 64, bx:bx + 64].copy()
        ctx.sao_edge_restore[var](a.ctypes.data, s_.ctypes.data, 64, 64, C.addressof(sao), borders.ctypes.data, w, h, c_idx, ve.ctypes.data,
                                  he.ctypes.data, de.ctypes.data)
        assert np.array_equal(a, want[by:by + 64, bx:bx + 64]), (var, eo, w, h)
