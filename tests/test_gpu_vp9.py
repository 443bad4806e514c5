"""GPU parity: VP9 inverse transforms + add (VP9DSPContext.itxfm_add[tx][txtp]) vs the oracle, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p, i16p
from test_oracle_vs_ref import vp9_block
import test_golden as G

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("tx", range(5))
def test_vp9_itxfm_batch(tx):
    """a picture's worth of blocks of one size: all four types mixed, dense / sparse / dc-only / wrap-around content, picture rows on
    and off the dword grid; the blocks are consumed exactly as the reference consumes them"""
    from ffmpeg_amd import vp9
    torch = _torch()
    rng = np.random.default_rng(400 + tx)
    n = 4 if tx == 4 else 4 << tx
    gx, gy = 24, 10
    nb = gx * gy
    stride = gx * n + 5
    dst = rng.integers(0, 256, (gy * n, stride), dtype=np.uint8)
    want = dst.copy()
    coeffs = np.zeros((nb + 1, n * n), np.int16)
    wcoef = coeffs.copy()
    rec = np.zeros(nb, vp9.TU_DTYPE)
    O = ffi.oracle()
    order = rng.permutation(nb)
    for j, b in enumerate(order):                   # blocks listed in scrambled order
        kind = int(rng.integers(0, 5))
        txtp = int(rng.integers(0, 4))
        blk = vp9_block(rng, n, kind)
        eob = 1 if kind == 2 else n * n
        by, bx = divmod(int(b), gx)
        coeffs[b] = blk
        rec[j] = (b * n * n, by * n * stride + bx * n, txtp, int(eob == 1), (0, 0))
        wb = blk.copy()
        O.ffo_vp9_itxfm_add(tx, txtp, C.cast(want.ctypes.data + by * n * stride + bx * n, u8p), stride, ptr(wb, i16p), eob)
        wcoef[b] = wb
    coeffs[nb] = wcoef[nb] = 1234                    # a block nobody names stays untouched
    d_dst, d_co = torch.from_numpy(dst.copy()).cuda(), torch.from_numpy(coeffs.copy()).cuda()
    vp9.itxfm_add_batch(tx, d_co, d_dst, stride, torch.from_numpy(rec.view(np.uint8).reshape(nb, 12).copy()).cuda(), nb)
    torch.cuda.synchronize()
    assert (want != dst).sum() > 1000
    assert np.array_equal(d_dst.cpu().numpy(), want)
    assert np.array_equal(d_co.cpu().numpy(), wcoef)


def test_vp9_itxfm_host_faces():
    from ffmpeg_amd import vp9
    _torch()
    c = vp9.dsp_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(410)
    for tx in range(5):
        n = 4 if tx == 4 else 4 << tx
        for txtp in range(4):
            for kind in (1, 2):
                blk = vp9_block(rng, n, kind)
                eob = 1 if kind == 2 else n * n
                dst0 = rng.integers(0, 256, (n, n + 7), dtype=np.uint8)
                a, b, ba, bb = dst0.copy(), dst0.copy(), blk.copy(), blk.copy()
                c.itxfm_add[tx][txtp](a.ctypes.data, n + 7, ba.ctypes.data, eob)
                O.ffo_vp9_itxfm_add(tx, txtp, ptr(b), n + 7, ptr(bb, i16p), eob)
                assert np.array_equal(a, b) and np.array_equal(ba, bb), (tx, txtp, kind)


def test_vp9_golden_gpu():
    from ffmpeg_amd import vp9
    torch = _torch()
    d = G.load("vp9")
    for tx in range(5):
        n = 4 if tx == 4 else 4 << tx
        par = d["tx%d_par" % tx]
        nb = len(par)
        rec = np.zeros(nb, vp9.TU_DTYPE)
        rec["coeff_offset"] = np.arange(nb) * n * n
        rec["dst_offset"] = np.arange(nb) * n * n
        rec["txtp"], rec["dc_only"] = par[:, 0], par[:, 1] == 1
        d_dst = torch.from_numpy(np.ascontiguousarray(d["tx%d_dst" % tx])).cuda()      # block i = rows of width n, pitch n
        d_co = torch.from_numpy(np.ascontiguousarray(d["tx%d_blk" % tx])).cuda()
        vp9.itxfm_add_batch(tx, d_co, d_dst, n, torch.from_numpy(rec.view(np.uint8).reshape(nb, 12).copy()).cuda(), nb)
        torch.cuda.synchronize()
        assert np.array_equal(d_dst.cpu().numpy(), d["tx%d_out" % tx]), tx
        assert np.array_equal(d_co.cpu().numpy(), d["tx%d_oblk" % tx]), tx


@pytest.mark.parametrize("aligned", [0, 1])
def test_vp9_mc_batch(aligned):
    """a picture's worth of prediction blocks: 5 widths x heights x 4 filters x all (mx, my) classes x put / avg in one batch;
    destination rows on and off the dword grid"""
    from ffmpeg_amd import vp9
    torch = _torch()
    rng = np.random.default_rng(420 + aligned)
    W, H, P = 640, 512, 16
    ss = W + 2 * P + 3
    sd = W + (8 if aligned else 9)
    ref = rng.integers(0, 256, (H + 2 * P, ss), dtype=np.uint8)
    ref[:60] = rng.choice(np.array([0, 255], np.uint8), (60, ss))
    dst = rng.integers(0, 256, (H, sd), dtype=np.uint8)
    want = dst.copy()
    O = ffi.oracle()
    recs = []
    for by in range(0, H, 64):
        for bx in range(0, W, 64):
            w = int(rng.choice([4, 8, 16, 32, 64])); h = int(rng.choice([1, 2, 4, 8, 16, 32, 64]))
            f, avg = int(rng.integers(0, 4)), int(rng.integers(0, 2))
            mx, my = (int(v) for v in rng.integers(0, 16, 2))
            r = rng.random()
            if r < .15:
                mx = 0
            elif r < .3:
                my = 0
            elif r < .35:
                mx = my = 0
            dy, dx = (int(v) for v in rng.integers(-8, 9, 2))
            so = (by + P + dy) * ss + bx + P + dx
            recs.append((by * sd + bx, so, w, h, f, mx, my, avg, (0, 0)))
            O.ffo_vp9_mc(f, avg, C.cast(want.ctypes.data + by * sd + bx, u8p), sd, C.cast(ref.ctypes.data + so, u8p), ss, w, h, mx, my)
    n = len(recs)
    rec = np.array(recs, vp9.MC_DTYPE)
    d_dst = torch.from_numpy(dst.copy()).cuda()
    vp9.mc_batch(d_dst, sd, torch.from_numpy(ref).cuda(), ss, torch.from_numpy(rec.view(np.uint8).reshape(n, 16).copy()).cuda(), n)
    torch.cuda.synchronize()
    assert (want != dst).sum() > 1000
    got = d_dst.cpu().numpy()
    assert np.array_equal(got, want), "%d mismatches, first %s" % ((got != want).sum(), np.argwhere(got != want)[:3])


@pytest.mark.parametrize("m", ["default", "0"])
@pytest.mark.parametrize("case", ["all16", "mixed_sizes", "unaligned_dst", "ragged_n", "big_blocks"])
def test_vp9_mc16_matrix_cores(case, m, monkeypatch):
    """a 16-byte-aligned source stride: the batch's 16 x 16 blocks run on k_vp9_mc_m (vp9_mc.hip), the rest on k_vp9_mc in a second
    launch that skips them.  All four filters x every (mx, my) of the 16 x 16 grid, put and avg, every source alignment modulo 16,
    saturating content, other block sizes in between, destinations off the dword grid, a batch that leaves a wave's group of four partly
    empty — against the oracle, with and without the matrix-core kernel (FFHIP_VP9_MC_M=0)"""
    from ffmpeg_amd import vp9
    torch = _torch()
    if m != "default":
        monkeypatch.setenv("FFHIP_VP9_MC_M", m)
    rng = np.random.default_rng({"all16": 11, "mixed_sizes": 12, "unaligned_dst": 13, "ragged_n": 14, "big_blocks": 15}[case])
    W, H, P = 1024, 512, 24
    ss = W + 2 * P                       # 1072 = 16 * 67
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
                w, h = int(rng.choice([4, 8, 16])), int(rng.choice([2, 4, 8, 16]))
                if w == 16 and h == 16:
                    h = 8
            if case == "big_blocks":     # every width, heights down to one row: tiles cut off at the block's edge (round 6)
                w, h = int(rng.choice([4, 8, 16, 32, 64])), int(rng.choice([1, 3, 8, 16, 24, 32, 33, 64]))
            dy, dx = rng.integers(-20, 21, 2)
            doff = by * sd + bx + (int(rng.integers(0, 4)) if case == "unaligned_dst" and bx + 20 < W else 0)
            f = (i >> 8) & 3
            blocks.append((doff, (by + P + int(dy)) * ss + bx + P + int(dx), w, h, f, i & 15, (i >> 4) & 15, int(rng.integers(0, 2)), (0, 0)))
            i += 1
    if case == "ragged_n":
        blocks = blocks[:2045]
    if case == "unaligned_dst":
        blocks = blocks[::2]             # shifted destinations must not overlap their neighbours
    n = len(blocks)
    O = ffi.oracle()
    dst = rng.integers(0, 256, (H + 1, sd), dtype=np.uint8)
    want = dst.copy()
    for (do, so, w, h, f, mx, my, avg, _) in blocks:
        O.ffo_vp9_mc(f, avg, C.cast(want.ctypes.data + do, u8p), sd, C.cast(ref.ctypes.data + so, u8p), ss, w, h, mx, my)
    assert len({(b[1] - 3 - 3 * ss) & 15 for b in blocks}) == 16 or case == "big_blocks"
    rec = np.array(blocks, vp9.MC_DTYPE)
    d_dst = torch.from_numpy(dst.copy()).cuda()
    vp9.mc_batch(d_dst, sd, torch.from_numpy(ref).cuda(), ss, torch.from_numpy(rec.view(np.uint8).reshape(n, 16).copy()).cuda(), n)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    assert (want != dst).sum() > (30000 if case == "big_blocks" else 100000)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (case, bad[:5], len(bad))


def test_vp9_mc_host_faces():
    from ffmpeg_amd import vp9
    _torch()
    c = vp9.mc_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(421)
    src = rng.integers(0, 256, (90, 100), dtype=np.uint8)
    for rep in range(32):
        idx = rep % 5
        w = 64 >> idx
        f, avg = (rep // 5) % 4, rep & 1
        h = int(rng.choice([2, 8, 64]))
        mx, my = (int(v) for v in rng.integers(1, 16, 2))
        hx, vy = (rep >> 1) & 1, (rep >> 2) & 1
        sp = src.ctypes.data + 10 * 100 + 12
        d0 = rng.integers(0, 256, (64, 72), dtype=np.uint8)
        a, b = d0.copy(), d0.copy()
        c.mc[idx][f][avg][hx][vy](a.ctypes.data, 72, sp, 100, h, mx, my)
        O.ffo_vp9_mc(f, avg, ptr(b), 72, C.cast(sp, u8p), 100, w, h, mx if hx else 0, my if vy else 0)
        assert np.array_equal(a, b), (idx, f, avg, hx, vy, h, mx, my)


def test_vp9_mc_golden_gpu():
    from ffmpeg_amd import vp9
    torch = _torch()
    d = G.load("vp9")
    par = d["mc_par"]
    n = len(par)
    rec = np.zeros(n, vp9.MC_DTYPE)
    rec["dst_offset"] = np.arange(n) * 4096
    rec["src_offset"] = par[:, 6] * 96 + par[:, 7]
    rec["filter"], rec["avg"], rec["width"], rec["height"], rec["mx"], rec["my"] = par[:, 0], par[:, 1], par[:, 2], par[:, 3], par[:, 4], par[:, 5]
    base = np.ascontiguousarray(d["mc_in"])
    d_dst = torch.from_numpy(np.stack([base] * n)).cuda()
    vp9.mc_batch(d_dst, 64, torch.from_numpy(np.ascontiguousarray(d["mc_ref"])).cuda(), 96, torch.from_numpy(rec.view(np.uint8).reshape(n, 16).copy()).cuda(), n)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    for i, (f, avg, w, h, mx, my, y0, x0) in enumerate(par.tolist()):
        assert np.array_equal(got[i][:h, :w], d["mc_out"][i][:h, :w]), i
        assert np.array_equal(got[i][h:], base[h:]) and np.array_equal(got[i][:, w:], base[:, w:]), i


def test_vp9_loop_filter_batch():
    """disjoint 8-sample segments all over a plane: both directions, the three widths, limits from wide open to tight; content
    that reaches the flat16 / flat8 / narrow / hev branches"""
    from ffmpeg_amd import vp9
    from test_oracle_vs_ref import vp9_lf_plane
    torch = _torch()
    rng = np.random.default_rng(430)
    gy, gx = 14, 20
    plane = np.zeros((gy * 48, gx * 48 + 3), np.uint8)
    for ty in range(gy):
        for tx in range(gx):
            plane[ty * 48:(ty + 1) * 48, tx * 48:(tx + 1) * 48] = vp9_lf_plane(rng)
    stride = plane.shape[1]
    want = plane.copy()
    O = ffi.oracle()
    recs = []
    WD = [4, 8, 16]
    for ty in range(gy):
        for tx in range(gx):
            d, w = int(rng.integers(0, 2)), int(rng.integers(0, 3))
            E, I, H = int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 16))
            if (ty + tx) % 3 == 0:
                E, I = 255, 63
            nseg = int(rng.integers(1, 3))
            for sgm in range(nseg):                    # one or two segments along the edge through the tile's centre
                off = (ty * 48 + 24) * stride + tx * 48 + 24 + 8 * sgm * (1 if d else stride)
                recs.append((off, w, d, E, I, H, (0, 0, 0)))
                O.ffo_vp9_loop_filter(WD[w], d, C.cast(want.ctypes.data + off, u8p), stride, E, I, H)
    n = len(recs)
    rec = np.array(recs, vp9.EDGE_DTYPE)
    d_pl = torch.from_numpy(plane.copy()).cuda()
    vp9.loop_filter_batch(d_pl, stride, torch.from_numpy(rec.view(np.uint8).reshape(n, 12).copy()).cuda(), n)
    torch.cuda.synchronize()
    assert (want != plane).sum() > 2000
    got = d_pl.cpu().numpy()
    assert np.array_equal(got, want), "%d mismatches" % (got != want).sum()


def test_vp9_loop_filter_host_faces():
    from ffmpeg_amd import vp9
    from test_oracle_vs_ref import vp9_lf_plane
    _torch()
    c = vp9.lf_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(431)
    WD = [4, 8, 16]
    for rep in range(36):
        pl = vp9_lf_plane(rng)
        a, b = pl.copy(), pl.copy()
        d = rep & 1
        E, I, H = (255, 63, int(rng.integers(0, 16))) if rep % 3 == 0 else (int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 16)))
        p0 = 24 * 48 + 24
        seg2 = 8 * (1 if d else 48)
        which = rep % 3
        if which == 0:
            w = (rep // 3) % 3
            c.loop_filter_8[w][d](a.ctypes.data + p0, 48, E, I, H)
            O.ffo_vp9_loop_filter(WD[w], d, C.cast(b.ctypes.data + p0, u8p), 48, E, I, H)
        elif which == 1:
            c.loop_filter_16[d](a.ctypes.data + p0, 48, E, I, H)
            for sgm in range(2):
                O.ffo_vp9_loop_filter(16, d, C.cast(b.ctypes.data + p0 + sgm * seg2, u8p), 48, E, I, H)
        else:
            w1, w2 = (rep // 3) & 1, (rep // 6) & 1
            E2, I2, H2 = int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 16))
            c.loop_filter_mix2[w1][w2][d](a.ctypes.data + p0, 48, E | E2 << 8, I | I2 << 8, H | H2 << 8)
            O.ffo_vp9_loop_filter(WD[w1], d, C.cast(b.ctypes.data + p0, u8p), 48, E, I, H)
            O.ffo_vp9_loop_filter(WD[w2], d, C.cast(b.ctypes.data + p0 + seg2, u8p), 48, E2, I2, H2)
        assert np.array_equal(a, b), (rep, which, d)


@pytest.mark.parametrize("tx", range(4))
def test_vp9_intra_pred_batch(tx):
    """hundreds of blocks of one size, all 15 modes, edge lines with extreme content; destination rows on and off the dword grid"""
    from ffmpeg_amd import vp9
    torch = _torch()
    rng = np.random.default_rng(440 + tx)
    n = 4 << tx
    gx, gy = 30, 12
    nb = gx * gy
    stride = gx * n + (4 if tx & 1 else 5)
    dst = rng.integers(0, 256, (gy * n, stride), dtype=np.uint8)
    want = dst.copy()
    slot = n + 1 + max(n, 8)
    edges = rng.integers(0, 256, (nb, slot), dtype=np.uint8)
    edges[::4] = rng.choice(np.array([0, 255], np.uint8), (len(edges[::4]), slot))
    rec = np.zeros(nb, vp9.INTRA_DTYPE)
    O = ffi.oracle()
    for b in range(nb):
        by, bx = divmod(b, gx)
        mode = b % 15
        rec[b] = (by * n * stride + bx * n, b * slot, mode, (0, 0, 0))
        e = edges[b]
        left = np.ascontiguousarray(e[:n])
        topbuf = np.zeros(16 + 64, np.uint8)
        topbuf[15] = e[n]
        topbuf[16:16 + max(n, 8)] = e[n + 1:]
        O.ffo_vp9_intra_pred(tx, mode, C.cast(want.ctypes.data + by * n * stride + bx * n, u8p), stride, ptr(left),
                             C.cast(topbuf.ctypes.data + 16, u8p))
    d_dst = torch.from_numpy(dst.copy()).cuda()
    vp9.intra_pred_batch(tx, d_dst, stride, torch.from_numpy(edges).cuda(), torch.from_numpy(rec.view(np.uint8).reshape(nb, 12).copy()).cuda(), nb)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    assert np.array_equal(got, want), "%d mismatches, first %s" % ((got != want).sum(), np.argwhere(got != want)[:3])


def test_vp9_intra_pred_host_faces():
    from ffmpeg_amd import vp9
    _torch()
    c = vp9.intra_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(441)
    for tx in range(4):
        n = 4 << tx
        for mode in range(15):
            left = rng.integers(0, 256, n, dtype=np.uint8)
            topbuf = rng.integers(0, 256, 16 + 2 * n + 8, dtype=np.uint8)
            a = rng.integers(0, 256, (n, n + 3), dtype=np.uint8)
            b = a.copy()
            c.intra_pred[tx][mode](a.ctypes.data, n + 3, left.ctypes.data, topbuf.ctypes.data + 16)
            O.ffo_vp9_intra_pred(tx, mode, ptr(b), n + 3, ptr(left), C.cast(topbuf.ctypes.data + 16, u8p))
            assert np.array_equal(a, b), (tx, mode)


def test_vp9_scaled_mc_batch():
    """scaled prediction blocks: steps from 16x up- to 2x down-scaling of the reference, all filters, put / avg"""
    from ffmpeg_amd import vp9
    from test_oracle_vs_ref import vp9_smc_case
    torch = _torch()
    rng = np.random.default_rng(450)
    W, H, P = 640, 384, 8
    ss = 2 * W + 2 * P + 3                                   # the reference may be twice as large
    sd = W + 9
    ref = rng.integers(0, 256, (2 * H + 2 * P + 8, ss), dtype=np.uint8)
    ref[:60] = rng.choice(np.array([0, 255], np.uint8), (60, ss))
    dst = rng.integers(0, 256, (H, sd), dtype=np.uint8)
    want = dst.copy()
    O = ffi.oracle()
    recs = []
    for by in range(0, H, 64):
        for bx in range(0, W, 64):
            f, avg, w, h, mx, my, dx, dy = vp9_smc_case(rng)
            so = (2 * by + P) * ss + 2 * bx + P
            recs.append((by * sd + bx, so, w, h, f, mx, my, avg, dx, dy))
            O.ffo_vp9_smc(f, avg, C.cast(want.ctypes.data + by * sd + bx, u8p), sd, C.cast(ref.ctypes.data + so, u8p), ss, w, h, mx, my, dx, dy)
    n = len(recs)
    rec = np.array(recs, vp9.SMC_DTYPE)
    d_dst = torch.from_numpy(dst.copy()).cuda()
    vp9.scaled_mc_batch(d_dst, sd, torch.from_numpy(ref).cuda(), ss, torch.from_numpy(rec.view(np.uint8).reshape(n, 16).copy()).cuda(), n)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    assert (want != dst).sum() > 1000
    assert np.array_equal(got, want), "%d mismatches, first %s" % ((got != want).sum(), np.argwhere(got != want)[:3])


def test_vp9_scaled_mc_host_faces():
    from ffmpeg_amd import vp9
    from test_oracle_vs_ref import vp9_smc_case
    _torch()
    c = vp9.smc_init(8)
    O = ffi.oracle()
    rng = np.random.default_rng(451)
    src = rng.integers(0, 256, (160, 160), dtype=np.uint8)
    for rep in range(24):
        f, avg, w, h, mx, my, dx, dy = vp9_smc_case(rng)
        idx = {64: 0, 32: 1, 16: 2, 8: 3, 4: 4}[w]
        sp = src.ctypes.data + 6 * 160 + 7
        d0 = rng.integers(0, 256, (64, 72), dtype=np.uint8)
        a, b = d0.copy(), d0.copy()
        c.smc[idx][f][avg](a.ctypes.data, 72, sp, 160, h, mx, my, dx, dy)
        O.ffo_vp9_smc(f, avg, ptr(b), 72, C.cast(sp, u8p), 160, w, h, mx, my, dx, dy)
        assert np.array_equal(a, b), (f, avg, w, h, mx, my, dx, dy)
