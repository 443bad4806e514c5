"""GPU parity of vp9dsp above 8 bits (profiles 2 / 3): the *_hbd batch faces vs the oracle's *_bd functions (pinned to the
reference's 10- and 12-bit instantiations in tests/test_oracle_vs_ref_hbd.py), bit-exact.  Samples are uint16, itxfm_add's
coefficients int32; strides and record offsets in bytes."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p, i32p
from test_gpu_hevc_hbd import at, back, dev, pix
from test_oracle_vs_ref_hbd import vp9_block32, vp9_lf_plane16

pytestmark = pytest.mark.gpu
DEPTHS = [10, 12]


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("bd", DEPTHS)
@pytest.mark.parametrize("tx", range(5))
def test_vp9_itxfm_batch_hbd(tx, bd):
    from ffmpeg_amd import vp9
    torch = _torch()
    rng = np.random.default_rng(400 + tx + bd)
    n = 4 if tx == 4 else 4 << tx
    gx, gy = 24, 10
    nb = gx * gy
    stride = gx * n + 5                                 # samples
    dst = pix(rng, (gy * n, stride), bd, extremes=True)
    want = dst.copy()
    coeffs = np.zeros((nb + 1, n * n), np.int32)
    wcoef = coeffs.copy()
    rec = np.zeros(nb, vp9.TU_DTYPE)
    O = ffi.oracle()
    order = rng.permutation(nb)
    for j, b in enumerate(order):
        kind = int(rng.integers(0, 5))
        txtp = int(rng.integers(0, 4))
        blk = vp9_block32(rng, n, kind, bd)
        eob = 1 if kind == 2 else n * n
        by, bx = divmod(int(b), gx)
        coeffs[b] = blk
        off = 2 * (by * n * stride + bx * n)
        rec[j] = (b * n * n, off, txtp, int(eob == 1), (0, 0))
        wb = blk.copy()
        O.ffo_vp9_itxfm_add_bd(bd, tx, txtp, at(want, off), 2 * stride, ptr(wb, i32p), eob)
        wcoef[b] = wb
    coeffs[nb] = wcoef[nb] = 1234
    d_dst, d_co = dev(torch, dst), torch.from_numpy(coeffs.copy()).cuda()
    vp9.itxfm_add_batch(tx, d_co, d_dst, 2 * stride, torch.from_numpy(rec.view(np.uint8).reshape(nb, 12).copy()).cuda(), nb, bit_depth=bd)
    torch.cuda.synchronize()
    assert (want != dst).sum() > 1000
    assert np.array_equal(back(d_dst, dst), want)
    assert np.array_equal(d_co.cpu().numpy(), wcoef)


@pytest.mark.parametrize("bd", DEPTHS)
def test_vp9_mc_batch_hbd(bd):
    from ffmpeg_amd import vp9
    torch = _torch()
    rng = np.random.default_rng(420 + bd)
    W, H, P = 640, 512, 16
    ss, sd = W + 2 * P + 3, W + 9                        # samples
    ref = pix(rng, (H + 2 * P, ss), bd)
    ref[:60] = rng.choice(np.array([0, (1 << bd) - 1], np.uint16), (60, ss))
    dst = pix(rng, (H, sd), bd)
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
            so = 2 * ((by + P + dy) * ss + bx + P + dx)
            do = 2 * (by * sd + bx)
            recs.append((do, so, w, h, f, mx, my, avg, (0, 0)))
            O.ffo_vp9_mc_bd(bd, f, avg, at(want, do), 2 * sd, at(ref, so), 2 * ss, w, h, mx, my)
    n = len(recs)
    rec = np.array(recs, vp9.MC_DTYPE)
    d_dst = dev(torch, dst)
    vp9.mc_batch(d_dst, 2 * sd, dev(torch, ref), 2 * ss, torch.from_numpy(rec.view(np.uint8).reshape(n, 16).copy()).cuda(), n, bit_depth=bd)
    torch.cuda.synchronize()
    got = back(d_dst, dst)
    assert (want != dst).sum() > 1000
    assert np.array_equal(got, want), "%d mismatches, first %s" % ((got != want).sum(), np.argwhere(got != want)[:3])


@pytest.mark.parametrize("bd", DEPTHS)
def test_vp9_scaled_mc_batch_hbd(bd):
    from ffmpeg_amd import vp9
    from test_oracle_vs_ref import vp9_smc_case
    torch = _torch()
    rng = np.random.default_rng(450 + bd)
    W, H, P = 640, 384, 8
    ss, sd = 2 * W + 2 * P + 3, W + 9
    ref = pix(rng, (2 * H + 2 * P + 8, ss), bd)
    ref[:60] = rng.choice(np.array([0, (1 << bd) - 1], np.uint16), (60, ss))
    dst = pix(rng, (H, sd), bd)
    want = dst.copy()
    O = ffi.oracle()
    recs = []
    for by in range(0, H, 64):
        for bx in range(0, W, 64):
            f, avg, w, h, mx, my, dx, dy = vp9_smc_case(rng)
            so, do = 2 * ((2 * by + P) * ss + 2 * bx + P), 2 * (by * sd + bx)
            recs.append((do, so, w, h, f, mx, my, avg, dx, dy))
            O.ffo_vp9_smc_bd(bd, f, avg, at(want, do), 2 * sd, at(ref, so), 2 * ss, w, h, mx, my, dx, dy)
    n = len(recs)
    rec = np.array(recs, vp9.SMC_DTYPE)
    d_dst = dev(torch, dst)
    vp9.scaled_mc_batch(d_dst, 2 * sd, dev(torch, ref), 2 * ss, torch.from_numpy(rec.view(np.uint8).reshape(n, 16).copy()).cuda(), n, bit_depth=bd)
    torch.cuda.synchronize()
    got = back(d_dst, dst)
    assert (want != dst).sum() > 1000
    assert np.array_equal(got, want), "%d mismatches, first %s" % ((got != want).sum(), np.argwhere(got != want)[:3])


@pytest.mark.parametrize("bd", DEPTHS)
def test_vp9_loop_filter_batch_hbd(bd):
    from ffmpeg_amd import vp9
    torch = _torch()
    rng = np.random.default_rng(430 + bd)
    gy, gx = 14, 20
    plane = np.zeros((gy * 48, gx * 48 + 3), np.uint16)
    for ty in range(gy):
        for tx in range(gx):
            plane[ty * 48:(ty + 1) * 48, tx * 48:(tx + 1) * 48] = vp9_lf_plane16(rng, bd)
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
            for sgm in range(int(rng.integers(1, 3))):
                off = 2 * ((ty * 48 + 24) * stride + tx * 48 + 24 + 8 * sgm * (1 if d else stride))
                recs.append((off, w, d, E, I, H, (0, 0, 0)))
                O.ffo_vp9_loop_filter_bd(bd, WD[w], d, at(want, off), 2 * stride, E, I, H)
    n = len(recs)
    rec = np.array(recs, vp9.EDGE_DTYPE)
    d_pl = dev(torch, plane)
    vp9.loop_filter_batch(d_pl, 2 * stride, torch.from_numpy(rec.view(np.uint8).reshape(n, 12).copy()).cuda(), n, bit_depth=bd)
    torch.cuda.synchronize()
    got = back(d_pl, plane)
    assert (want != plane).sum() > 2000
    assert np.array_equal(got, want), "%d mismatches" % (got != want).sum()


@pytest.mark.parametrize("bd", DEPTHS)
@pytest.mark.parametrize("tx", range(4))
def test_vp9_intra_pred_batch_hbd(tx, bd):
    from ffmpeg_amd import vp9
    torch = _torch()
    rng = np.random.default_rng(440 + tx + bd)
    n = 4 << tx
    gx, gy = 30, 12
    nb = gx * gy
    stride = gx * n + (4 if tx & 1 else 5)
    mxv = (1 << bd) - 1
    dst = pix(rng, (gy * n, stride), bd)
    want = dst.copy()
    slot = n + 1 + max(n, 8)
    edges = rng.integers(0, mxv + 1, (nb, slot)).astype(np.uint16)
    edges[::4] = rng.choice(np.array([0, mxv], np.uint16), (len(edges[::4]), slot))
    rec = np.zeros(nb, vp9.INTRA_DTYPE)
    O = ffi.oracle()
    for b in range(nb):
        by, bx = divmod(b, gx)
        mode = b % 15
        off = 2 * (by * n * stride + bx * n)
        rec[b] = (off, 2 * b * slot, mode, (0, 0, 0))
        e = edges[b]
        left = np.ascontiguousarray(e[:n])
        topbuf = np.zeros(16 + 64, np.uint16)
        topbuf[15] = e[n]
        topbuf[16:16 + max(n, 8)] = e[n + 1:]
        O.ffo_vp9_intra_pred_bd(bd, tx, mode, at(want, off), 2 * stride, ptr(left), C.cast(topbuf.ctypes.data + 32, u8p))
    d_dst = dev(torch, dst)
    vp9.intra_pred_batch(tx, d_dst, 2 * stride, dev(torch, edges), torch.from_numpy(rec.view(np.uint8).reshape(nb, 12).copy()).cuda(), nb,
                         bit_depth=bd)
    torch.cuda.synchronize()
    got = back(d_dst, dst)
    assert np.array_equal(got, want), "%d mismatches, first %s" % ((got != want).sum(), np.argwhere(got != want)[:3])


@pytest.mark.parametrize("bd", DEPTHS)
def test_vp9_host_faces_hbd(bd):
    """ff_vp9dsp_*_init_hip(c, 10 / 12): the reference's signatures with host pointers on uint16 planes / int32 blocks"""
    from ffmpeg_amd import vp9
    from test_oracle_vs_ref import vp9_smc_case
    _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(700 + bd)
    mxv = (1 << bd) - 1
    # itxfm_add
    c = vp9.dsp_init(bd)
    for tx in range(5):
        n = 4 if tx == 4 else 4 << tx
        for txtp in range(4):
            for kind in (1, 2):
                blk = vp9_block32(rng, n, kind, bd)
                eob = 1 if kind == 2 else n * n
                dst0 = pix(rng, (n, n + 7), bd)
                a, b, ba, bb = dst0.copy(), dst0.copy(), blk.copy(), blk.copy()
                c.itxfm_add[tx][txtp](a.ctypes.data, 2 * (n + 7), ba.ctypes.data, eob)
                O.ffo_vp9_itxfm_add_bd(bd, tx, txtp, ptr(b), 2 * (n + 7), ptr(bb, i32p), eob)
                assert np.array_equal(a, b) and np.array_equal(ba, bb), (tx, txtp, kind)
    # mc
    c = vp9.mc_init(bd)
    src = pix(rng, (90, 100), bd, extremes=True)
    for rep in range(32):
        idx = rep % 5
        w = 64 >> idx
        f, avg = (rep // 5) % 4, rep & 1
        h = int(rng.choice([2, 8, 64]))
        mx, my = (int(v) for v in rng.integers(1, 16, 2))
        hx, vy = (rep >> 1) & 1, (rep >> 2) & 1
        sp = src.ctypes.data + 2 * (10 * 100 + 12)
        d0 = pix(rng, (64, 72), bd)
        a, b = d0.copy(), d0.copy()
        c.mc[idx][f][avg][hx][vy](a.ctypes.data, 144, sp, 200, h, mx, my)
        O.ffo_vp9_mc_bd(bd, f, avg, ptr(b), 144, C.cast(sp, u8p), 200, w, h, mx if hx else 0, my if vy else 0)
        assert np.array_equal(a, b), (idx, f, avg, hx, vy, h, mx, my)
    # loop filter
    c = vp9.lf_init(bd)
    WD = [4, 8, 16]
    for rep in range(36):
        pl = vp9_lf_plane16(rng, bd)
        a, b = pl.copy(), pl.copy()
        d = rep & 1
        E, I, H = (255, 63, int(rng.integers(0, 16))) if rep % 3 == 0 else (int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 16)))
        p0 = 2 * (24 * 48 + 24)
        seg2 = 2 * 8 * (1 if d else 48)
        which = rep % 3
        if which == 0:
            w = (rep // 3) % 3
            c.loop_filter_8[w][d](a.ctypes.data + p0, 96, E, I, H)
            O.ffo_vp9_loop_filter_bd(bd, WD[w], d, at(b, p0), 96, E, I, H)
        elif which == 1:
            c.loop_filter_16[d](a.ctypes.data + p0, 96, E, I, H)
            for sgm in range(2):
                O.ffo_vp9_loop_filter_bd(bd, 16, d, at(b, p0 + sgm * seg2), 96, E, I, H)
        else:
            w1, w2 = (rep // 3) & 1, (rep // 6) & 1
            E2, I2, H2 = int(rng.integers(0, 256)), int(rng.integers(0, 64)), int(rng.integers(0, 16))
            c.loop_filter_mix2[w1][w2][d](a.ctypes.data + p0, 96, E | E2 << 8, I | I2 << 8, H | H2 << 8)
            O.ffo_vp9_loop_filter_bd(bd, WD[w1], d, at(b, p0), 96, E, I, H)
            O.ffo_vp9_loop_filter_bd(bd, WD[w2], d, at(b, p0 + seg2), 96, E2, I2, H2)
        assert np.array_equal(a, b), (rep, which, d)
    # intra prediction
    c = vp9.intra_init(bd)
    for tx in range(4):
        n = 4 << tx
        for mode in range(15):
            left = rng.integers(0, mxv + 1, n).astype(np.uint16)
            topbuf = rng.integers(0, mxv + 1, 16 + 2 * n + 8).astype(np.uint16)
            a = pix(rng, (n, n + 3), bd)
            b = a.copy()
            c.intra_pred[tx][mode](a.ctypes.data, 2 * (n + 3), left.ctypes.data, topbuf.ctypes.data + 32)
            O.ffo_vp9_intra_pred_bd(bd, tx, mode, ptr(b), 2 * (n + 3), ptr(left), C.cast(topbuf.ctypes.data + 32, u8p))
            assert np.array_equal(a, b), (tx, mode)
    # scaled mc
    c = vp9.smc_init(bd)
    src = pix(rng, (160, 160), bd)
    for rep in range(24):
        f, avg, w, h, mx, my, dx, dy = vp9_smc_case(rng)
        idx = {64: 0, 32: 1, 16: 2, 8: 3, 4: 4}[w]
        sp = src.ctypes.data + 2 * (6 * 160 + 7)
        d0 = pix(rng, (64, 72), bd)
        a, b = d0.copy(), d0.copy()
        c.smc[idx][f][avg](a.ctypes.data, 144, sp, 320, h, mx, my, dx, dy)
        O.ffo_vp9_smc_bd(bd, f, avg, ptr(b), 144, C.cast(sp, u8p), 320, w, h, mx, my, dx, dy)
        assert np.array_equal(a, b), (f, avg, w, h, mx, my, dx, dy)
