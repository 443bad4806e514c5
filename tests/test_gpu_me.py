"""GPU parity: me_cmp SAD/SATD and the exhaustive search vs the oracle, exact integers."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p, i16p

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("width", [16, 8])
def test_me_cmp_batch(kind, width):
    """tests/checkasm/motion.c:37-92 shape: 64x64 random images, random positions, h in {8,16} (SAD: 4..16 even)"""
    from ffmpeg_amd import me
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(kind * 2 + width)
    W = 64
    a = rng.integers(0, 256, (W, W), dtype=np.uint8)
    b = rng.integers(0, 256, (W, W), dtype=np.uint8)
    b[:32] = np.clip(a[:32].astype(int) + rng.integers(-3, 4, (32, W)), 0, 255)
    n = 500
    hs = [8, 16] if kind else [4, 6, 8, 10, 12, 14, 16]
    for h in hs:
        o1 = (rng.integers(0, W - 16, n) * W + rng.integers(0, (W - 16) // 16 + 1, n) * 16).astype(np.int32)
        o2 = (rng.integers(0, W - 16, n) * W + rng.integers(0, W - 16, n)).astype(np.int32)
        want = np.zeros(n, np.int32)
        for i in range(n):
            pa, pb = C.cast(a.ctypes.data + int(o1[i]), u8p), C.cast(b.ctypes.data + int(o2[i]), u8p)
            if kind == 0:
                want[i] = O.ffo_sad(width, pa, pb, W, h)
            elif width == 16:
                want[i] = O.ffo_hadamard8_diff16(pa, pb, W, h)
            else:
                want[i] = O.ffo_hadamard8_diff8x8(pa, pb, W)
        out = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        me.cmp_batch(kind, width, h, torch.from_numpy(a).cuda(), torch.from_numpy(o1).cuda(), torch.from_numpy(b).cuda(),
                     torch.from_numpy(o2).cuda(), W, out)
        assert np.array_equal(out.cpu().numpy(), want), "h=%d" % h


@pytest.mark.parametrize("width", [16, 8])
@pytest.mark.parametrize("kind", [2, 3, 4, 5, 6], ids=["x2", "y2", "xy2", "sse", "nsse"])
def test_me_cmp_halfpel_sse_nsse_batch(kind, width):
    """pix_abs*_x2 / _y2 / _xy2, sse, nsse (me_cmp.c:53-104,184-440) over many positions: == the oracle (pinned to the reference)"""
    torch = _torch()
    from ffmpeg_amd import _lib
    L, O = _lib.lib(), ffi.oracle()
    O.ffo_me_cmp_other.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int]
    rng = np.random.default_rng(kind * 3 + width)
    W = 80
    a = rng.integers(0, 256, (W, W), dtype=np.uint8)
    b = rng.integers(0, 256, (W, W), dtype=np.uint8)
    b[:40] = np.clip(a[:40].astype(int) + rng.integers(-6, 7, (40, W)), 0, 255)
    n = 400
    for h in (4, 8, 16):
        o1 = (rng.integers(0, W - 18, n) * W + rng.integers(0, W - 18, n)).astype(np.int32)
        o2 = (rng.integers(0, W - 18, n) * W + rng.integers(0, W - 18, n)).astype(np.int32)
        want = np.array([O.ffo_me_cmp_other(kind, width, C.cast(a.ctypes.data + int(o1[i]), u8p), C.cast(b.ctypes.data + int(o2[i]), u8p), W, h)
                         for i in range(n)], np.int32)
        out = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        da, db, d1, d2 = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(o1).cuda(), torch.from_numpy(o2).cuda()
        assert L.ffhip_me_cmp_batch_dev(kind, width, h, da.data_ptr(), d1.data_ptr(), db.data_ptr(), d2.data_ptr(), W, out.data_ptr(), n, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want), (kind, width, h)


def _shifted_pair(rng, w, h, stride, R, flat=False):
    big = rng.integers(0, 256, (h + 64, w + 64), dtype=np.uint8)
    if flat:
        big[:] = 128
        big[::5, ::7] = 130
    dx, dy = rng.integers(-R, R + 1, 2)
    ref = np.zeros((h, stride), np.uint8)
    cur = np.zeros((h, stride), np.uint8)
    ref[:, :w] = big[32:32 + h, 32:32 + w]
    cur[:, :w] = np.clip(big[32 + dy:32 + dy + h, 32 + dx:32 + dx + w].astype(int) + rng.integers(-2, 3, (h, w)), 0, 255)
    return cur, ref, int(dx), int(dy)


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("w,h,pad,mb,R", [(64, 48, 0, 16, 7), (176, 144, 16, 16, 7), (176, 144, 0, 8, 7), (96, 80, 3, 16, 16),
                                          (40, 40, 0, 8, 3), (72, 56, 0, 16, 0), (128, 96, 0, 16, 24), (160, 128, 5, 16, 32),
                                          (64, 64, 0, 8, 20), (52, 36, 1, 16, 5)])
@pytest.mark.parametrize("share", ["default", "1", "0", "3"], ids=["product", "shared", "per-candidate", "column"])
def test_esa_frames(kind, w, h, pad, mb, R, share, monkeypatch):
    from ffmpeg_amd import me
    torch = _torch()
    if share != "default":   # "default": no knob -> the product library; the variants run libffhip_measure.so (conftest.py)
        monkeypatch.setenv("FFHIP_ME_SATD_SHARE", "0" if share == "0" else "1")   # SATD: column transforms shared through LDS / per candidate
        monkeypatch.setenv("FFHIP_ME_SAD_QUAD", share)     # SAD 16x16: four candidates per lane (side by side) / one / four in a column
    rng = np.random.default_rng(w + h + mb + R + kind)
    stride = w + pad
    nf = 3
    curs, refs = [], []
    for f in range(nf):
        c, r, _, _ = _shifted_pair(rng, w, h, stride, max(R, 1), flat=(f == 2))
        if f == 1:
            c[:mb * 2, :mb * 2] = r[:mb * 2, :mb * 2]       # exact matches -> zero-cost early-out of the reference
        curs.append(c); refs.append(r)
    cur, ref = np.stack(curs), np.stack(refs)
    bw, bh = w // mb, h // mb
    wmv = np.zeros((nf, bh * bw * 2), np.int16)
    wcost = np.zeros((nf, bh * bw), np.uint32)
    for f in range(nf):
        ffi.oracle().ffo_me_esa_frame(ptr(cur[f]), ptr(ref[f]), stride, w, h, mb, R, kind, ptr(wmv[f], i16p),
                                      wcost[f].ctypes.data_as(C.POINTER(C.c_uint32)))
    d_mv = torch.zeros((nf, bh * bw * 2), dtype=torch.int16, device="cuda:0")
    d_cost = torch.zeros((nf, bh * bw), dtype=torch.int32, device="cuda:0")
    me.esa_batch(torch.from_numpy(cur).cuda(), torch.from_numpy(ref).cuda(), w, h, stride, h * stride, nf, mb, R, kind, d_mv,
                 d_cost)
    torch.cuda.synchronize()
    assert np.array_equal(d_cost.cpu().numpy().view(np.uint32), wcost)
    assert np.array_equal(d_mv.cpu().numpy(), wmv)


@pytest.mark.parametrize("kind,R", [(0, 7), (0, 16), (1, 7)])
def test_esa_4k_known_motion(kind, R):
    """3840x2160, frame t = frame t-1 shifted by a known (dx,dy) plus +-2 LSB noise:
    interior MBs must find exactly that shift; a sample of MBs is checked against the oracle's search."""
    from ffmpeg_amd import me
    torch = _torch()
    rng = np.random.default_rng(R + kind)
    w, h, mb = 3840, 2160, 16
    cur, ref, dx, dy = _shifted_pair(rng, w, h, w, R)
    bw, bh = w // mb, h // mb
    d_mv = torch.zeros((bh * bw * 2,), dtype=torch.int16, device="cuda:0")
    d_cost = torch.zeros((bh * bw,), dtype=torch.int32, device="cuda:0")
    me.esa_batch(torch.from_numpy(cur).cuda(), torch.from_numpy(ref).cuda(), w, h, w, h * w, 1, mb, R, kind, d_mv, d_cost)
    torch.cuda.synchronize()
    mv = d_mv.cpu().numpy().reshape(bh, bw, 2)
    cost = d_cost.cpu().numpy().view(np.uint32).reshape(bh, bw)
    gx, gy = np.meshgrid(np.arange(bw) * mb, np.arange(bh) * mb)
    inner = (slice(2, bh - 2), slice(2, bw - 2))
    assert (mv[..., 0][inner] - gx[inner] == dx).all() and (mv[..., 1][inner] - gy[inner] == dy).all()
    for by, bx in zip(rng.integers(0, bh, 60), rng.integers(0, bw, 60)):
        m = (C.c_int * 2)(int(bx) * mb, int(by) * mb)
        c = ffi.oracle().ffo_me_search_esa(ptr(cur), ptr(ref), w, w, h, mb, R, kind, int(bx) * mb, int(by) * mb, m)
        assert (int(mv[by, bx, 0]), int(mv[by, bx, 1]), int(cost[by, bx])) == (m[0], m[1], int(c)), (bx, by)
