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
