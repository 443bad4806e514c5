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
