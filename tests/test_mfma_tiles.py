"""CPU check of the host side of the matrix-core horizontal pass: the per-tile MFMA operand records
(ffhip_sws_mfma_tiles_host) reproduce hScale8To15_c when the i8 matrix product is emulated in numpy —
D = (src ^ 0x80 as int8) . (256*B_hi + B_lo) + bias, per 32-sample tile over its 32-byte window."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, i16p, i32p
from ffmpeg_amd import _lib, swscale as S

REC = 2320


def tiles(f, p, n, srcW, pair, swap):
    L = _lib.lib()
    f = np.ascontiguousarray(f, np.int16); p = np.ascontiguousarray(p, np.int32)
    nt = L.ffhip_sws_mfma_tiles_host(f.ctypes.data, p.ctypes.data, n, srcW, pair, swap, None, 0)
    if nt < 0:
        return nt, None
    buf = np.zeros(nt * REC, np.uint8)
    assert L.ffhip_sws_mfma_tiles_host(f.ctypes.data, p.ctypes.data, n, srcW, pair, swap, buf.ctypes.data, buf.size) == nt
    return nt, buf.reshape(nt, REC)


def emulate(rec, row_bytes):
    """one tile, one source row -> the 32 horizontal sums (before >> 7)"""
    bhi = rec[:1024].view(np.int8).reshape(64, 16).astype(np.int64)
    blo = rec[1024:2048].view(np.int8).reshape(64, 16).astype(np.int64)
    bias = rec[2048:2304].view(np.int32).astype(np.int64)
    kb = int(rec[2304:2308].view(np.int32)[0])
    a = (row_bytes[kb:kb + 32].astype(np.int64) - 128)           # src ^ 0x80 read as int8
    out = np.zeros(32, np.int64)
    for l in range(64):
        j, g = l & 31, l >> 5
        out[j] += (a[16 * g:16 * g + 16] * (256 * bhi[l] + blo[l])).sum()
    assert np.array_equal(bias[:32], bias[32:])
    return out + bias[:32]


@pytest.mark.parametrize("sw,dw", [(1920, 3840), (960, 1920), (64, 192), (200, 520), (96, 128)])
def test_single_plane_tiles(sw, dw):
    ht = S.HostTables(sw, 64, 0, dw, 128, 0, S.SWS_BICUBIC)
    f, p, fs, n = ht.bank("hLum")
    assert fs == 4 and n == dw
    nt, recs = tiles(f, p, n, sw, 0, 0)
    assert nt == (dw + 31) // 32
    rng = np.random.default_rng(sw)
    row = rng.integers(0, 256, sw, dtype=np.uint8)
    row[::7] = 255; row[3::11] = 0
    want = (row[p[:, None] + np.arange(4)[None, :]].astype(np.int64) * f.reshape(n, 4)).sum(1)
    got = np.concatenate([emulate(recs[t], row) for t in range(nt)])[:n]
    assert np.array_equal(got, want)
    # and the int16 the reference stores
    ref = np.zeros(n, np.int16)
    ffi.oracle().ffo_hscale8to15(ptr(ref, i16p), n, ptr(row), ptr(np.ascontiguousarray(f), i16p), ptr(np.ascontiguousarray(p), i32p), 4)
    assert np.array_equal(np.minimum(got >> 7, 32767).astype(np.int16), ref)


@pytest.mark.parametrize("swap", [0, 1])
@pytest.mark.parametrize("sw,dw", [(1920, 3840), (128, 256), (96, 288)])
def test_interleaved_pair_tiles(sw, dw, swap):
    fmt = 24 if swap else 23
    ht = S.HostTables(sw, 64, fmt, dw, 128, fmt, S.SWS_BICUBIC)
    f, p, fs, n = ht.bank("hChr")
    csw = sw // 2
    assert fs == 4 and n == dw // 2
    nt, recs = tiles(f, p, n, csw, 1, swap)
    assert nt == (n + 15) // 16
    rng = np.random.default_rng(dw + swap)
    uv = rng.integers(0, 256, 2 * csw, dtype=np.uint8)
    u, v = (uv[1::2], uv[0::2]) if swap else (uv[0::2], uv[1::2])
    fw = f.reshape(n, 4).astype(np.int64)
    want_u = (u[p[:, None] + np.arange(4)[None, :]].astype(np.int64) * fw).sum(1)
    want_v = (v[p[:, None] + np.arange(4)[None, :]].astype(np.int64) * fw).sum(1)
    got = np.stack([emulate(recs[t], uv) for t in range(nt)])
    assert np.array_equal(got[:, :16].reshape(-1)[:n], want_u)
    assert np.array_equal(got[:, 16:].reshape(-1)[:n], want_v)


def test_ineligible_banks_are_refused():
    ht = S.HostTables(1920, 1080, 23, 640, 360, 23, S.SWS_BICUBIC)          # down-scaling: 11 taps, wide footprint
    f, p, fs, n = ht.bank("hLum")
    if fs == 4:
        assert tiles(f, p, n, 1920, 0, 0)[0] < 0
    f = np.full((64, 4), 32767, np.int16)                                   # high byte 128 does not fit int8
    p = (np.arange(64) // 2).astype(np.int32)
    assert tiles(f.reshape(-1), p, 64, 64, 0, 0)[0] < 0
