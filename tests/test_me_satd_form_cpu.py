"""CPU tier: the form k_me_esa_satd_mx (me_satd.hip) computes is the reference's SATD.

hadamard8_diff8x8_c (libavcodec/me_cmp.c:514-562) runs a butterfly network over the rows and the columns of cur - ref and sums the
absolute values; the kernel multiplies by H8 (x) H8 in Sylvester order on the matrix cores, with the samples offset by 128 (so that
they are int8) and the current block's transform entering as the accumulator input.  The absolute sum does not see the order of the
coefficients, the transform is linear, and the offset cancels: restated in numpy and compared with the oracle (pinned to the reference).
"""
import ctypes as C

import numpy as np

import ffi
from ffi import u8p


def _h8():
    i = np.arange(8)
    par = np.array([[bin(a & b).count("1") & 1 for b in i] for a in i])
    return 1 - 2 * par          # Sylvester-Hadamard: H[a][b] = (-1)^popcount(a & b)


def test_dense_hadamard_product_is_hadamard8_diff():
    O = ffi.oracle()
    H = _h8()
    K = np.kron(H, H)            # row m = (u, v), column k = (y, x): H[u][y] * H[v][x] — the kernel's A operand
    assert set(np.unique(K)) == {-1, 1} and np.array_equal(K @ K.T, 64 * np.eye(64, dtype=int))
    rng = np.random.default_rng(3)
    W = 48
    for trial in range(200):
        a = rng.integers(0, 256, (W, W), dtype=np.uint8)
        b = rng.integers(0, 256, (W, W), dtype=np.uint8)
        if trial % 4 == 0:
            b = np.clip(a.astype(int) + rng.integers(-2, 3, (W, W)), 0, 255).astype(np.uint8)
        if trial % 50 == 0:
            a[:] = 255 * (trial % 100 == 0)
            b[:] = 255 - a                                          # the extremes: every coefficient's range
        ya, xa, yb, xb = rng.integers(0, W - 8, 4)
        pa = C.cast(a.ctypes.data + int(ya) * W + int(xa), u8p)
        pb = C.cast(b.ctypes.data + int(yb) * W + int(xb), u8p)
        want = O.ffo_hadamard8_diff8x8(pa, pb, W)
        cur = a[ya:ya + 8, xa:xa + 8].astype(np.int64) - 128       # what the kernel stages: samples - 128 as int8
        ref = b[yb:yb + 8, xb:xb + 8].astype(np.int64) - 128
        bias = 1 << 20
        acc_in = bias - K @ cur.reshape(64)                         # the accumulator input: BIAS - T(cur)
        d = K @ ref.reshape(64) + acc_in                            # the MFMA's result: BIAS + T(ref - cur)
        assert d.min() >= 0 and d.max() < (1 << 31)
        assert int(np.abs(d - bias).sum()) == want, trial           # v_sad_u32 against BIAS
