"""av_tx's DCT-I / DST-I (AV_TX_FLOAT_DCT_I / AV_TX_FLOAT_DST_I, libavutil/tx_template.c:2006-2075) on the CPU tier: oracle/ffo_tx.c's
ffo_dcst1_run against the reference compiled in place (when /root/reference's build is present) and against the committed vectors of
tests/golden/tx_dcst1.npz (always).  Float transforms: every output within 2^-18 of the transform's largest one (the reference's own
sub-FFT of n -+ 1 points sums in float in an order of its own)."""
import os

import numpy as np
import pytest

import ffi

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tx_dcst1.npz")
TOL = 2.0 ** -18


def oracle_run(typ, n, scale, x, stride=1):
    O = ffi.oracle()
    xi = np.zeros(n * stride, np.float32)
    xi[::stride] = x
    out = np.zeros(n, np.float32)
    O.ffo_dcst1_run(int(typ == 15), n, float(scale), ffi.ptr(out, ffi.f32p), ffi.ptr(xi, ffi.f32p), 4 * stride)
    return out


def close(got, want):
    return np.abs(got.astype(np.float64) - want).max() <= TOL * np.abs(want).max()


def test_golden_vectors():
    d = np.load(GOLD)
    keys = sorted(k[:-3] for k in d.files if k.endswith("_in"))
    assert len(keys) == 10
    for key in keys:
        typ, n = int(key[1:3]), int(key.split("_")[1])
        x, want, scale = d[key + "_in"], d[key + "_out"], float(d[key + "_scale"][0])
        for t in range(x.shape[0]):
            assert close(oracle_run(typ, n, scale, x[t]), want[t]), (key, t)


def test_textbook_sums_at_unit_scale():
    """at *scale == 1 the transforms are the textbook ones (the middle pair's quirk needs another scale)"""
    rng = np.random.default_rng(3)
    for n in (4, 6, 64, 100):
        x = rng.standard_normal(n)
        k = np.arange(n)[:, None]
        j = np.arange(n)[None, :]
        dct = x[0] + (-1.0) ** np.arange(n) * x[-1] + 2 * (x[None, 1:-1] * np.cos(np.pi * k * j[:, 1:-1] / (n - 1))).sum(1)
        dst = 2 * (x[None, :] * np.sin(np.pi * (k + 1) * (j + 1) / (n + 1))).sum(1)
        assert close(oracle_run(12, n, 1.0, x.astype(np.float32)), dct)
        assert close(oracle_run(15, n, 1.0, x.astype(np.float32)), dst)


def test_strided_input():
    rng = np.random.default_rng(4)
    x = rng.standard_normal(64).astype(np.float32)
    for typ in (12, 15):
        assert np.array_equal(oracle_run(typ, 64, 0.5, x), oracle_run(typ, 64, 0.5, x, stride=3))


needs_ref = pytest.mark.skipif(not ffi.have_ref(), reason="oracle/_ref/libffref.so not built (no /root/reference here)")


@needs_ref
@pytest.mark.parametrize("typ", [12, 15])
@pytest.mark.parametrize("n", [4, 6, 8, 10, 16, 30, 62, 64, 66, 100, 128, 254, 256, 258, 512, 1000, 1024])
def test_oracle_vs_reference(typ, n):
    R = ffi.ref()
    rng = np.random.default_rng(n * 31 + typ)
    for scale in (1.0, 1.0 / 64, 0.75, -1.5, 2.0 / (n + 1)):
        rc = R.ffref_tx_create(typ, 0, n, scale, 0)
        assert rc
        for mag in (1e-3, 1.0, 1e3):
            x = (rng.standard_normal(n) * mag).astype(np.float32)
            xi = np.zeros(2 * n + 8, np.float32)
            xi[:n] = x
            o = np.zeros(2 * n + 8, np.float32)
            R.ffref_tx_run(rc, ffi.ptr(o, ffi.f32p), ffi.ptr(xi, ffi.f32p), 4)
            assert close(oracle_run(typ, n, scale, x), o[:n].astype(np.float64)), (typ, n, scale, mag)
        R.ffref_tx_free(rc)


@needs_ref
def test_reference_refuses_odd_lengths():
    R = ffi.ref()
    for typ in (12, 15):
        for n in (5, 63, 65):
            assert not R.ffref_tx_create(typ, 0, n, 1.0, 0)
