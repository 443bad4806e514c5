"""av_tx's double and int32 FFT / MDCT (AV_TX_DOUBLE_FFT / _MDCT, AV_TX_INT32_FFT / _MDCT; libavutil/tx_double.c, tx_int32.c) on the
CPU tier: oracle/ffo_tx_wide.c against the reference compiled in place (bit for bit, when /root/reference's build is present) and
against the committed vectors of tests/golden/tx_wide.npz (always)."""
import os

import numpy as np
import pytest

import ffi

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tx_wide.npz")
TYPES = {"d_fft": 2, "d_mdct": 3, "i_fft": 4, "i_mdct": 5}


def oracle_run(kind, inv, len_, scale, x):
    O = ffi.oracle()
    is_int, mdct = kind[0] == "i", kind.endswith("mdct")
    out = np.zeros(len_ if mdct else 2 * len_, x.dtype)
    xi = np.ascontiguousarray(x)
    if mdct:
        O.ffo_txw_mdct_run(int(is_int), inv, len_, float(scale), out.ctypes.data, xi.ctypes.data)
    else:
        O.ffo_txw_fft_run(int(is_int), inv, len_, out.ctypes.data, xi.ctypes.data)
    return out


def same_bits(a, b):
    return np.array_equal(a.view(np.uint8), b.view(np.uint8))


def make_input(kind, inv, len_, rng, full_range=False):
    n_in = 2 * len_ if (kind.endswith("fft") or not inv) else len_
    if kind[0] == "d":
        return rng.standard_normal(n_in) * 10.0 ** float(rng.integers(-3, 4))
    hi = 2 ** 31 - 1 if (kind.endswith("fft") or full_range) else 2 ** 24
    return rng.integers(-hi, hi, n_in, dtype=np.int64).astype(np.int32)


def test_golden_vectors():
    d = np.load(GOLD)
    keys = sorted(k[:-3] for k in d.files if k.endswith("_in"))
    assert len(keys) == 44
    for key in keys:
        kind, inv, len_ = key.rsplit("_", 2)
        x, want, scale = d[key + "_in"], d[key + "_out"], float(d[key + "_scale"][0])
        for t in range(x.shape[0]):
            got = oracle_run(kind, int(inv), int(len_), scale, x[t])
            assert same_bits(got, want[t]), key


@pytest.mark.skipif(not ffi.have_ref(), reason="the reference build (oracle/_ref) is absent")
@pytest.mark.parametrize("inv", [0, 1])
@pytest.mark.parametrize("kind,len_", [(k, n) for k in ("d_fft", "i_fft") for n in (4, 8, 16, 32, 64, 128, 512, 2048, 8192, 16384)] +
                         [(k, n) for k in ("d_mdct", "i_mdct") for n in (16, 32, 64, 128, 256, 1024, 4096, 16384, 32768)])
def test_oracle_is_the_reference(kind, len_, inv):
    R = ffi.ref()
    if not hasattr(R, "ffref_tx_create_d"):
        pytest.skip("oracle/_ref predates ffref_tx_create_d")
    is_int, mdct = kind[0] == "i", kind.endswith("mdct")
    rng = np.random.default_rng(len_ * 4 + inv * 2 + is_int)
    for scale in ([1.0] if not mdct else [1.0 / len_ if inv else 1.0, -0.37, 1.0 / 64]):
        rc = R.ffref_tx_create(TYPES[kind], inv, len_, scale, 0) if is_int else R.ffref_tx_create_d(TYPES[kind], inv, len_, scale, 0)
        assert rc
        for rep in range(3):
            x = make_input(kind, inv, len_, rng, full_range=rep == 2)
            want = np.zeros(len_ if mdct else 2 * len_, x.dtype)
            xi = x.copy()
            R.ffref_tx_run(rc, want.ctypes.data, xi.ctypes.data, x.dtype.itemsize * (1 if mdct else 2))
            # the scale an int32 context sees is the caller's float
            got = oracle_run(kind, inv, len_, np.float32(scale) if is_int else scale, x)
            assert same_bits(got, want), (kind, len_, inv, scale, rep)
        R.ffref_tx_free(rc)
