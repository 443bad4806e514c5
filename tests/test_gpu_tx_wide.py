"""-m gpu: av_tx's double and int32 FFT / MDCT through libffhip's C ABI (kernels/tx_wide.hip) against oracle/ffo_tx_wide.c — pinned
bit for bit to the reference on the CPU tier — and against the reference's own outputs in tests/golden/tx_wide.npz.  Every
comparison is on the bytes: integers and doubles alike."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_oracle_vs_ref_tx_wide import GOLD, TYPES, make_input, oracle_run, same_bits  # noqa: E402


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _ctx(kind, inv, len_, scale):
    from ffmpeg_amd import tx
    return tx.TxContext(TYPES[kind], inv, len_, scale)


def test_golden_vectors_on_the_gpu():
    torch = _torch()
    d = np.load(GOLD)
    for key in sorted(k[:-3] for k in d.files if k.endswith("_in")):
        kind, inv, len_ = key.rsplit("_", 2)
        x, want, scale = d[key + "_in"], d[key + "_out"], float(d[key + "_scale"][0])
        ctx = _ctx(kind, int(inv), int(len_), scale)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros(want.shape, dtype=d_in.dtype, device="cuda:0")
        ctx.batch(d_out, d_in)
        torch.cuda.synchronize()
        assert same_bits(d_out.cpu().numpy(), want), key
        # the av_tx_fn-shaped single transform on host pointers
        out1 = np.zeros_like(want[0])
        ctx.fn(out1, x[0].copy(), x.dtype.itemsize * (1 if kind.endswith("mdct") else 2))
        assert same_bits(out1, want[0]), key
        ctx.close()


@pytest.mark.parametrize("inv", [0, 1])
@pytest.mark.parametrize("kind,len_,nt", [("d_fft", 4, 1000), ("d_fft", 8, 37), ("d_fft", 16, 1), ("d_fft", 128, 513), ("d_fft", 1024, 300),
                                          ("d_fft", 2048, 17), ("d_fft", 8192, 9), ("i_fft", 4, 1000), ("i_fft", 16, 3), ("i_fft", 32, 515),
                                          ("i_fft", 512, 700), ("i_fft", 1024, 64), ("i_fft", 4096, 33), ("i_fft", 16384, 5),
                                          ("d_mdct", 16, 1000), ("d_mdct", 64, 5), ("d_mdct", 1024, 300), ("d_mdct", 2048, 130),
                                          ("d_mdct", 4096, 31), ("d_mdct", 16384, 7), ("i_mdct", 16, 1000), ("i_mdct", 128, 77),
                                          ("i_mdct", 256, 301), ("i_mdct", 512, 600), ("i_mdct", 1024, 2000), ("i_mdct", 2048, 64),
                                          ("i_mdct", 8192, 11), ("i_mdct", 32768, 3)])
def test_batches_against_the_oracle(kind, len_, nt, inv):
    torch = _torch()
    is_int, mdct = kind[0] == "i", kind.endswith("mdct")
    rng = np.random.default_rng(len_ * 8 + inv)
    scale = 1.0 if not mdct else (-0.37 if len_ == 1024 else 1.0 / len_ if inv else 1.0 / 64)
    x = np.stack([make_input(kind, inv, len_, rng, full_range=t % 3 == 2) for t in range(nt)]).astype(np.int32 if is_int else np.float64)
    n_out = len_ if mdct else 2 * len_
    pad = 4 if (len_ >= 64 and nt > 1) else 0          # a row pitch wider than the row
    ctx = _ctx(kind, inv, len_, scale)
    d_in = torch.zeros((nt, x.shape[1] + pad), dtype=torch.from_numpy(x).dtype, device="cuda:0")
    d_in[:, :x.shape[1]] = torch.from_numpy(x).cuda()
    d_out = torch.zeros((nt, n_out + pad), dtype=d_in.dtype, device="cuda:0")
    ctx.batch(d_out[:, :n_out], d_in[:, :x.shape[1]])
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert not got[:, n_out:].any()
    check = range(nt) if nt <= 64 else sorted(set(rng.integers(0, nt, 40).tolist()) | {0, nt - 1})
    for t in check:
        want = oracle_run(kind, inv, len_, np.float32(scale) if is_int else scale, x[t])
        assert same_bits(got[t, :n_out], want), (kind, len_, inv, t)
    ctx.close()


def test_mdct_int32_large_batch_properties():
    """a batch the oracle does not walk whole (8,192 AAC-shaped frames, forward then inverse int32 MDCT-1024): sampled rows of both
    stages equal the oracle's, and the forward transform is additive up to its roundings — mdct(a) + mdct(b) against mdct(a + b) differ
    by the accumulated half-LSB errors of the fold and of ten levels of CMUL (a random walk over 512 terms: tens of LSB, bounded here
    at 256 on coefficients of magnitude 2^25)"""
    torch = _torch()
    len_, nt = 1024, 8192
    rng = np.random.default_rng(5)
    x = rng.integers(-2 ** 22, 2 ** 22, (nt, 2 * len_), dtype=np.int64).astype(np.int32)
    f = _ctx("i_mdct", 0, len_, 1.0)
    b = _ctx("i_mdct", 1, len_, 1.0 / 32)
    d_x = torch.from_numpy(x).cuda()
    d_c = torch.zeros((nt, len_), dtype=torch.int32, device="cuda:0")
    d_y = torch.zeros((nt, len_), dtype=torch.int32, device="cuda:0")
    f.batch(d_c, d_x)
    b.batch(d_y, d_c)
    torch.cuda.synchronize()
    c, y = d_c.cpu().numpy(), d_y.cpu().numpy()
    for t in (0, nt // 2 - 1, nt - 1):
        assert same_bits(c[t], oracle_run("i_mdct", 0, len_, np.float32(1.0), x[t]))
        assert same_bits(y[t], oracle_run("i_mdct", 1, len_, np.float32(1.0 / 32), c[t]))
    s = (x[0:128:2].astype(np.int64) + x[1:128:2]).astype(np.int32)
    d_s = torch.zeros((64, len_), dtype=torch.int32, device="cuda:0")
    f.batch(d_s, torch.from_numpy(s).cuda())
    torch.cuda.synchronize()
    err = np.abs(d_s.cpu().numpy().astype(np.int64) - (c[0:128:2].astype(np.int64) + c[1:128:2]))
    assert 0 < err.max() <= 256, err.max()
    f.close(); b.close()


def test_refusals():
    _torch()
    from ffmpeg_amd import tx, _lib
    for type_, len_ in ((tx.DOUBLE_FFT, 24), (tx.DOUBLE_FFT, 16384), (tx.INT32_FFT, 32768), (tx.INT32_MDCT, 960), (tx.DOUBLE_MDCT, 8)):
        with pytest.raises(Exception):
            tx.TxContext(type_, 0, len_, 1.0)
    for type_ in (7, 8, 10, 11, 13, 14, 16, 17):       # the RDFT / DCT / DCT-I / DST-I forms of the wide types (float DCT-I / DST-I: test_gpu_tx_dcst1.py)
        with pytest.raises(Exception):
            tx.TxContext(type_, 0, 64, 1.0)
    with pytest.raises(Exception):
        tx.TxContext(tx.INT32_MDCT, 1, 64, 1.0, flags=tx.FULL_IMDCT)
