"""GPU parity: the AVFloatDSPContext vector operations vs the oracle, bit-identical floats."""
import numpy as np
import pytest

import ffi
from ffi import ptr, f32p
from test_oracle_vs_ref import fdsp_operands

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.mark.parametrize("n", [1024, 1000, 16, 37, 4, 1])
@pytest.mark.parametrize("op", range(7))
def test_fdsp_batch(op, n):
    """a batch of vectors with a shared third operand (the window), vector and scalar paths (n % 4, alignment)"""
    from ffmpeg_amd import fdsp
    torch = _torch()
    rng = np.random.default_rng(op * 100 + n)
    nvec = 9
    rows = [fdsp_operands(rng, op, n) for _ in range(nvec)]
    mul = rows[0][4]
    dst = np.stack([r[0] for r in rows]); s0 = np.stack([r[1] for r in rows]); s1 = np.stack([r[2] for r in rows])
    s2 = rows[0][3]                                             # shared across the batch (pitch 0)
    want_d, want_0 = dst.copy(), s0.copy()
    O = ffi.oracle()
    for v in range(nvec):
        O.ffo_fdsp(op, ptr(want_d[v], f32p), ptr(want_0[v], f32p), ptr(s1[v], f32p), ptr(s2, f32p), mul, n)
    d_d, d_0, d_1, d_2 = [torch.from_numpy(a.copy()).cuda() for a in (dst, s0, s1, s2)]
    fdsp.batch(op, d_d, d_0, d_1, d_2, mul, n)
    torch.cuda.synchronize()
    assert np.array_equal(d_d.cpu().numpy().view(np.uint32), want_d.view(np.uint32)), "dst"
    assert np.array_equal(d_0.cpu().numpy().view(np.uint32), want_0.view(np.uint32)), "src0 (written by butterflies only)"


def test_fdsp_host_faces():
    from ffmpeg_amd import fdsp
    _torch()
    c = fdsp.dsp_init()
    O = ffi.oracle()
    rng = np.random.default_rng(8)
    for n in (256, 1000, 5):
        for op in range(7):
            dst, s0, s1, s2, mul = fdsp_operands(rng, op, n)
            a, a0 = dst.copy(), s0.copy()
            b, b0 = dst.copy(), s0.copy()
            O.ffo_fdsp(op, ptr(b, f32p), ptr(b0, f32p), ptr(s1, f32p), ptr(s2, f32p), mul, n)
            p = lambda x: x.ctypes.data
            if op == 0: c.vector_fmul(p(a), p(a0), p(s1), n)
            elif op == 1: c.vector_fmac_scalar(p(a), p(a0), mul, n)
            elif op == 2: c.vector_fmul_scalar(p(a), p(a0), mul, n)
            elif op == 3: c.vector_fmul_window(p(a), p(a0), p(s1), p(s2), n)
            elif op == 4: c.vector_fmul_add(p(a), p(a0), p(s1), p(s2), n)
            elif op == 5: c.vector_fmul_reverse(p(a), p(a0), p(s1), n)
            else: c.butterflies_float(p(a), p(a0), n)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (op, n)
            assert np.array_equal(a0.view(np.uint32), b0.view(np.uint32)), (op, n)
