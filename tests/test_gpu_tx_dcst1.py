"""-m gpu: av_tx's DCT-I / DST-I (AV_TX_FLOAT_DCT_I / AV_TX_FLOAT_DST_I, forward) through libffhip's C ABI (kernels/tx_dcst1.hip)
against oracle/ffo_tx.c's ffo_dcst1_run — pinned to the reference on the CPU tier — and against the reference's own outputs in
tests/golden/tx_dcst1.npz.  Tolerance: every output within 2^-18 of its transform's largest one, the bound of the other float
transforms; the two middle outputs at *scale != 1 are the C code's, not the textbook's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_oracle_vs_ref_tx_dcst1 import GOLD, TOL, oracle_run  # noqa: E402


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _close_rows(got, want):
    tol = TOL * np.abs(want).max(axis=1, keepdims=True)
    return (np.abs(got.astype(np.float64) - want) <= tol).all()


def test_golden_vectors_on_the_gpu():
    torch = _torch()
    from ffmpeg_amd import tx
    d = np.load(GOLD)
    for key in sorted(k[:-3] for k in d.files if k.endswith("_in")):
        typ, n = int(key[1:3]), int(key.split("_")[1])
        x, want, scale = d[key + "_in"], d[key + "_out"], float(d[key + "_scale"][0])
        ctx = tx.TxContext(typ, 0, n, scale)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros(want.shape, dtype=torch.float32, device="cuda:0")
        ctx.batch(d_out, d_in)
        torch.cuda.synchronize()
        assert _close_rows(d_out.cpu().numpy(), want.astype(np.float64)), key
        # the av_tx_fn-shaped single transform on host pointers, inputs three floats apart
        xs = np.zeros(3 * n, np.float32)
        xs[::3] = x[0]
        out1 = np.zeros(n, np.float32)
        ctx.fn(out1, xs, 12)
        assert _close_rows(out1[None], want[:1].astype(np.float64)), key
        ctx.close()


@pytest.mark.parametrize("typ", [tuple((12,)), tuple((15,))])
@pytest.mark.parametrize("n,nt", [(4, 1000), (6, 3), (30, 129), (64, 4097), (64, 1), (66, 50), (100, 333), (128, 64), (254, 17), (256, 40),
                                  (258, 9), (512, 33), (1000, 5), (1024, 24)])
def test_batches_against_the_oracle(typ, n, nt):
    torch = _torch()
    from ffmpeg_amd import tx
    typ = typ[0]
    rng = np.random.default_rng(n * 3 + typ)
    scale = (1.0 / 64, 1.0, 0.75, -1.5)[(n // 2 + typ) % 4]
    x = (rng.standard_normal((nt, n)) * 10.0 ** rng.integers(-3, 4, (nt, 1))).astype(np.float32)
    ctx = tx.TxContext(typ, 0, n, scale)
    pad = 3 if nt > 1 else 0                                   # a row pitch wider than the row
    d_in = torch.zeros((nt, n + pad), dtype=torch.float32, device="cuda:0")
    d_in[:, :n] = torch.from_numpy(x).cuda()
    d_out = torch.full((nt, n + pad), 777.0, dtype=torch.float32, device="cuda:0")
    ctx.batch(d_out[:, :n], d_in[:, :n])
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert (got[:, n:] == 777.0).all()
    rows = range(nt) if nt <= 64 else list(rng.choice(nt, 48, replace=False)) + [0, nt - 1]
    for t in rows:
        want = oracle_run(typ, n, scale, x[t]).astype(np.float64)
        assert np.abs(got[t, :n] - want).max() <= TOL * np.abs(want).max(), (typ, n, t)
    ctx.close()


def test_strided_batch_inputs():
    """the input side carries av_tx_fn's stride (ff_tx_dctI reads src[i * stride], tx_template.c:2047)"""
    torch = _torch()
    from ffmpeg_amd import tx
    rng = np.random.default_rng(9)
    x = rng.standard_normal((40, 64)).astype(np.float32)
    for typ in (12, 15):
        ctx = tx.TxContext(typ, 0, 64, 1.0 / 64)
        wide = torch.zeros((40, 128), dtype=torch.float32, device="cuda:0")
        wide[:, ::2] = torch.from_numpy(x).cuda()
        a = torch.zeros((40, 64), dtype=torch.float32, device="cuda:0")
        b = torch.zeros((40, 64), dtype=torch.float32, device="cuda:0")
        ctx.batch(a, torch.from_numpy(x).cuda())
        ctx.batch(b, wide, stride=8)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        ctx.close()


def test_a_transform_is_its_own_inverse_up_to_the_scale():
    """DCT-I twice = 2 (n - 1) x, DST-I twice = 2 (n + 1) x at *scale 1 (libavutil/tx.h:110-112,122-124): a property at a size the
    oracle's naive sums do not reach comfortably"""
    torch = _torch()
    from ffmpeg_amd import tx
    rng = np.random.default_rng(10)
    for typ, n in ((12, 64), (15, 64), (12, 1024), (15, 1024)):
        x = torch.from_numpy(rng.standard_normal((20000 if n == 64 else 300, n)).astype(np.float32)).cuda()
        ctx = tx.TxContext(typ, 0, n, 1.0)
        y, z = torch.zeros_like(x), torch.zeros_like(x)
        ctx.batch(y, x)
        ctx.batch(z, y)
        torch.cuda.synchronize()
        k = 2.0 * (n - 1 if typ == 12 else n + 1)
        err = (z / k - x).abs().max().item()
        assert err <= 2.0 ** -16 * x.abs().max().item(), (typ, n, err)
        ctx.close()


def test_refusals_by_name():
    _torch()
    from ffmpeg_amd import tx, _lib
    for typ, inv, n in ((12, 1, 64), (15, 1, 64), (12, 0, 2), (15, 0, 63), (12, 0, 2048)):
        with pytest.raises(Exception) as e:
            tx.TxContext(typ, inv, n, 1.0)
        assert "DCT-I" in str(e.value), str(e.value)
