"""GPU parity: AACDecDSP.imdct_and_windowing (float AAC decoder, 1024-sample frames) vs the oracle and the reference's stored
outputs, through the C ABI.  The window tables are the decoder's own, taken from tests/golden/aac.npz (written from the reference
built in place) - the hip path receives them from its caller and never computes them."""
import numpy as np
import pytest

import ffi
import test_golden as G
from test_oracle_vs_ref import aac_sequences, aac_oracle_run, aac_tns_case, aac_tns_filters

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_aac_golden_gpu():
    """the reference's frames: batch face (all frames in one call) and the host face frame by frame"""
    from ffmpeg_amd import aac
    torch = _torch()
    d = G.load("aac")
    ctx = aac.AacImdct(G.aac_golden_windows(d))
    nf = len(d["seq"])
    d_out = torch.zeros((nf, 1, 1024), dtype=torch.float32, device="cuda:0")
    d_saved = torch.from_numpy(d["saved_in"].copy()).cuda().reshape(1, 512)
    ctx.batch(torch.from_numpy(np.ascontiguousarray(d["coeffs"])).cuda().reshape(nf, 1, 1024), d_out, d_saved, d["seq"], d["kb"],
              d["prev"][:1], d["prev"][1:])
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().reshape(nf, 1024).view(np.uint32), d["out"].view(np.uint32))
    assert np.array_equal(d_saved.cpu().numpy().reshape(512).view(np.uint32), d["saved_out"].view(np.uint32))
    saved = d["saved_in"].copy()
    prev = tuple(int(v) for v in d["prev"])
    for f in range(nf):
        out = np.zeros(1024, np.float32)
        ctx.frame(np.ascontiguousarray(d["coeffs"][f]), (int(d["seq"][f]), prev[0]), (int(d["kb"][f]), prev[1]), saved, out)
        assert np.array_equal(out.view(np.uint32), d["out"][f].view(np.uint32)), f
        prev = (int(d["seq"][f]), int(d["kb"][f]))
    assert np.array_equal(saved.view(np.uint32), d["saved_out"].view(np.uint32))
    ctx.close()


@pytest.mark.parametrize("nch,nframes", [(1, 1), (2, 300), (6, 97)])
def test_aac_imdct_and_windowing_batch(nch, nframes):
    """channels x frames in one call against the oracle run channel by channel, frame after frame; then a second call continues
    from the state the first one left (the overlap state and the previous window sequence / shape carry over)"""
    from ffmpeg_amd import aac
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(2710 + nch)
    d = G.load("aac")
    windows = G.aac_golden_windows(d)
    ctx = aac.AacImdct(windows)
    total = nframes + 40
    seq = np.zeros((total, nch), np.uint8); kb = np.zeros((total, nch), np.uint8)
    for c in range(nch):
        s, k = aac_sequences(rng, total)
        seq[:, c], kb[:, c] = s, k
    coeffs = (rng.standard_normal((total, nch, 1024)) * 3000.0 * 10.0 ** rng.integers(-2, 2, (total, nch, 1))).astype(np.float32)
    saved0 = (rng.standard_normal((nch, 512)) * 0.1).astype(np.float32)
    want = np.zeros_like(coeffs)
    wsaved = saved0.copy()
    for c in range(nch):
        sv = wsaved[c].copy()
        want[:, c] = aac_oracle_run(O, windows, np.ascontiguousarray(coeffs[:, c]), seq[:, c], kb[:, c], sv)   # previous of frame 0: (ONLY_LONG, kb[0])
        wsaved[c] = sv
    d_co = torch.from_numpy(coeffs).cuda()
    d_out = torch.zeros((total, nch, 1024), dtype=torch.float32, device="cuda:0")
    d_saved = torch.from_numpy(saved0.copy()).cuda()
    ctx.batch(d_co[:nframes], d_out[:nframes], d_saved, seq[:nframes], kb[:nframes], np.zeros(nch, np.uint8), kb[0])
    if total > nframes:
        ctx.batch(d_co[nframes:], d_out[nframes:], d_saved, seq[nframes:], kb[nframes:], seq[nframes - 1], kb[nframes - 1])
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "frames differing: %s" % np.argwhere((got != want).any(axis=2))[:5]
    assert np.array_equal(d_saved.cpu().numpy().view(np.uint32), wsaved.view(np.uint32))
    assert np.array_equal(d_co.cpu().numpy(), coeffs)          # coefficients are left alone
    ctx.close()


def test_aac_rejects_bad_sequence():
    from ffmpeg_amd import aac
    torch = _torch()
    ctx = aac.AacImdct(G.aac_golden_windows(G.load("aac")))
    z = torch.zeros((1, 1, 1024), dtype=torch.float32, device="cuda:0")
    with pytest.raises(RuntimeError, match="window sequence"):
        ctx.batch(z, z.clone(), torch.zeros((1, 512), dtype=torch.float32, device="cuda:0"), np.array([4], np.uint8), np.array([0], np.uint8),
                  np.array([0], np.uint8), np.array([0], np.uint8))
    ctx.close()


@pytest.mark.parametrize("decode", [1, 0])
def test_aac_apply_tns_batch(decode):
    """thousands of channel-frames' TNS filters in one launch (long and short windows, both directions, orders 1..20, clipped
    ranges) against the oracle filter by filter; the host-side range walk against the oracle's; untouched coefficients stay"""
    from ffmpeg_amd import aac
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(2730 + decode)
    nframes = 1500
    coeffs = (rng.standard_normal((nframes, 1024)) * 10.0 ** rng.integers(-2, 4, (nframes, 1))).astype(np.float32)
    want = coeffs.copy()
    recs = []
    for f in range(nframes):
        c = aac_tns_case(rng, f & 1)
        mine = aac.tns_filters(f, c["n_filt"], c["length"], c["direction"], c["order"], c["coef"], c["num_windows"], c["num_swb"], c["swb"],
                               c["tns_max_bands"], c["max_sfb"])
        ora = aac_tns_filters(O, c)
        assert len(mine) == len(ora)
        for a, b in zip(mine, ora):
            assert (a["start"], a["size"], a["inc"], a["order"]) == (b["start"], b["size"], b["inc"], b["order"])
            assert np.array_equal(a["coef"][:a["order"]], b["coef"][:b["order"]]) and a["frame"] == f
        recs.append(mine)
        row = np.ascontiguousarray(want[f])
        for r in ora:
            O.ffo_aac_tns_run(row.ctypes.data_as(ffi.f32p), np.array(r).ctypes.data, decode)
        want[f] = row
    rec = np.concatenate(recs)
    rec = rec[rng.permutation(len(rec))]                       # the order of the records is free
    assert len(rec) > 1500
    d_co = torch.from_numpy(coeffs.copy()).cuda()
    aac.apply_tns_batch(d_co, torch.from_numpy(rec.view(np.uint8).reshape(len(rec), 92).copy()).cuda(), len(rec), decode)
    torch.cuda.synchronize()
    got = d_co.cpu().numpy()
    assert (want != coeffs).sum() > 100000
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "frames differing: %s" % np.argwhere((got != want).any(axis=1))[:5].ravel()


@pytest.mark.parametrize("L", [960, 768])
@pytest.mark.parametrize("nch,nframes", [(1, 1), (2, 150), (5, 64)])
def test_aac_imdct_and_windowing_960_768(L, nch, nframes):
    """AACDecDSP.imdct_and_windowing_960 / _768 (frame lengths of DAB+ / DRM and of USAC's 768 mode): the batch face against the
    oracle's length-generic restatement (pinned to the reference's members in tests/test_aac_tools_cpu.py) channel by channel, frame
    after frame, a second call continuing from the first one's state; then the host face.  The transforms are the prime-factor ones
    (15 x 32 / 15 x 4 resp. 3 x 128 / 3 x 16)."""
    from ffmpeg_amd import aac
    import ctypes as C
    from ffi import ptr, f32p, i32p
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(2750 + L + nch)
    S = L // 8
    windows = [np.zeros(L, np.float32), np.zeros(S, np.float32), np.zeros(L, np.float32), np.zeros(S, np.float32)]
    O.ffo_aac_sine_window(ptr(windows[0], f32p), L); O.ffo_aac_sine_window(ptr(windows[1], f32p), S)
    O.ffo_aac_kbd_window(ptr(windows[2], f32p), 4.0, L); O.ffo_aac_kbd_window(ptr(windows[3], f32p), 6.0, S)
    wp = (f32p * 4)(*[ptr(w, f32p) for w in windows])
    ml, ms = O.ffo_mdct_create(1, L, np.float32((1.0 / L) / 32768.0)), O.ffo_mdct_create(1, S, np.float32((1.0 / S) / 32768.0))
    ctx = aac.AacImdct(windows, frame_len=L)
    total = nframes + 23
    seq = np.zeros((total, nch), np.uint8); kb = np.zeros((total, nch), np.uint8)
    for c in range(nch):
        seq[:, c], kb[:, c] = aac_sequences(rng, total)
    coeffs = (rng.standard_normal((total, nch, 1024)) * 3000.0 * 10.0 ** rng.integers(-2, 2, (total, nch, 1))).astype(np.float32)
    saved0 = np.zeros((nch, 512), np.float32)
    saved0[:, :L // 2] = (rng.standard_normal((nch, L // 2)) * 0.1).astype(np.float32)
    want = np.zeros((total, nch, 1024), np.float32)
    wsaved = saved0.copy()
    for c in range(nch):
        prev = (0, int(kb[0, c]))
        for f in range(total):
            s2, k2 = np.array([seq[f, c], prev[0]], np.int32), np.array([kb[f, c], prev[1]], np.int32)
            O.ffo_aac_imdct_and_windowing_len(L, 128 if L == 960 else 96, ml, ms, wp, ptr(np.ascontiguousarray(coeffs[f, c]), f32p),
                                              ptr(s2, i32p), ptr(k2, i32p), ptr(wsaved[c], f32p), ptr(want[f, c], f32p))
            prev = (int(seq[f, c]), int(kb[f, c]))
    d_co = torch.from_numpy(coeffs).cuda()
    d_out = torch.zeros((total, nch, 1024), dtype=torch.float32, device="cuda:0")
    d_saved = torch.from_numpy(saved0.copy()).cuda()
    ctx.batch(d_co[:nframes], d_out[:nframes], d_saved, seq[:nframes], kb[:nframes], np.zeros(nch, np.uint8), kb[0])
    ctx.batch(d_co[nframes:], d_out[nframes:], d_saved, seq[nframes:], kb[nframes:], seq[nframes - 1], kb[nframes - 1])
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "frames differing: %s" % np.argwhere((got != want).any(axis=2))[:5]
    assert not got[:, :, L:].any()
    assert np.array_equal(d_saved.cpu().numpy().view(np.uint32), wsaved.view(np.uint32))
    # host face, one channel
    sv = saved0[0, :L // 2].copy()
    prev = (0, int(kb[0, 0]))
    for f in range(min(total, 12)):
        out = np.zeros(L, np.float32)
        ctx.frame(np.ascontiguousarray(coeffs[f, 0]), (int(seq[f, 0]), prev[0]), (int(kb[f, 0]), prev[1]), sv, out)
        assert np.array_equal(out.view(np.uint32), want[f, 0, :L].view(np.uint32)), f
        prev = (int(seq[f, 0]), int(kb[f, 0]))
    O.ffo_mdct_free(ml); O.ffo_mdct_free(ms)
    ctx.close()


@pytest.mark.parametrize("kind", ["ld", "eld512", "eld480"])
@pytest.mark.parametrize("nch,nframes", [(1, 1), (2, 2), (3, 90)])
def test_aac_ld_eld_batch(kind, nch, nframes):
    """AACDecDSP.imdct_and_windowing_ld / _eld on device-resident frames against the oracle (pinned to the reference's members in
    tests/test_aac_tools_cpu.py) frame after frame, a second call continuing from the state the first one left — the ELD history
    is three frames deep, so runs shorter than that mix the caller's history with the batch's own"""
    from ffmpeg_amd import aac
    from ffi import ptr, f32p
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(2760 + nch + nframes)
    eld = kind != "ld"
    n = 480 if kind == "eld480" else 512
    if eld:
        win = [np.sin(np.linspace(0.01, 3.0, n * 15 // 4)).astype(np.float32) * np.float32(0.9)]      # any table: the caller hands it over
        depth = 3 * n
    else:
        win = [np.zeros(512, np.float32), np.zeros(128, np.float32)]
        O.ffo_aac_sine_window(ptr(win[0], f32p), 512); O.ffo_aac_sine_window(ptr(win[1], f32p), 128)
        depth = 256
    m = O.ffo_mdct_create(1, n, np.float32((1.0 / n) / 32768.0))
    ctx = aac.AacLd(win, eld=eld, frame_len=n)
    total = nframes + 5
    coeffs = (rng.standard_normal((total, nch, 1024)) * 3000.0).astype(np.float32)
    kbp = rng.integers(0, 2, (total, nch)).astype(np.uint8)
    saved0 = (rng.standard_normal((nch, depth)) * 0.1).astype(np.float32)
    want = np.zeros((total, nch, 1024), np.float32)
    wsaved = saved0.copy()
    for c in range(nch):
        for f in range(total):
            co = np.ascontiguousarray(coeffs[f, c])
            if eld:
                O.ffo_aac_imdct_and_windowing_eld(n, m, ptr(win[0], f32p), ptr(co, f32p), ptr(wsaved[c], f32p), ptr(want[f, c], f32p))
            else:
                O.ffo_aac_imdct_and_windowing_ld(m, ptr(win[0], f32p), ptr(win[1], f32p), ptr(co, f32p), int(kbp[f, c]), ptr(wsaved[c], f32p),
                                                 ptr(want[f, c], f32p))
    d_co = torch.from_numpy(coeffs).cuda()
    d_out = torch.zeros((total, nch, 1024), dtype=torch.float32, device="cuda:0")
    d_saved = torch.from_numpy(saved0.copy()).cuda()
    ctx.batch(d_co[:nframes], d_out[:nframes], d_saved, None if eld else kbp[:nframes])
    ctx.batch(d_co[nframes:], d_out[nframes:], d_saved, None if eld else kbp[nframes:])
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "frames differing: %s" % np.argwhere((got != want).any(axis=2))[:5]
    assert np.array_equal(d_saved.cpu().numpy().view(np.uint32), wsaved.view(np.uint32))
    assert np.array_equal(d_co.cpu().numpy(), coeffs)
    O.ffo_mdct_free(m)
    ctx.close()
