"""GPU parity of AACDecDSP's stereo tools and long-term prediction through the C ABI vs the oracle (pinned to
the reference's members in tests/test_aac_tools_cpu.py): M/S + intensity stereo of hundreds of channel pairs in one launch;
a run of frames of several channels through apply_ltp -> imdct_and_windowing -> update_ltp with the state carried on the device.
Bit-identical."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, f32p, i32p, u8p
import aac_gen as A
import test_golden as G
from test_oracle_vs_ref import aac_tns_case, aac_tns_filters, AAC_SCALES

pytestmark = pytest.mark.gpu
u16p = C.POINTER(C.c_uint16)
i8p = C.POINTER(C.c_int8)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _dev(torch, rec, width):
    return torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(len(rec), width).copy()).cuda()


def test_aac_stereo_tools_batch():
    import torch
    from ffmpeg_amd import aac
    O = ffi.oracle()
    rng = np.random.default_rng(3200)
    npairs = 400
    fr = np.stack([A.spectrum(rng) for _ in range(2 * npairs)])
    want = fr.copy()
    recs = []
    for p in range(npairs):
        c = A.cpe(rng, p % 3 == 0)
        f0, f1 = 2 * p, 2 * p + 1
        O.ffo_aac_apply_mid_side_stereo(ptr(want[f0], f32p), ptr(want[f1], f32p), c["num_window_groups"], ptr(c["group_len"], u8p), c["max_sfb"],
                                        ptr(c["ms_mask"], u8p), ptr(c["band_type0"], i32p), ptr(c["band_type1"], i32p), ptr(c["swb"], u16p))
        O.ffo_aac_apply_intensity_stereo(ptr(want[f0], f32p), ptr(want[f1], f32p), c["num_window_groups"], ptr(c["group_len"], u8p), c["max_sfb"],
                                         c["ms_present"], ptr(c["ms_mask"], u8p), ptr(c["band_type1"], i32p), ptr(c["sf1"], f32p), ptr(c["swb"], u16p))
        recs.append(aac.ms_bands(f0, f1, c["num_window_groups"], c["group_len"], c["max_sfb"], c["ms_mask"], c["band_type0"], c["band_type1"], c["swb"]))
        recs.append(aac.is_bands(f0, f1, c["num_window_groups"], c["group_len"], c["max_sfb"], c["ms_present"], c["ms_mask"], c["band_type1"],
                                 c["sf1"], c["swb"]))
    rec = np.concatenate(recs)
    rec = rec[rng.permutation(len(rec))]                    # a pair's ranges are disjoint: any order
    assert len(rec) > 3000
    d = torch.from_numpy(fr.copy()).cuda()
    aac.band_ops_batch(d, d, _dev(torch, rec, 20), len(rec))
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    assert (bits(want) != bits(fr)).sum() > 100000
    assert np.array_equal(bits(got), bits(want)), np.argwhere((bits(got) != bits(want)).any(axis=1))[:5].ravel()


@pytest.mark.parametrize("nch", [1, 5, 40])
def test_aac_ltp_chain(nch):
    import torch
    from ffmpeg_amd import aac
    O = ffi.oracle()
    rng = np.random.default_rng(3210 + nch)
    windows = G.aac_golden_windows(G.load("aac"))
    wp = (f32p * 4)(*[ptr(w, f32p) for w in windows])
    ctx = aac.AacImdct(windows)
    ctx.ltp_init()
    m1024, m128 = O.ffo_mdct_create(1, 1024, AAC_SCALES[0]), O.ffo_mdct_create(1, 128, AAC_SCALES[1])
    mltp = O.ffo_mdct_create(0, 1024, np.float32(aac.SCALE_LTP))
    nframes = 7
    state = (rng.standard_normal((nch, 3072)) * 2000).astype(np.float32)
    saved = (rng.standard_normal((nch, 512)) * 0.1).astype(np.float32)
    d_state, d_saved = torch.from_numpy(state.copy()).cuda(), torch.from_numpy(saved.copy()).cuda()
    d_pred = torch.zeros((nch, 1024), dtype=torch.float32, device="cuda:0")
    d_out = torch.zeros((1, nch, 1024), dtype=torch.float32, device="cuda:0")
    prev_seq, prev_kb = np.zeros(nch, np.uint8), rng.integers(0, 2, nch).astype(np.uint8)
    npred = 0
    for f in range(nframes):
        coeffs = np.stack([A.spectrum(rng) for _ in range(nch)]) * np.float32(30)
        seq = rng.choice([0, 1, 2, 3], nch, p=[.5, .15, .2, .15]).astype(np.uint8)
        kb = rng.integers(0, 2, nch).astype(np.uint8)
        want = coeffs.copy()
        lrec, tns_rec, add_rec = np.zeros(nch, aac.LTP_DTYPE), [], []
        n = 0
        for c in range(nch):
            if rng.random() < .25 or seq[c] == A.EIGHT_SHORT:          # no LTP on this frame (the member returns at once on short windows)
                continue
            l = A.ltp(rng, int(seq[c]))
            s2, k2 = np.array([seq[c], prev_seq[c]], np.int32), np.array([kb[c], prev_kb[c]], np.int32)
            t = aac_tns_case(rng, 0)
            t["swb"], t["num_swb"], t["max_sfb"] = l["swb"], l["num_swb"], l["max_sfb"]
            has_tns = rng.random() < .6
            orec = aac_tns_filters(O, t) if has_tns else []
            pf = np.zeros(1024, np.float32)
            O.ffo_aac_apply_ltp(mltp, wp, ptr(want[c], f32p), ptr(state[c], f32p), l["lag"], l["coef"], ptr(l["used"], i8p), ptr(s2, i32p),
                                ptr(k2, i32p), l["max_sfb"], ptr(l["swb"], u16p), orec.ctypes.data if len(orec) else None, len(orec), ptr(pf, f32p))
            lrec[n] = (c, l["lag"], seq[c], int(kb[c]) | int(prev_kb[c]) << 1, l["coef"], 0)
            if has_tns:
                tns_rec.append(aac.tns_filters(n, t["n_filt"], t["length"], t["direction"], t["order"], t["coef"], 1, t["num_swb"], t["swb"],
                                               t["tns_max_bands"], t["max_sfb"]))
            add_rec.append(aac.ltp_bands(c, n, l["max_sfb"], l["used"], l["swb"]))
            n += 1
        npred += n
        d_co = torch.from_numpy(coeffs.copy()).cuda()
        if n:
            ctx.ltp_predict(d_state, d_pred, _dev(torch, lrec[:n], 16), n)
            tr = np.concatenate(tns_rec) if tns_rec else []
            if len(tr):
                aac.apply_tns_batch(d_pred, _dev(torch, tr, 92), len(tr), decode=0)
            ar = np.concatenate(add_rec)
            if len(ar):
                aac.band_ops_batch(d_co, d_pred, _dev(torch, ar, 20), len(ar))
        torch.cuda.synchronize()
        assert np.array_equal(bits(d_co.cpu().numpy()), bits(want)), (f, np.argwhere((bits(d_co.cpu().numpy()) != bits(want)).any(axis=1)).ravel()[:5])
        # the frame's samples, then the state for the next frame
        wout = np.zeros((nch, 1024), np.float32)
        for c in range(nch):
            s2, k2 = np.array([seq[c], prev_seq[c]], np.int32), np.array([kb[c], prev_kb[c]], np.int32)
            buf = np.zeros(1024, np.float32)
            if seq[c] == A.EIGHT_SHORT:
                for w in range(8):
                    O.ffo_mdct_run(m128, C.cast(buf.ctypes.data + 512 * w, f32p), C.cast(want[c].ctypes.data + 512 * w, f32p), 4)
            else:
                O.ffo_mdct_run(m1024, ptr(buf, f32p), ptr(want[c], f32p), 4)
            O.ffo_aac_imdct_and_windowing(m1024, m128, wp, ptr(want[c], f32p), ptr(s2, i32p), ptr(k2, i32p), ptr(saved[c], f32p), ptr(wout[c], f32p))
            O.ffo_aac_update_ltp(wp, ptr(state[c], f32p), ptr(buf, f32p), ptr(saved[c], f32p), ptr(wout[c], f32p), int(seq[c]), int(kb[c]))
        ctx.batch(d_co.reshape(1, nch, 1024), d_out, d_saved, seq.reshape(1, nch), kb.reshape(1, nch), prev_seq, prev_kb)
        ctx.update_ltp(d_state, d_out, nch)
        torch.cuda.synchronize()
        assert np.array_equal(bits(d_out.cpu().numpy().reshape(nch, 1024)), bits(wout)), f
        assert np.array_equal(bits(d_saved.cpu().numpy()), bits(saved)), f
        assert np.array_equal(bits(d_state.cpu().numpy()), bits(state)), (f, np.argwhere((bits(d_state.cpu().numpy()) != bits(state)).any(axis=1)).ravel()[:5])
        prev_seq, prev_kb = seq, kb
    assert npred >= nframes * nch // 3
    for m in (m1024, m128, mltp):
        O.ffo_mdct_free(m)
    ctx.close()


def test_aac_ltp_rejects():
    import torch
    from ffmpeg_amd import aac
    ctx = aac.AacImdct(G.aac_golden_windows(G.load("aac")))
    z = torch.zeros((1, 3072), dtype=torch.float32, device="cuda:0")
    with pytest.raises(RuntimeError, match="ltp_init"):
        ctx.ltp_predict(z, z, torch.zeros((1, 16), dtype=torch.uint8, device="cuda:0"), 1)
    with pytest.raises(RuntimeError, match="follows"):
        ctx.update_ltp(z, z, 1)
    ctx.close()


@pytest.mark.parametrize("nch", [1, 7, 64])
def test_aac_apply_prediction_chain(nch):
    """AAC Main's predictors on the device: runs of frames of several channels, the state carried in device memory, against the
    oracle (pinned to the reference's member on the CPU) — group resets, short-window resets, uninitialised first frames"""
    import torch
    from ffmpeg_amd import aac
    O = ffi.oracle()
    rng = np.random.default_rng(3300 + nch)
    state = rng.standard_normal((nch, 672, 8)).astype(np.float32)                 # garbage until a channel's first frame
    d_state = torch.from_numpy(state.copy()).cuda()
    init = np.zeros(nch, np.int32)
    base = (rng.standard_normal((nch, 1024)) * 300).astype(np.float32)           # a tonal signal: the predictors lock on
    changed = 0
    for f in range(12):
        coeffs = (base * (1 + .05 * rng.standard_normal((nch, 1024)))).astype(np.float32)
        want = coeffs.copy()
        recs = []
        for c in range(nch):
            if f > 0 and rng.random() < .2:
                continue                                                            # a channel with no element in this frame
            is_long = int(rng.random() < .85)
            present = int(rng.random() < .7)
            used = rng.integers(0, 2, 41).astype(np.uint8)
            reset_group = int(rng.integers(1, 31)) if rng.random() < .3 else 0
            recs.append(aac.prediction_record(c, c, is_long, int(init[c]), present, used, 40, A.SWB_1024, reset_group))
            ini = np.array([init[c]], np.int32)
            st = np.ascontiguousarray(state[c]).reshape(-1)
            O.ffo_aac_apply_prediction(ptr(st, f32p), ptr(want[c], f32p), is_long, ptr(ini, i32p), present, ptr(used, u8p), 40,
                                       ptr(A.SWB_1024, u16p), reset_group)
            state[c] = st.reshape(672, 8)
            init[c] = ini[0]
        if not recs:
            continue
        rec = np.concatenate(recs)
        d_co = torch.from_numpy(coeffs.copy()).cuda()
        aac.apply_prediction_batch(d_state, d_co, _dev(torch, rec, 100), len(rec))
        torch.cuda.synchronize()
        assert np.array_equal(bits(d_co.cpu().numpy()), bits(want)), f
        changed += int((bits(want) != bits(coeffs)).sum())
        got = d_state.cpu().numpy()
        touched = np.array([int(r["channel"][0]) for r in recs])
        assert np.array_equal(bits(got[touched][:, :, :6]), bits(state[touched][:, :, :6])), f
    assert changed > 100 * nch


def test_aac_coupling_batch():
    import torch
    from ffmpeg_amd import aac
    O = ffi.oracle()
    rng = np.random.default_rng(3310)
    nf = 200
    fr = np.stack([A.spectrum(rng) for _ in range(2 * nf)])
    want = fr.copy()
    recs = []
    for p in range(nf):
        c = A.ics(rng, p % 3 == 0)
        n = c["num_window_groups"] * c["max_sfb"]
        if n > 120:
            c["max_sfb"] = 120 // c["num_window_groups"]
            n = c["num_window_groups"] * c["max_sfb"]
        bt = np.zeros(128, np.int32)
        bt[:n] = rng.integers(0, 3, n)
        gain = (2.0 ** (rng.integers(-20, 20, 120) / 8.0)).astype(np.float32)
        O.ffo_aac_apply_dependent_coupling(ptr(want[2 * p], f32p), ptr(want[2 * p + 1], f32p), c["num_window_groups"], ptr(c["group_len"], u8p),
                                           c["max_sfb"], ptr(bt, i32p), ptr(gain, f32p), ptr(c["swb"], u16p))
        recs.append(aac.coupling_bands(2 * p, 2 * p + 1, c["num_window_groups"], c["group_len"], c["max_sfb"], bt, gain, c["swb"]))
    rec = np.concatenate(recs)
    d = torch.from_numpy(fr.copy()).cuda()
    aac.band_ops_batch(d, d, _dev(torch, rec, 20), len(rec))
    torch.cuda.synchronize()
    assert np.array_equal(bits(d.cpu().numpy()), bits(want))
    assert (bits(want) != bits(fr)).sum() > 10000
    # apply_independent_coupling: one FMAC record over the output samples of a frame (2048 with SBR)
    a, b = (rng.standard_normal((2, 2048)) * 100).astype(np.float32)
    w = a.copy()
    O.ffo_aac_apply_independent_coupling(ptr(w, f32p), ptr(b, f32p), 0.7071, 2048)
    one = np.zeros(1, aac.BAND_OP_DTYPE)
    one[0] = (0, 1, 0, 2048, 0.7071, aac.BAND_FMAC, (0, 0, 0))
    d2 = torch.from_numpy(np.stack([a[:1024], b[:1024]]).copy()).cuda()       # rows of 1024: dest = rows 0.., src = rows 1..: use a flat pair
    flat = torch.from_numpy(np.concatenate([a, b]).copy()).cuda()
    one[0]["frame1"] = 2                                                        # src starts 2 * 1024 floats in
    aac.band_ops_batch(flat, flat, _dev(torch, one, 20), 1)
    torch.cuda.synchronize()
    assert np.array_equal(bits(flat.cpu().numpy()[:2048]), bits(w))
    del d2
