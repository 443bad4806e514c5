"""GPU: several pictures' intra reconstruction wavefronts in ONE launch (ffhip_h264_intra_frames_dev, round 4).

Each picture of the batch — its own planes, records and coefficient runs — must come out exactly as from a launch of its own
(ffhip_h264_intra_frame_dev[_hbd]), which tests/test_gpu_h264_picture.py pins to the reference's ff_h264_hl_decode_mb() through the
picture object; one picture is also checked against the oracle directly here.  Batches larger than one launch holds (32 pictures),
pictures with only some macroblocks intra (the rest untouched), 8 and 10 bits."""
import ctypes as C

import numpy as np
import pytest

import ffi
import h264_intra_gen as G

pytestmark = pytest.mark.gpu


class IntraPic(C.Structure):   # FFHipH264IntraPic (include/ffhip.h)
    _fields_ = [("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p), ("recs", C.c_void_p), ("row_start", C.c_void_p), ("coefs", C.c_void_p)]


def _pack_picture(L, rng, mb_w, mb_h, frac, depth):
    """records sorted by (mb_y, mb_x), row starts, packed coefficient runs of a picture with `frac` of its macroblocks intra"""
    recs, coefs, ncoef, states = [], np.zeros(mb_w * mb_h * 900 + 16, np.int16), 0, []
    for my in range(mb_h):
        for mx in range(mb_w):
            if rng.random() >= frac:
                continue
            d = G.make_intra_mb(rng, mx, my, mb_w, mb_h, depth=depth) if depth > 8 else G.make_intra_mb(rng, mx, my, mb_w, mb_h)
            rec = G.to_record(d)
            mb = d["mb"].copy()
            n = C.c_int32(ncoef)
            if depth > 8:
                r = L.ffhip_h264_intra_pack_hbd(depth, rec.ctypes.data, d["nnzc"].ctypes.data, mb.ctypes.data, d["luma_dc"].ctypes.data,
                                                G._p(d["pcm"], C.c_uint8), G._p(coefs, C.c_int16), C.byref(n), C.c_int32(coefs.size))
            else:
                r = L.ffhip_h264_intra_pack(rec.ctypes.data, d["nnzc"].ctypes.data, mb.ctypes.data, d["luma_dc"].ctypes.data,
                                            G._p(d["pcm"], C.c_uint8), G._p(coefs, C.c_int16), C.byref(n), C.c_int32(coefs.size))
            assert r == 0
            ncoef = n.value
            recs.append(rec)
            states.append(d)
    rows = np.zeros(mb_h + 1, np.int32)
    for r in recs:
        rows[int(r["mb_y"][0]) + 1] += 1
    rows = np.cumsum(rows).astype(np.int32)
    rec_arr = np.concatenate(recs) if recs else np.zeros(1, G.INTRA_DT)
    return rec_arr, rows, coefs[:max(ncoef, 8)].copy(), states


@pytest.mark.parametrize("depth,mb_w,mb_h,npics,frac", [(8, 20, 9, 5, 1.0), (8, 13, 7, 37, 0.6), (10, 16, 6, 4, 1.0), (8, 120, 68, 3, 1.0)])
def test_intra_frames_batch_equals_single_launches(depth, mb_w, mb_h, npics, frac):
    import torch
    from ffmpeg_amd import _lib
    L = _lib.lib()
    assert torch.cuda.is_available()
    L.ffhip_h264_intra_pack.restype = C.c_int
    L.ffhip_h264_intra_pack_hbd.restype = C.c_int
    rng = np.random.default_rng(depth * 1000 + mb_w + npics)
    ps = 2 if depth > 8 else 1
    sy, sc = mb_w * 16 * ps, mb_w * 8 * ps
    dt = np.uint16 if depth > 8 else np.uint8
    hi = 1 << depth
    pics, keep, singles = [], [], []
    for i in range(npics):
        rec, rows, coefs, states = _pack_picture(L, rng, mb_w, mb_h, frac, depth)
        planes = [rng.integers(0, hi, (mb_h * 16, mb_w * 16)).astype(dt), rng.integers(0, hi, (mb_h * 8, mb_w * 8)).astype(dt),
                  rng.integers(0, hi, (mb_h * 8, mb_w * 8)).astype(dt)]
        d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1, 108).copy()).cuda()
        d_rows, d_coef = torch.from_numpy(rows).cuda(), torch.from_numpy(coefs).cuda()
        batch = [torch.from_numpy(p.view(np.uint8).reshape(p.shape[0], -1).copy()).cuda() for p in planes]
        single = [t.clone() for t in batch]
        keep.append((d_rec, d_rows, d_coef, batch, single, planes, states))
        pics.append(IntraPic(batch[0].data_ptr(), batch[1].data_ptr(), batch[2].data_ptr(), d_rec.data_ptr(), d_rows.data_ptr(), d_coef.data_ptr()))
        singles.append(single)
    arr = (IntraPic * npics)(*pics)
    _lib.check(L.ffhip_h264_intra_frames_dev(depth, npics, C.cast(arr, C.c_void_p), sy, sc, mb_w, mb_h, None), "ffhip_h264_intra_frames_dev")
    for i in range(npics):
        d_rec, d_rows, d_coef, batch, single, planes, states = keep[i]
        if depth > 8:
            _lib.check(L.ffhip_h264_intra_frame_dev_hbd(depth, single[0].data_ptr(), single[1].data_ptr(), single[2].data_ptr(), sy, sc, mb_w, mb_h,
                                                        d_rec.data_ptr(), d_rows.data_ptr(), d_coef.data_ptr(), None), "intra_frame_dev_hbd")
        else:
            _lib.check(L.ffhip_h264_intra_frame_dev(single[0].data_ptr(), single[1].data_ptr(), single[2].data_ptr(), sy, sc, mb_w, mb_h,
                                                    d_rec.data_ptr(), d_rows.data_ptr(), d_coef.data_ptr(), None), "intra_frame_dev")
    torch.cuda.synchronize()
    changed = 0
    for i in range(npics):
        d_rec, d_rows, d_coef, batch, single, planes, states = keep[i]
        for pl in range(3):
            a, b = batch[pl].cpu().numpy(), single[pl].cpu().numpy()
            assert np.array_equal(a, b), (i, pl, np.argwhere(a != b)[:3])
            changed += int((a != planes[pl].view(np.uint8).reshape(a.shape)).sum())
    assert changed > 1000
    # picture 0 against the oracle, macroblock by macroblock in decoding order (8 bits: the oracle's restatement of hl_decode_mb)
    if depth == 8 and mb_w <= 20:
        O = ffi.oracle()
        d_rec, d_rows, d_coef, batch, single, planes, states = keep[0]
        want = [p.copy() for p in planes]
        for d in states:
            G.oracle_decode(O, d, want, [mb_w * 16, mb_w * 8, mb_w * 8])
        for pl in range(3):
            assert np.array_equal(batch[pl].cpu().numpy(), want[pl]), pl
