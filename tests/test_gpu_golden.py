"""GPU parity against the REAL reference's outputs directly: the HIP kernels on the committed golden inputs
(tests/golden/*.npz, written by tools/make_golden.py from oracle/_ref/libffref.so) must reproduce the reference's
outputs bit for bit — no oracle in between."""
import ctypes as C

import numpy as np
import pytest

import ffi
import test_golden as G

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_sws_golden_gpu():
    from ffmpeg_amd import swscale as S, _lib
    torch = _torch()
    n = 0
    for (sf, sw, sh, df, dw, dh, fl, unscaled), src, want, banks in G.sws_cases():
        tabs = None
        if banks is not None:   # drop-in construction from the REFERENCE's own banks
            ht = S.HostTables(sw, sh, sf, dw, dh, df, fl)
            tabs = _lib.SwsTables()
            C.memmove(C.byref(tabs), C.byref(ht.t), C.sizeof(tabs))
            keep = []
            for name in ("hLum", "hChr", "vLum", "vChr"):
                f, p, fs, nn = banks[name]
                f = np.ascontiguousarray(f, np.int16); p = np.ascontiguousarray(p, np.int32)
                keep += [f, p]
                setattr(tabs, name, _lib.SwsFilter(ffi.ptr(f, ffi.i16p), ffi.ptr(p, ffi.i32p), fs, nn))
        ctx = S.SwsContext(sw, sh, sf, dw, dh, df, fl, tables=tabs)
        dsrc = [torch.from_numpy(np.ascontiguousarray(a)).cuda().unsqueeze(0) for a in src]
        ddst = [torch.zeros((1,) + a.shape, dtype=torch.uint8, device="cuda:0") for a in want]
        ctx.scale_batch(dsrc, ddst)
        torch.cuda.synchronize()
        for p, a in enumerate(want):
            got = ddst[p][0].cpu().numpy()
            wv = a.shape[1] if not unscaled else (3 if df in (2, 3) else 4) * (sw & ~1)
            assert np.array_equal(got[:, :wv], a[:, :wv]), (sf, sw, sh, df, dw, dh, fl, p)
        # and the SwsFunc-shaped host face
        hd = [np.zeros_like(a) for a in want]
        assert ctx.scale(src, hd) == dh
        for a, b in zip(hd, want):
            wv = a.shape[1] if not unscaled else (3 if df in (2, 3) else 4) * (sw & ~1)
            assert np.array_equal(a[:, :wv], b[:, :wv])
        ctx.close()
        n += 1
    assert n >= 8


def test_h264_golden_gpu():
    from ffmpeg_amd import h264
    torch = _torch()
    d = G.load("h264")
    for which in range(4):
        dst, coef = d["idct%d_in_dst" % which].copy(), d["idct%d_in_coef" % which].copy()
        nblk, size, stride = dst.shape
        plane = torch.from_numpy(dst.reshape(nblk * size, stride)).cuda()
        offs = torch.arange(nblk, dtype=torch.int32, device="cuda:0") * (size * stride)
        dc = torch.from_numpy(coef).cuda()
        h264.idct_add_batch(which, plane, stride, offs, dc)
        assert np.array_equal(plane.cpu().numpy().reshape(dst.shape), d["idct%d_out_dst" % which])
        assert np.array_equal(dc.cpu().numpy(), d["idct%d_out_coef" % which])
    # loop filters: one 32x32 image per call, edge at (8, 8)
    imgs = d["lf_in"].copy()
    n = imgs.shape[0]
    ed = np.zeros(n, np.dtype([("offset", np.int32), ("kind", np.uint8), ("alpha", np.uint8), ("beta", np.uint8), ("pad", np.uint8),
                               ("tc0", np.int8, 4)]))
    for i, (which, alpha, beta, *tc0) in enumerate(d["lf_par"]):
        ed[i] = (i * 32 * 32 + 8 * 32 + 8, which, alpha, beta, 0, tc0)
    dimg = torch.from_numpy(imgs.reshape(n * 32, 32)).cuda()
    h264.loop_filter_batch(dimg, 32, torch.from_numpy(ed.view(np.uint8).reshape(n, 12)).cuda(), n)
    assert np.array_equal(dimg.cpu().numpy().reshape(imgs.shape), d["lf_out"])
    # qpel: 96 functions on one source image
    src = torch.from_numpy(np.ascontiguousarray(d["qpel_src"])).cuda()
    par = d["qpel_par"]
    dsts = torch.from_numpy(np.tile(d["qpel_dst"], (len(par), 1))).cuda()          # one 32-row copy per call
    blk = np.zeros(len(par), np.dtype([("d", np.int32), ("s", np.int32), ("mc", np.uint8), ("sz", np.uint8), ("avg", np.uint8),
                                       ("flags", np.uint8), ("sx", np.int16), ("sy", np.int16)]))
    # src and dst live in different tensors: the batch face takes one stride and two bases
    for i, (avg, size_idx, mc) in enumerate(par):
        blk[i] = (i * 32 * 64 + 6 * 64 + 8, 6 * 64 + 8, mc, size_idx, avg, 0, 0, 0)
    h264.qpel_batch(dsts, src, 64, torch.from_numpy(blk.view(np.uint8).reshape(-1, 16)).cuda(), len(par))
    got = dsts.cpu().numpy().reshape(len(par), 32, 64)
    for i in range(len(par)):
        assert np.array_equal(got[i, 6:22, 8:24], d["qpel_out"][i]), tuple(par[i])
    # chroma MC and weights
    csrc = torch.from_numpy(np.ascontiguousarray(d["chroma_src"])).cuda()
    cpar = d["chroma_par"]
    cd = torch.from_numpy(np.tile(d["chroma_dst"], (len(cpar), 1))).cuda()
    cb = np.zeros(len(cpar), np.dtype([("d", np.int32), ("s", np.int32), ("w", np.uint8), ("h", np.uint8), ("x", np.uint8),
                                       ("y", np.uint8), ("avg", np.uint8), ("flags", np.uint8), ("sx", np.int16), ("sy", np.int16), ("pad", np.int16)]))
    for i, (avg, idx, x, y) in enumerate(cpar):
        cb[i] = (i * 24 * 32 + 2 * 32 + 8, 2 * 32 + 8, idx, 8, x, y, avg, 0, 0, 0, 0)
    h264.chroma_mc_batch(cd, csrc, 32, torch.from_numpy(cb.view(np.uint8).reshape(-1, 20)).cuda(), len(cpar))
    got = cd.cpu().numpy().reshape(len(cpar), 24, 32)
    for i in range(len(cpar)):
        assert np.array_equal(got[i, 2:10, 8:16], d["chroma_out"][i]), tuple(cpar[i])
    wpar = d["weight_par"]
    wd = torch.from_numpy(np.tile(d["chroma_dst"], (len(wpar), 1))).cuda()
    wb = np.zeros(len(wpar), np.dtype([("d", np.int32), ("s", np.int32), ("w", np.uint8), ("h", np.uint8), ("ld", np.uint8),
                                       ("bi", np.uint8), ("wd", np.int16), ("ws", np.int16), ("of", np.int16), ("pad", np.int16)]))
    for i, (bi, idx, ld, wt, ws, of) in enumerate(wpar):
        wb[i] = (i * 24 * 32 + 2 * 32 + 8, 2 * 32 + 8, idx, 16, ld, bi, wt, ws, of, 0)
    h264.weight_batch(wd, csrc, 32, torch.from_numpy(wb.view(np.uint8).reshape(-1, 20)).cuda(), len(wpar))
    got = wd.cpu().numpy().reshape(len(wpar), 24, 32)
    for i in range(len(wpar)):
        assert np.array_equal(got[i, 2:18, 8:24], d["weight_out"][i]), tuple(wpar[i])


def test_me_golden_gpu():
    from ffmpeg_amd import me
    torch = _torch()
    d = G.load("me")
    a, b = torch.from_numpy(np.ascontiguousarray(d["cmp_a"])).cuda(), torch.from_numpy(np.ascontiguousarray(d["cmp_b"])).cuda()
    pos, vals = d["cmp_pos"], d["cmp_vals"]
    o1 = torch.from_numpy((pos[:, 0] * 64 + pos[:, 1]).astype(np.int32)).cuda()
    o2 = torch.from_numpy((pos[:, 2] * 64 + pos[:, 3]).astype(np.int32)).cuda()
    for col, (kind, width, h) in enumerate(((0, 16, 16), (0, 16, 8), (0, 8, 8), (1, 16, 16), (1, 16, 8), (1, 8, 8))):
        out = torch.zeros(len(pos), dtype=torch.int32, device="cuda:0")
        me.cmp_batch(kind, width, h, a, o1, b, o2, 64, out)
        assert np.array_equal(out.cpu().numpy(), vals[:, col]), (kind, width, h)
    cur, ref = np.ascontiguousarray(d["esa_cur"]), np.ascontiguousarray(d["esa_ref"])
    h, w = cur.shape
    for Rr in (3, 7):
        mv = torch.zeros((h // 16) * (w // 16) * 2, dtype=torch.int16, device="cuda:0")
        cost = torch.zeros((h // 16) * (w // 16), dtype=torch.int32, device="cuda:0")
        me.esa_batch(torch.from_numpy(cur).cuda(), torch.from_numpy(ref).cuda(), w, h, w, w * h, 1, 16, Rr, me.SAD, mv, cost)
        assert np.array_equal(mv.cpu().numpy().reshape(-1, 2), d["esa_mv_r%d" % Rr])
        assert np.array_equal(cost.cpu().numpy().view(np.uint32), d["esa_cost_r%d" % Rr].astype(np.uint32))


def test_tx_golden_gpu():
    from ffmpeg_amd import tx
    torch = _torch()
    d = G.load("tx")
    for key in d["keys"]:
        key = str(key)
        _, inv, scale = key.split("_")
        len_ = int(key.split("_")[0][4:])
        x, want = d[key + "_in"], d[key + "_out"]
        ctx = tx.TxContext(tx.FLOAT_MDCT, int(inv), len_, float(scale), flags=tx.BITEXACT)
        out = torch.zeros((x.shape[0], len_), dtype=torch.float32, device="cuda:0")
        ctx.batch(out, torch.from_numpy(np.ascontiguousarray(x)).cuda())
        got = out.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), key
        ctx.close()


def test_hevc_golden_gpu():
    """the HIP transforms on the reference's golden inputs reproduce the reference's outputs"""
    from ffmpeg_amd import hevc
    torch = _torch()
    d = G.load("hevc")
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        blocks, limits = d["in%d" % lg], d["lim%d" % lg]
        nt = len(blocks)
        tus = np.zeros(nt, hevc.TU_DTYPE)
        tus["coeff_offset"] = np.arange(nt) * n * n
        tus["dst_offset"] = -1
        tus["col_limit"] = limits
        d_t = torch.from_numpy(tus.view(np.uint8).reshape(nt, 12).copy()).cuda()
        for kind, key in ((hevc.IDCT, "idct%d" % lg), (hevc.IDCT_DC, "dc%d" % lg)) + (((hevc.DST_4X4, "dst4"),) if lg == 2 else ()):
            d_c = torch.from_numpy(blocks.copy()).cuda()
            hevc.idct_batch(kind, lg, d_c, None, 0, d_t, nt)
            torch.cuda.synchronize()
            assert np.array_equal(d_c.cpu().numpy(), d[key]), key
        pic = d["pic%d" % lg]
        tus["dst_offset"] = np.arange(nt) * pic.shape[1] * 48 + 48 + 5
        d_t = torch.from_numpy(tus.view(np.uint8).reshape(nt, 12).copy()).cuda()
        d_c = torch.from_numpy(blocks.copy()).cuda()
        d_p = torch.from_numpy(pic.copy()).cuda()
        hevc.idct_batch(hevc.ADD_ONLY, lg, d_c, d_p, 48, d_t, nt)
        torch.cuda.synchronize()
        assert np.array_equal(d_p.cpu().numpy(), d["add%d" % lg]), lg


def test_hevc_lf_golden_gpu():
    """the HIP loop filters on the reference's golden neighbourhoods (one call per direction: pixels are disjoint)"""
    from ffmpeg_amd import hevc
    torch = _torch()
    d = G.load("hevc")
    par = d["lf_par"]
    n = len(par)
    ed = np.zeros(n, hevc.EDGE_DTYPE)
    ed["offset"] = np.arange(n) * 256 + np.where(par[:, 0] & 1, 4 * 16 + 8, 8 * 16 + 4)
    ed["kind"], ed["beta"] = par[:, 0], par[:, 1]
    ed["tc"], ed["no_p"], ed["no_q"] = par[:, 2:4], par[:, 4:6], par[:, 6:8]
    pic = torch.from_numpy(d["lf_in"].copy()).cuda()
    hevc.loop_filter_batch(pic, 16, torch.from_numpy(ed.view(np.uint8).reshape(n, 16).copy()).cuda(), n)
    torch.cuda.synchronize()
    assert np.array_equal(pic.cpu().numpy(), d["lf_out"])


def test_hevc_sao_golden_gpu():
    from ffmpeg_amd import hevc
    torch = _torch()
    d = G.load("hevc")
    par = d["sao_par"]
    n = len(par)
    rec = np.zeros(n, hevc.SAO_DTYPE)
    rec["dst_offset"] = np.arange(n) * 32 * 64
    rec["src_offset"] = np.arange(n) * 34 * 192 + 193
    rec["edge"], rec["cls"], rec["width"], rec["height"] = par[:, 0], par[:, 1], par[:, 2], par[:, 3]
    rec["offset_val"] = par[:, 4:9]
    dst = torch.zeros((n * 32, 64), dtype=torch.uint8, device="cuda:0")
    hevc.sao_batch(dst, 64, torch.from_numpy(d["sao_src"].copy()).cuda(), 192, torch.from_numpy(rec.view(np.uint8).reshape(n, 24).copy()).cuda(), n)
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy().reshape(n, 32, 64), d["sao_out"])


def test_hevc_mc_golden_gpu():
    from ffmpeg_amd import hevc
    torch = _torch()
    d = G.load("hevc")
    par = d["mc_par"]
    ref = torch.from_numpy(d["mc_ref"].copy()).cuda()
    for chroma in (0, 1):
        sel = np.nonzero(par[:, 0] == chroma)[0]
        n = len(sel)
        rec = np.zeros(n, hevc.MC_DTYPE)
        rec["src_offset"] = par[sel, 5] * 96 + par[sel, 6]
        rec["width"], rec["height"], rec["mx"], rec["my"] = par[sel, 1], par[sel, 2], par[sel, 3], par[sel, 4]
        rec["dst_offset"] = np.arange(n) * 4096
        d_rec = torch.from_numpy(rec.view(np.uint8).reshape(n, 12).copy()).cuda()
        o16 = torch.zeros((n, 64, 64), dtype=torch.int16, device="cuda:0")
        hevc.mc_batch(chroma, 0, o16, 0, ref, 96, d_rec, n)
        o8 = torch.zeros((n, 64, 64), dtype=torch.uint8, device="cuda:0")
        hevc.mc_batch(chroma, 1, o8, 64, ref, 96, d_rec, n)
        torch.cuda.synchronize()
        assert np.array_equal(o16.cpu().numpy(), d["mc_out16"][sel]) and np.array_equal(o8.cpu().numpy(), d["mc_out8"][sel])


def test_hevc_mc_weighted_golden_gpu():
    from ffmpeg_amd import hevc
    torch = _torch()
    d = G.load("hevc")
    par = d["mcw_par"]
    ref = torch.from_numpy(d["mc_ref"].copy()).cuda()
    src2 = torch.from_numpy(d["mcw_src2"].copy()).cuda()
    for chroma in (0, 1):
        for mode in (2, 3, 4):
            sel = np.nonzero((par[:, 0] == chroma) & (par[:, 1] == mode))[0]
            n = len(sel)
            assert n
            rec = np.zeros(n, hevc.MCW_DTYPE)
            rec["src_offset"] = par[sel, 6] * 96 + par[sel, 7]
            rec["width"], rec["height"], rec["mx"], rec["my"] = par[sel, 2], par[sel, 3], par[sel, 4], par[sel, 5]
            rec["denom"], rec["wx0"], rec["wx1"], rec["ox"] = par[sel, 8], par[sel, 9], par[sel, 10], par[sel, 11]
            rec["dst_offset"] = np.arange(n) * 4096
            o8 = torch.zeros((n, 64, 64), dtype=torch.uint8, device="cuda:0")
            hevc.mc_w_batch(chroma, mode, o8, 64, ref, 96, src2, torch.from_numpy(rec.view(np.uint8).reshape(n, 24).copy()).cuda(), n)
            torch.cuda.synchronize()
            assert np.array_equal(o8.cpu().numpy(), d["mcw_out"][sel]), (chroma, mode)


def test_fdsp_golden_gpu():
    from ffmpeg_amd import fdsp
    torch = _torch()
    d = G.load("fdsp")
    for op in range(7):
        for n in (1024, 37):
            k = "op%d_n%d_" % (op, n)
            t = [torch.from_numpy(np.ascontiguousarray(d[k + name]).reshape(1, -1)).cuda() for name in ("dst", "s0", "s1", "s2")]
            fdsp.batch(op, t[0], t[1], t[2], t[3], float(d[k + "mul"][0]), n)
            torch.cuda.synchronize()
            assert np.array_equal(t[0].cpu().numpy().reshape(-1).view(np.uint32), d[k + "out"]), (op, n)
            assert np.array_equal(t[1].cpu().numpy().reshape(-1).view(np.uint32), d[k + "out0"]), (op, n)


def test_fft_golden_gpu():
    from ffmpeg_amd import tx
    torch = _torch()
    d = G.load("fft")
    for len_ in (8, 256, 1024):
        for inv in (0, 1):
            x, want = d["fft%d_%d_in" % (len_, inv)], d["fft%d_%d_out" % (len_, inv)]
            ctx = tx.TxContext(tx.FLOAT_FFT, inv, len_, 1.0, flags=tx.BITEXACT)
            out = torch.zeros((x.shape[0], 2 * len_), dtype=torch.float32, device="cuda:0")
            ctx.batch(out, torch.from_numpy(np.ascontiguousarray(x)).cuda())
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)), (len_, inv)
            ctx.close()


def test_rdft_golden_gpu():
    from ffmpeg_amd import tx
    torch = _torch()
    d = G.load("fft")
    for len_ in (16, 1024):
        for inv in (0, 1):
            x, want = d["rdft%d_%d_in" % (len_, inv)], d["rdft%d_%d_out" % (len_, inv)]
            ctx = tx.TxContext(tx.FLOAT_RDFT, inv, len_, 1.0, flags=tx.BITEXACT)
            out = torch.zeros(want.shape, dtype=torch.float32, device="cuda:0")
            ctx.batch(out, torch.from_numpy(np.ascontiguousarray(x)).cuda())
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)), (len_, inv)
            ctx.close()


def test_rdft_half_golden_gpu():
    from ffmpeg_amd import tx
    torch = _torch()
    d = G.load("fft")
    for len_ in (16, 1024):
        for mode in (1, 2):
            x, want = d["rdfth%d_%d_in" % (len_, mode)], d["rdfth%d_%d_out" % (len_, mode)]
            ctx = tx.TxContext(tx.FLOAT_RDFT, 0, len_, 1.0, flags=(tx.REAL_TO_REAL if mode == 1 else tx.REAL_TO_IMAGINARY) | tx.BITEXACT)
            out = torch.zeros((want.shape[0], len_ // 2 + 2), dtype=torch.float32, device="cuda:0")
            ctx.batch(out[:, :want.shape[1]], torch.from_numpy(np.ascontiguousarray(x)).cuda())
            assert np.array_equal(np.ascontiguousarray(out.cpu().numpy()[:, :want.shape[1]]).view(np.uint32), want.view(np.uint32)), (len_, mode)
            ctx.close()


def test_dct_golden_gpu():
    from ffmpeg_amd import tx
    torch = _torch()
    d = G.load("fft")
    for n in (16, 1024):
        for inv in (0, 1):
            x, want = d["dct%d_%d_in" % (n, inv)], d["dct%d_%d_out" % (n, inv)]
            ctx = tx.TxContext(tx.FLOAT_DCT, inv, n >> inv, 1.0, flags=tx.BITEXACT)
            out = torch.zeros(want.shape, dtype=torch.float32, device="cuda:0")
            ctx.batch(out, torch.from_numpy(np.ascontiguousarray(x)).cuda())
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)), (n, inv)
            ctx.close()


def test_h264_pred_golden_gpu():
    """H264PredContext batch kinds against the reference's stored outputs (tests/golden/h264pred.npz)"""
    from test_gpu_h264_pred import hip_pred_apply
    G.h264_pred_golden_check(hip_pred_apply, G.load("h264pred"))


def test_h264_pred422_golden_gpu():
    """pred8x8[] at chroma_format_idc 2 against the reference's stored outputs (tests/golden/h264pred422.npz): the batch face at 8
    bits, the host faces of ff_h264_pred_init_hip(h, H264, 10, 2) at 10"""
    import ctypes as C
    from ffmpeg_amd import h264
    from test_gpu_h264_pred import hip_pred_apply
    d = G.load("h264pred422")
    G.h264_pred422_golden_check(hip_pred_apply, d)
    hc = h264.pred_init(bit_depth=10, chroma_format_idc=2)
    p = np.ascontiguousarray(d["p10_in"])
    for mode in range(11):
        hc.pred8x8[mode](C.c_void_p(p[mode].ctypes.data + (8 * 48 + 16) * 2), 96)
    assert np.array_equal(p, d["p10_out"])


def test_round4_alpha_both_sides_golden_gpu():
    """planar YUVA -> planar YUVA through libffhip == the reference's four planes (tests/golden/round4.npz)"""
    import torch
    from ffmpeg_amd import swscale as S
    d = G.load("round4")
    for k in range(int(d["a_n"])):
        sf, sw, sh, df, dw, dh, flags = (int(v) for v in d["a%d_meta" % k])
        src = [np.ascontiguousarray(d["a%d_src%d" % (k, p)]) for p in range(4)]
        want = [d["a%d_dst%d" % (k, p)] for p in range(4)]
        ctx = S.SwsContext(sw, sh, sf, dw, dh, df, flags)
        dsrc = [torch.from_numpy(a).cuda()[None].contiguous() for a in src]
        ddst = [torch.zeros((1,) + a.shape, dtype=torch.uint8, device="cuda") for a in want]
        ctx.scale_batch(dsrc, ddst)
        torch.cuda.synchronize()
        for p in range(4):
            got = ddst[p][0].cpu().numpy()
            w = want[p].shape[1]
            assert np.array_equal(got[:, :w], want[p]), (k, p, int((got[:, :w] != want[p]).sum()))
        ctx.close()


def test_round4_vp9_loopfilter_422_440_golden_gpu():
    """a picture of one superblock at 4:2:2 / 4:4:0 through ffhip_vp9_loopfilter_frame_ssc_dev == the reference's ff_vp9_loopfilter_sb
    (the fixture's first-row, first-column cases)"""
    import torch
    import vp9_lf_gen as VG
    from ffmpeg_amd import vp9
    d = G.load("round4")
    lim, mblim = np.ascontiguousarray(d["lf_lim"]), np.ascontiguousarray(d["lf_mblim"])
    ran = 0
    for n in range(int(d["lf_n"])):
        bd, ss_h, ss_v, row, col = (int(v) for v in d["lf%d_par" % n])
        if row or col:
            continue
        cw, chh = 64 >> ss_h, 64 >> ss_v
        pos = ((64, 64), (chh, cw), (chh, cw))
        f = np.zeros(1, VG.FILTER_DT)
        f["level"][0], f["mask"][0] = d["lf%d_level" % n], d["lf%d_mask" % n]
        tabs, ctabs = vp9.lf_sb_tables_ss(f.view(np.uint8).reshape(1, 192), 1, 1, lim, mblim, (ss_h, ss_v))
        raw = [d["lf%d_in%d" % (n, k)] for k in range(3)]
        ins = []                                 # the fixture's rows are w + 7 samples: the frame kernel wants 4-byte aligned pitches
        for a in raw:
            b = np.zeros((a.shape[0], (a.shape[1] + 7) & ~7), a.dtype)
            b[:, :a.shape[1]] = a
            ins.append(b)
        dev = [torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda() for a in ins]
        at = [t[r * a.strides[0] + c * a.itemsize:] for t, a, (r, c) in zip(dev, ins, pos)]
        vp9.loopfilter_frame_ssc(at[0], at[1], at[2], ins[0].strides[0], ins[1].strides[0], 8, 8, torch.from_numpy(tabs.view(np.int32)).cuda(),
                                 torch.from_numpy(ctabs.view(np.int32)).cuda(), (ss_h, ss_v), bit_depth=bd)
        torch.cuda.synchronize()
        for k in range(3):
            got = dev[k].cpu().numpy().view(ins[k].dtype).reshape(ins[k].shape)
            assert np.array_equal(got[:, :raw[k].shape[1]], d["lf%d_out%d" % (n, k)]), (n, k)
            assert not got[:, raw[k].shape[1]:].any()
        ran += 1
    assert ran == 4
