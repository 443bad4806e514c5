"""The oracle against the committed golden vectors (tests/golden/*.npz, outputs of the real reference written by
tools/make_golden.py).  Runs anywhere — this is what pins oracle/ on a box without /root/reference."""
import ctypes as C
import os

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p, i8p, i16p, i32p, f32p

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def at(a, off):
    return C.cast(a.ctypes.data + int(off), u8p)


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def sws_cases():
    d = load("sws")
    for i in range(int(d["ncases"][0])):
        sf, sw, sh, df, dw, dh, fl, unscaled = [int(v) for v in d["c%d_meta" % i]]
        src = [np.ascontiguousarray(d["c%d_src%d" % (i, p)]) for p in range(3) if "c%d_src%d" % (i, p) in d]
        dst = [d["c%d_dst%d" % (i, p)] for p in range(3) if "c%d_dst%d" % (i, p) in d]
        banks = None
        if not unscaled:
            banks = {}
            for name in ("hLum", "hChr", "vLum", "vChr"):
                fs, n = [int(v) for v in d["c%d_%s_s" % (i, name)]]
                banks[name] = (d["c%d_%s_f" % (i, name)], d["c%d_%s_p" % (i, name)], fs, n)
        yield (sf, sw, sh, df, dw, dh, fl, unscaled), src, dst, banks


def test_sws_golden():
    O = ffi.oracle()
    n = 0
    for (sf, sw, sh, df, dw, dh, fl, unscaled), src, want, banks in sws_cases():
        sp, ss = ffi.planes(src)
        if unscaled:
            luts = ffi.OLuts()
            k = ffi.OYuv2RgbCoeffs(*[ffi.DEFAULT_COEFFS[x] for x in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
            O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
            got = np.zeros_like(want[0])
            O.ffo_yuv420p_to_rgb24(C.byref(luts), sw, sp, ss, 0, sh, ptr(got), got.strides[0], ffi.RGB_LAYOUT[df])
            wv = (3 if df in (2, 3) else 4) * (sw & ~1)
            assert np.array_equal(got[:, :wv], want[0][:, :wv])
        else:
            t = ffi.make_otables(sw, sh, sf, dw, dh, df, fl, banks)
            got = ffi.alloc_frame(df, dw, dh)
            dp, ds = ffi.planes(got)
            assert O.ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
        n += 1
    assert n >= 8


def test_sws_host_tables_golden():
    """our initFilter() restatement (host logic of the product) reproduces the reference's banks"""
    from ffmpeg_amd import swscale as S
    for (sf, sw, sh, df, dw, dh, fl, unscaled), _, _, banks in sws_cases():
        ht = S.HostTables(sw, sh, sf, dw, dh, df, fl)
        assert ht.unscaled_yuv2rgb == bool(unscaled)
        if unscaled:
            continue
        for name in ("hLum", "hChr", "vLum", "vChr"):
            f, p, fs, n = ht.bank(name)
            rf, rp, rfs, rn = banks[name]
            assert (fs, n) == (rfs, rn) and np.array_equal(p, rp) and np.array_equal(f, rf), name


def test_h264_golden():
    O = ffi.oracle()
    d = load("h264")
    fns = [O.ffo_h264_idct_add, O.ffo_h264_idct8_add, O.ffo_h264_idct_dc_add, O.ffo_h264_idct8_dc_add]
    for which in range(4):
        dst, coef = d["idct%d_in_dst" % which].copy(), d["idct%d_in_coef" % which].copy()
        for i in range(dst.shape[0]):
            fns[which](ptr(dst[i]), ptr(coef[i], i16p), dst.shape[2])
        assert np.array_equal(dst, d["idct%d_out_dst" % which]) and np.array_equal(coef, d["idct%d_out_coef" % which])
    imgs = d["lf_in"].copy()
    for i, (which, alpha, beta, *tc0) in enumerate(d["lf_par"]):
        O.ffo_h264_loop_filter(int(which), at(imgs[i], 8 * 32 + 8), 32, int(alpha), int(beta), ptr(np.array(tc0, np.int8), i8p))
    assert np.array_equal(imgs, d["lf_out"])
    src = np.ascontiguousarray(d["qpel_src"])
    for i, (avg, size_idx, mc) in enumerate(d["qpel_par"]):
        o = d["qpel_dst"].copy()
        O.ffo_h264_qpel(int(avg), int(size_idx), int(mc), at(o, 6 * 64 + 8), at(src, 6 * 64 + 8), 64)
        assert np.array_equal(o[6:22, 8:24], d["qpel_out"][i]), (avg, size_idx, mc)


def test_h264_misc_golden():
    """idct_add8, luma / chroma dc_dequant_idct, add_pixels{4,8}_clear: oracle == the reference's committed outputs"""
    from test_oracle_vs_ref import h264_misc_apply
    g = load("h264_misc")
    bo, stride = g["bo"], int(g["stride"])
    names = ("cb", "cr", "blocks", "out", "dc", "cdc", "px4", "res4", "px8", "res8")
    for i in range(int(g["n"])):
        k = {key: (int(g["in%d_%s" % (i, key)]) if key == "qmul" else g["in%d_%s" % (i, key)])
             for key in ("blocks", "nnzc", "cb", "cr", "dc", "out", "cdc", "qmul", "px", "res")}
        for nm, v in zip(names, h264_misc_apply(ffi.oracle(), "ffo", bo, stride, k)):
            assert np.array_equal(v, g["out%d_%s" % (i, nm)]), (i, nm)


def test_h264_chroma_weight_golden():
    O = ffi.oracle()
    d = load("h264")
    src = np.ascontiguousarray(d["chroma_src"])
    for i, (avg, idx, x, y) in enumerate(d["chroma_par"]):
        o = d["chroma_dst"].copy()
        O.ffo_h264_chroma_mc(int(avg), 8 >> int(idx), at(o, 2 * 32 + 8), at(src, 2 * 32 + 8), 32, 8, int(x), int(y))
        assert np.array_equal(o[2:10, 8:16], d["chroma_out"][i]), (avg, idx, x, y)
    for i, (bi, idx, ld, wt, ws, of) in enumerate(d["weight_par"]):
        o = d["chroma_dst"].copy()
        if bi:
            O.ffo_h264_biweight(16 >> int(idx), at(o, 2 * 32 + 8), at(src, 2 * 32 + 8), 32, 16, int(ld), int(wt), int(ws), int(of))
        else:
            O.ffo_h264_weight(16 >> int(idx), at(o, 2 * 32 + 8), 32, 16, int(ld), int(wt), int(of))
        assert np.array_equal(o[2:18, 8:24], d["weight_out"][i]), (bi, idx, ld, wt, ws, of)


def test_me_golden():
    O = ffi.oracle()
    d = load("me")
    a, b = np.ascontiguousarray(d["cmp_a"]), np.ascontiguousarray(d["cmp_b"])
    for (y1, x1, y2, x2), want in zip(d["cmp_pos"], d["cmp_vals"]):
        pa, pb = at(a, y1 * 64 + x1), at(b, y2 * 64 + x2)
        got = [O.ffo_sad(16, pa, pb, 64, 16), O.ffo_sad(16, pa, pb, 64, 8), O.ffo_sad(8, pa, pb, 64, 8),
               O.ffo_hadamard8_diff16(pa, pb, 64, 16), O.ffo_hadamard8_diff16(pa, pb, 64, 8), O.ffo_hadamard8_diff8x8(pa, pb, 64)]
        assert got == list(want)
    cur, ref = np.ascontiguousarray(d["esa_cur"]), np.ascontiguousarray(d["esa_ref"])
    h, w = cur.shape
    for Rr in (3, 7):
        mv = np.zeros((h // 16) * (w // 16) * 2, np.int16)
        cost = np.zeros((h // 16) * (w // 16), np.uint32)
        O.ffo_me_esa_frame(ptr(cur), ptr(ref), w, w, h, 16, Rr, 0, ptr(mv, i16p), cost.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert np.array_equal(mv.reshape(-1, 2), d["esa_mv_r%d" % Rr]) and np.array_equal(cost, d["esa_cost_r%d" % Rr])


def test_tx_golden():
    O = ffi.oracle()
    d = load("tx")
    n = 0
    for key in d["keys"]:
        key = str(key)
        _, inv, scale = key.split("_")
        len_ = int(key.split("_")[0][4:])
        x, want = d[key + "_in"], d[key + "_out"]
        s = O.ffo_mdct_create(int(inv), len_, float(scale))
        for t in range(x.shape[0]):
            out = np.zeros(len_, np.float32)
            O.ffo_mdct_run(s, ptr(out, f32p), ptr(np.ascontiguousarray(x[t]), f32p), 4)
            assert np.array_equal(out.view(np.uint32), want[t].view(np.uint32)), key
        O.ffo_mdct_free(s)
        n += 1
    assert n == 12


def test_hevc_golden():
    O = ffi.oracle()
    d = load("hevc")
    for lg in (2, 3, 4, 5):
        blocks, limits = d["in%d" % lg], d["lim%d" % lg]
        a = blocks.copy()
        for t in range(len(a)):
            O.ffo_hevc_idct(lg, ptr(a[t], i16p), int(limits[t]))
        assert np.array_equal(a, d["idct%d" % lg]), lg
        a = blocks.copy()
        for t in range(len(a)):
            O.ffo_hevc_idct_dc(lg, ptr(a[t], i16p))
        assert np.array_equal(a, d["dc%d" % lg]), lg
        o = d["pic%d" % lg].copy()
        for t in range(len(o)):
            O.ffo_hevc_add_residual(lg, at(o[t], 48 + 5), ptr(np.ascontiguousarray(blocks[t]), i16p), 48)
        assert np.array_equal(o, d["add%d" % lg]), lg
    a = d["in2"].copy()
    for t in range(len(a)):
        O.ffo_hevc_transform_4x4_luma(ptr(a[t], i16p))
    assert np.array_equal(a, d["dst4"])
    o = d["lf_in"].copy()
    for i, (which, beta, t0, t1, p0, p1, q0, q1) in enumerate(d["lf_par"]):
        off = 4 * 16 + 8 if which & 1 else 8 * 16 + 4
        O.ffo_hevc_loop_filter(int(which) >> 1, int(which) & 1, at(o[i], off), 16, int(beta), ptr(np.array([t0, t1], np.int32), i32p),
                               ptr(np.array([p0, p1], np.uint8)), ptr(np.array([q0, q1], np.uint8)))
    assert np.array_equal(o, d["lf_out"])
    for i, (edge, cls, w, h, *off) in enumerate(d["sao_par"]):
        dst = np.zeros((32, 64), np.uint8)
        src = np.ascontiguousarray(d["sao_src"][i])
        o16 = np.array(off, np.int16)
        if edge:
            O.ffo_hevc_sao_edge(ptr(dst), at(src, 193), 64, 192, ptr(o16, i16p), int(cls), int(w), int(h))
        else:
            O.ffo_hevc_sao_band(ptr(dst), at(src, 193), 64, 192, ptr(o16, i16p), int(cls), int(w), int(h))
        assert np.array_equal(dst, d["sao_out"][i]), i
    mref = np.ascontiguousarray(d["mc_ref"])
    for i, (chroma, w, h, mx, my, y0, x0) in enumerate(d["mc_par"]):
        a16, a8 = np.zeros((64, 64), np.int16), np.zeros((64, 64), np.uint8)
        O.ffo_hevc_mc(int(chroma), 0, a16.ctypes.data, 0, at(mref, y0 * 96 + x0), 96, int(h), int(mx), int(my), int(w))
        O.ffo_hevc_mc(int(chroma), 1, a8.ctypes.data, 64, at(mref, y0 * 96 + x0), 96, int(h), int(mx), int(my), int(w))
        assert np.array_equal(a16, d["mc_out16"][i]) and np.array_equal(a8, d["mc_out8"][i]), i
    src2 = np.ascontiguousarray(d["mcw_src2"])
    for i, (chroma, mode, w, h, mx, my, y0, x0, den, wx0, wx1, ox) in enumerate(d["mcw_par"].tolist()):
        a8 = np.zeros((64, 64), np.uint8)
        O.ffo_hevc_mc_w(chroma, mode, ptr(a8), 64, at(mref, y0 * 96 + x0), 96, ptr(src2, i16p), h, den, wx0, wx1, ox, mx, my, w)
        assert np.array_equal(a8, d["mcw_out"][i]), ("mcw", i)


def test_vp9_golden():
    O = ffi.oracle()
    d = load("vp9")
    for tx in range(5):
        n = 4 if tx == 4 else 4 << tx
        for i, (txtp, eob) in enumerate(d["tx%d_par" % tx].tolist()):
            o, b = d["tx%d_dst" % tx][i].copy(), d["tx%d_blk" % tx][i].copy()
            O.ffo_vp9_itxfm_add(tx, txtp, ptr(o), n, ptr(b, i16p), eob)
            assert np.array_equal(o, d["tx%d_out" % tx][i]) and np.array_equal(b, d["tx%d_oblk" % tx][i]), (tx, i)


def test_vp9_mc_golden():
    O = ffi.oracle()
    d = load("vp9")
    mref = np.ascontiguousarray(d["mc_ref"])
    for i, (f, avg, w, h, mx, my, y0, x0) in enumerate(d["mc_par"].tolist()):
        a = d["mc_in"].copy()
        O.ffo_vp9_mc(f, avg, ptr(a), 64, at(mref, y0 * 96 + x0), 96, w, h, mx, my)
        assert np.array_equal(a[h:], d["mc_in"][h:]) and np.array_equal(a[:, w:], d["mc_in"][:, w:])
        assert np.array_equal(a[:h, :w], d["mc_out"][i][:h, :w]), i


def h264_pred_golden_check(apply, d):
    """apply(kind, pic, recs, coeffs): every kind's blocks against the stored reference outputs; nothing else may change"""
    from test_oracle_vs_ref import H264_PRED_KINDS
    for kind, (n, _, _) in enumerate(H264_PRED_KINDS):
        recs = d["k%d_rec" % kind]
        coeffs = d["k%d_coef" % kind].copy() if kind >= 4 else None
        pic = d["pic"].copy()
        pic = apply(kind, pic, recs, coeffs)
        mask = np.ones(pic.shape, bool)
        for i, (x, y, *_) in enumerate(recs.tolist()):
            assert np.array_equal(pic[y:y + n, x:x + n], d["k%d_out" % kind][i]), (kind, i, recs[i])
            mask[y:y + n, x:x + n] = False
        assert np.array_equal(pic[mask], d["pic"][mask]), kind


def test_h264_pred_golden():
    from test_oracle_vs_ref import h264_pred_apply
    O = ffi.oracle()

    def apply(kind, pic, recs, coeffs):
        h264_pred_apply(O, "ffo", kind, pic, recs, coeffs)
        assert coeffs is None or not coeffs.any()
        return pic
    h264_pred_golden_check(apply, load("h264pred"))


def h264_pred422_golden_check(apply, d):
    """kind 7 (pred8x8[] at 4:2:2, 8 bits): the blocks against the reference's stored outputs; nothing else may change"""
    recs = d["k7_rec"]
    pic = apply(7, d["pic"].copy(), recs, None)
    mask = np.ones(pic.shape, bool)
    for i, (x, y, *_) in enumerate(recs.tolist()):
        assert np.array_equal(pic[y:y + 16, x:x + 8], d["k7_out"][i]), (i, recs[i])
        mask[y:y + 16, x:x + 8] = False
    assert np.array_equal(pic[mask], d["pic"][mask])


def test_h264_pred422_golden():
    from test_oracle_vs_ref import h264_pred_apply
    O = ffi.oracle()

    def apply(kind, pic, recs, coeffs):
        h264_pred_apply(O, "ffo", kind, pic, recs, coeffs)
        return pic
    h264_pred422_golden_check(apply, load("h264pred422"))


def aac_golden_windows(d):
    return [np.ascontiguousarray(d[k]) for k in ("sine_1024", "sine_128", "kbd_long_1024", "kbd_short_128")]


def test_aac_golden():
    """the oracle on the stored frames (decoder's own window tables from the fixture): outputs and final overlap state"""
    O = ffi.oracle()
    d = load("aac")
    win = aac_golden_windows(d)
    wp = (f32p * 4)(*[ptr(w, f32p) for w in win])
    m1024, m128 = O.ffo_mdct_create(1, 1024, 2.0 ** -25), O.ffo_mdct_create(1, 128, 2.0 ** -22)
    saved = d["saved_in"].copy()
    prev = tuple(int(v) for v in d["prev"])
    for f in range(len(d["seq"])):
        s2 = np.array([d["seq"][f], prev[0]], np.int32); k2 = np.array([d["kb"][f], prev[1]], np.int32)
        out = np.zeros(1024, np.float32)
        O.ffo_aac_imdct_and_windowing(m1024, m128, wp, ptr(np.ascontiguousarray(d["coeffs"][f]), f32p), ptr(s2, i32p), ptr(k2, i32p),
                                      ptr(saved, f32p), ptr(out, f32p))
        assert np.array_equal(out.view(np.uint32), d["out"][f].view(np.uint32)), f
        prev = (int(d["seq"][f]), int(d["kb"][f]))
    assert np.array_equal(saved.view(np.uint32), d["saved_out"].view(np.uint32))
    O.ffo_mdct_free(m1024); O.ffo_mdct_free(m128)


def test_fdsp_golden():
    O = ffi.oracle()
    d = load("fdsp")
    for op in range(7):
        for n in (1024, 37):
            k = "op%d_n%d_" % (op, n)
            dst, s0 = d[k + "dst"].copy(), d[k + "s0"].copy()
            O.ffo_fdsp(op, ptr(dst, f32p), ptr(s0, f32p), ptr(np.ascontiguousarray(d[k + "s1"]), f32p),
                       ptr(np.ascontiguousarray(d[k + "s2"]), f32p), float(d[k + "mul"][0]), n)
            assert np.array_equal(dst.view(np.uint32), d[k + "out"]) and np.array_equal(s0.view(np.uint32), d[k + "out0"]), (op, n)


def test_fft_golden():
    O = ffi.oracle()
    d = load("fft")
    for len_ in (8, 256, 1024):
        for inv in (0, 1):
            x, want = d["fft%d_%d_in" % (len_, inv)], d["fft%d_%d_out" % (len_, inv)]
            for t in range(x.shape[0]):
                out = np.zeros(2 * len_, np.float32)
                O.ffo_fft_run(inv, len_, ptr(out, f32p), ptr(np.ascontiguousarray(x[t]), f32p))
                assert np.array_equal(out.view(np.uint32), want[t].view(np.uint32)), (len_, inv)
    for len_ in (16, 1024):
        for inv in (0, 1):
            x, want = d["rdft%d_%d_in" % (len_, inv)], d["rdft%d_%d_out" % (len_, inv)]
            for t in range(x.shape[0]):
                out = np.zeros(want.shape[1], np.float32)
                O.ffo_rdft_run(inv, len_, 1.0, ptr(out, f32p), ptr(np.ascontiguousarray(x[t]), f32p))
                assert np.array_equal(out.view(np.uint32), want[t].view(np.uint32)), ("rdft", len_, inv)
    for n in (16, 1024):
        for inv in (0, 1):
            x, want = d["dct%d_%d_in" % (n, inv)], d["dct%d_%d_out" % (n, inv)]
            for t in range(x.shape[0]):
                out = np.zeros(n, np.float32)
                O.ffo_dct_run(inv, n, 1.0, ptr(out, f32p), ptr(np.ascontiguousarray(x[t]), f32p))
                assert np.array_equal(out.view(np.uint32), want[t].view(np.uint32)), ("dct", n, inv)
    for len_ in (16, 1024):
        for mode in (1, 2):
            x, want = d["rdfth%d_%d_in" % (len_, mode)], d["rdfth%d_%d_out" % (len_, mode)]
            for t in range(x.shape[0]):
                out = np.zeros(len_ // 2 + 1, np.float32)
                O.ffo_rdft_half_run(mode, len_, 1.0, ptr(out, f32p), ptr(np.ascontiguousarray(x[t]), f32p))
                assert np.array_equal(out[:want.shape[1]].view(np.uint32), want[t].view(np.uint32)), ("rdft half", len_, mode)


def test_sws_uops_golden():
    """SwsOpBackend row: the oracle's interpreter on the micro-op lists of real conversions == backend_c's committed output"""
    import swsops as S
    O = ffi.oracle()
    g = S.declare(O, "ffo_sws_uops_")
    cases = S.golden_cases(os.path.join(os.path.dirname(__file__), "golden", "sws_uops.npz"))
    assert len(cases) >= 10
    for name, size, lst, src, dst in cases:
        h = C.c_void_p()
        assert g("compile")(lst.uops, lst.n, C.byref(h)) == 0, name
        got, _ = S.run_golden(g("func"), h, g("block_size")(h), lst, size, src, [d.shape for d in dst])
        g("free")(C.byref(h))
        for i, (a, b) in enumerate(zip(got, dst)):
            assert np.array_equal(a, b), (name, size, "plane %d: %d bytes differ" % (i, (a != b).sum()))


# ---- round 2's additions (tests/golden/round2.npz, written by tools/make_golden.py round2 from the reference built in place) ----
def test_round2_mdct_pfa_golden():
    """ff_tx_mdct_pfa_{3,5,7,9}xM: the oracle on the stored inputs == the stored reference outputs, bit for bit"""
    O = ffi.oracle()
    d = load("round2")
    for key in d["mdct_keys"]:
        key = str(key)
        _, inv, scale = key.split("_")
        len_ = int(key.split("_")[0][4:])
        assert O.ffo_mdct_pfa_factor(len_) in (3, 5, 7, 9)
        s = O.ffo_mdct_create(int(inv), len_, float(scale))
        for t in range(d[key + "_in"].shape[0]):
            out = np.zeros(len_, np.float32)
            O.ffo_mdct_run(s, ptr(out, f32p), ptr(np.ascontiguousarray(d[key + "_in"][t]), f32p), 4)
            assert np.array_equal(out.view(np.uint32), d[key + "_out"][t].view(np.uint32)), key
        O.ffo_mdct_free(s)


def test_round2_vp9_loopfilter_sb_golden():
    O = ffi.oracle()
    d = load("round2")
    lim, mblim = np.ascontiguousarray(d["lf_lim"]), np.ascontiguousarray(d["lf_mblim"])
    changed = 0
    for n in range(int(d["lf_n"])):
        bd, row, col = (int(v) for v in d["lf%d_par" % n])
        pl = [d["lf%d_in%d" % (n, k)].copy() for k in range(3)]
        level, mask = np.ascontiguousarray(d["lf%d_level" % n]), np.ascontiguousarray(d["lf%d_mask" % n])
        O.ffo_vp9_loopfilter_sb(bd, 1, 1, ptr(level, u8p), ptr(mask, u8p), row, col,
                                *(C.cast(p.ctypes.data + k * p.strides[0] + k * p.itemsize, u8p) for p, k in zip(pl, (64, 32, 32))),
                                pl[0].strides[0], pl[1].strides[0], ptr(lim, u8p), ptr(mblim, u8p))
        for k in range(3):
            assert np.array_equal(pl[k], d["lf%d_out%d" % (n, k)]), (n, k)
            changed += int((pl[k] != d["lf%d_in%d" % (n, k)]).sum())
    assert changed > 1000


def test_round2_aac_tools_golden():
    """mid/side + intensity stereo, apply_ltp, update_ltp, imdct_and_windowing_960 against the reference's stored outputs"""
    O = ffi.oracle()
    d = load("round2")
    u16p, i8p_ = C.POINTER(C.c_uint16), C.POINTER(C.c_int8)
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
    for k in range(2):
        ng, max_sfb, ms_present = (int(v) for v in d["st%d_par" % k])
        g = {name: np.ascontiguousarray(d["st%d_%s" % (k, name)]) for name in ("group_len", "ms_mask", "band_type0", "band_type1", "sf1", "swb")}
        a0, a1 = d["st%d_in0" % k].copy(), d["st%d_in1" % k].copy()
        O.ffo_aac_apply_mid_side_stereo(ptr(a0, f32p), ptr(a1, f32p), ng, ptr(g["group_len"], u8p), max_sfb, ptr(g["ms_mask"], u8p),
                                        ptr(g["band_type0"], i32p), ptr(g["band_type1"], i32p), ptr(g["swb"], u16p))
        O.ffo_aac_apply_intensity_stereo(ptr(a0, f32p), ptr(a1, f32p), ng, ptr(g["group_len"], u8p), max_sfb, ms_present, ptr(g["ms_mask"], u8p),
                                         ptr(g["band_type1"], i32p), ptr(g["sf1"], f32p), ptr(g["swb"], u16p))
        assert np.array_equal(bits(a0), bits(d["st%d_out0" % k])) and np.array_equal(bits(a1), bits(d["st%d_out1" % k])), k
    win = [np.ascontiguousarray(d["win%d" % k]) for k in range(4)]
    wp = (f32p * 4)(*[ptr(w, f32p) for w in win])
    # apply_ltp
    from test_oracle_vs_ref import aac_tns_filters
    lag, max_sfb, num_swb, tmb = (int(v) for v in d["ltp_par"])
    t = {name: np.ascontiguousarray(d["ltp_tns_" + name]) for name in ("n_filt", "length", "direction", "order", "coef")}
    t.update(num_windows=1, num_swb=num_swb, swb=np.ascontiguousarray(d["ltp_swb"]), tns_max_bands=tmb, max_sfb=max_sfb)
    rec = aac_tns_filters(O, t)
    m = O.ffo_mdct_create(0, 1024, np.float32(-32786.0 * 2 + 36))
    co, pf = d["ltp_in"].copy(), np.zeros(1024, np.float32)
    O.ffo_aac_apply_ltp(m, wp, ptr(co, f32p), ptr(np.ascontiguousarray(d["ltp_state"]), f32p), lag, float(d["ltp_coef"]),
                        ptr(np.ascontiguousarray(d["ltp_used"]), i8p_), ptr(np.ascontiguousarray(d["ltp_seq"]), i32p),
                        ptr(np.ascontiguousarray(d["ltp_kb"]), i32p), max_sfb, ptr(t["swb"], u16p), rec.ctypes.data if len(rec) else None, len(rec),
                        ptr(pf, f32p))
    O.ffo_mdct_free(m)
    assert np.array_equal(bits(pf), bits(d["ltp_pred"])) and np.array_equal(bits(co), bits(d["ltp_out"]))
    assert not np.array_equal(bits(co), bits(d["ltp_in"]))
    # update_ltp
    for k in range(3):
        st = d["ul%d_in" % k].copy()
        seq0, kb0 = (int(v) for v in d["ul%d_par" % k])
        O.ffo_aac_update_ltp(wp, ptr(st, f32p), ptr(np.ascontiguousarray(d["ul%d_buf" % k]), f32p), ptr(np.ascontiguousarray(d["ul%d_saved" % k]), f32p),
                             ptr(np.ascontiguousarray(d["ul%d_output" % k]), f32p), seq0, kb0)
        assert np.array_equal(bits(st), bits(d["ul%d_out" % k])), k
    # imdct_and_windowing_960
    w960 = [np.ascontiguousarray(d["w960_%d" % k]) for k in range(4)]
    wp960 = (f32p * 4)(*[ptr(w, f32p) for w in w960])
    ml, ms = O.ffo_mdct_create(1, 960, np.float32((1.0 / 960) / 32768.0)), O.ffo_mdct_create(1, 120, np.float32((1.0 / 120) / 32768.0))
    saved = d["a960_saved_in"].copy()
    prev = (0, 0)
    for f in range(len(d["a960_seq"])):
        s2 = np.array([d["a960_seq"][f], prev[0]], np.int32); k2 = np.array([d["a960_kb"][f], prev[1]], np.int32)
        out = np.zeros(960, np.float32)
        O.ffo_aac_imdct_and_windowing_len(960, 128, ml, ms, wp960, ptr(np.ascontiguousarray(d["a960_coeffs"][f]), f32p), ptr(s2, i32p),
                                          ptr(k2, i32p), ptr(saved, f32p), ptr(out, f32p))
        assert np.array_equal(bits(out), bits(d["a960_out"][f])), f
        prev = (int(d["a960_seq"][f]), int(d["a960_kb"][f]))
    assert np.array_equal(bits(saved), bits(d["a960_saved_out"]))
    O.ffo_mdct_free(ml); O.ffo_mdct_free(ms)


# ---- round 3 ------------------------------------------------------------------------------------------------------------------
HBD_FMT = {"yuv420p9le": (60, 9, 0), "yuv420p10le": (62, 10, 0), "yuv420p12le": (123, 12, 0), "yuv420p14le": (125, 14, 0),
           "yuv420p16le": (45, 16, 0), "p010le": (158, 10, 1), "p012le": (209, 12, 1), "p016le": (169, 16, 1)}


def _planes(arrs):
    p = (u8p * 4)()
    s = (C.c_int * 4)()
    for i, a in enumerate(arrs):
        p[i] = C.cast(a.ctypes.data, u8p)
        s[i] = a.strides[0]
    return p, s


def test_round3_sws_golden():
    """the scaler above 8 bits, range conversion, full-range and 4:2:2 sources to packed RGB: the oracle on the stored frames, with
    the banks / constants / coefficients libffhip's host side derives (no device needed), == the stored reference outputs"""
    from ffmpeg_amd import swscale as S
    O = ffi.oracle()
    O.ffo_sws_scale_frame_hbd.argtypes = [C.POINTER(ffi.OSwsTables), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(C.c_int),
                                          C.POINTER(u8p), C.POINTER(C.c_int)]
    d = load("round3")
    base = {"yuvj420p": "yuv420p"}
    for k, case in enumerate(d["sws_cases"]):
        sn, sw, sh, dn, dw, dh, flags, sr, dr = str(case).split()
        sw, sh, dw, dh, flags, sr, dr = (int(v) for v in (sw, sh, dw, dh, flags, sr, dr))
        src = [np.ascontiguousarray(d["sws%d_src%d" % (k, i)]) for i in range(3) if "sws%d_src%d" % (k, i) in d]
        want = [d["sws%d_out%d" % (k, i)] for i in range(3) if "sws%d_out%d" % (k, i) in d]
        got = [np.zeros_like(a) for a in want]
        sfmt = HBD_FMT[sn][0] if sn in HBD_FMT else ffi.PIX[sn]
        dfmt = HBD_FMT[dn][0] if dn in HBD_FMT else ffi.PIX[dn]
        hb = sn in HBD_FMT or dn in HBD_FMT
        ht = S.HostTables(sw, sh, sfmt, dw, dh, dfmt, flags, ranges=(sr, dr) if hb and sr != dr else None)
        sp, ss = _planes(src)
        gp, gs = _planes(got)
        if hb:
            sd, sl = HBD_FMT[sn][1:] if sn in HBD_FMT else (8, 0)
            dd, dl = HBD_FMT[dn][1:] if dn in HBD_FMT else (8, 0)
            t = ffi.make_otables(sw, sh, sfmt, dw, dh, dfmt, flags, ht.banks(), ht.coeffs(), ranges=(sr, dr), dst_depth=dd)
            assert O.ffo_sws_scale_frame_hbd(C.byref(t), sd, sl, dd, dl, sp, ss, gp, gs) == 0, case
        else:
            rgb = dn in ("rgb24", "bgra")
            t = ffi.make_otables(sw, sh, ffi.PIX[base.get(sn, sn)], dw, dh, ffi.PIX[base.get(dn, dn)], flags, ht.banks(), ht.coeffs(),
                                 ranges=None if rgb else (sr, dr))
            if ht.unscaled_yuv2rgb:
                co = ht.coeffs()
                luts = ffi.OLuts()
                kk = ffi.OYuv2RgbCoeffs(*[co[n] for n in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
                O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(kk))
                if sn == "yuv422p":   # the table converter's 4:2:2 form (YUV422FUNC, yuv2rgb.c:238-320): a chroma row per luma row
                    O.ffo_yuv2rgb_unscaled(C.byref(luts), sw, sp, ss, 0, sh, gp, gs, ffi.RGB_LAYOUT[ffi.PIX[dn]], 1, 0)
                else:
                    O.ffo_yuv420p_to_rgb24(C.byref(luts), sw, sp, ss, 0, sh, ptr(got[0]), got[0].strides[0], ffi.RGB_LAYOUT[ffi.PIX[dn]])
            else:
                assert O.ffo_sws_scale_frame(C.byref(t), sp, ss, gp, gs) == dh, case
        for a, b in zip(got, want):
            assert np.array_equal(a, b), (str(case), int((a != b).sum()))


def test_round3_fft_golden():
    """ff_tx_fft_pfa (120 / 960 / 96 / 1280) and FFT-4096: the oracle on the stored inputs == the stored reference outputs, bit for bit"""
    O = ffi.oracle()
    d = load("round3")
    for key in d["fft_keys"]:
        key = str(key)
        len_, inv = int(key[3:].split("_")[0]), int(key.split("_")[1])
        for t in range(d[key + "_in"].shape[0]):
            out = np.zeros(2 * len_, np.float32)
            O.ffo_fft_run(inv, len_, ptr(out, f32p), ptr(np.ascontiguousarray(d[key + "_in"][t]), f32p))
            assert np.array_equal(out.view(np.uint32), d[key + "_out"][t].view(np.uint32)), key


def test_round3_h264_hbd_golden():
    """h264dsp / qpel / chroma / weight at 10 and 12 bits: the oracle's *_bd functions on the stored inputs == the stored outputs"""
    O = ffi.oracle()
    O.ffo_h264_idct_bd.argtypes = [C.c_int, C.c_int, u8p, i16p, C.c_ssize_t]
    O.ffo_h264_loop_filter_bd.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.POINTER(C.c_int8)]
    O.ffo_h264_qpel_bd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t]
    O.ffo_h264_chroma_mc_bd.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
    O.ffo_h264_biweight_bd.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    d = load("round3")

    def pat(a, row, col):
        return C.cast(a.ctypes.data + (row * a.shape[1] + col) * a.itemsize, u8p)
    for bd in (10, 12):
        for kind in (0, 1):
            c = np.ascontiguousarray(d["h%d_idct%d_c" % (bd, kind)])
            pic = np.ascontiguousarray(d["h%d_idct%d_in" % (bd, kind)])
            O.ffo_h264_idct_bd(bd, kind, pat(pic, 2, 5), C.cast(c.ctypes.data, i16p), pic.strides[0])
            assert np.array_equal(pic, d["h%d_idct%d_out" % (bd, kind)]) and not c.any(), (bd, kind)
        tc = np.array([0, 1, 3, -1], np.int8)
        for kind, inner in ((0, 4), (5, 4), (2, 2), (7, 2)):
            pic = np.ascontiguousarray(d["h%d_lf%d_in" % (bd, kind)])
            O.ffo_h264_loop_filter_bd(bd, kind, inner, pat(pic, 12, 12), pic.strides[0], 40, 9, tc.ctypes.data_as(C.POINTER(C.c_int8)))
            assert np.array_equal(pic, d["h%d_lf%d_out" % (bd, kind)]), (bd, kind)
        src = np.ascontiguousarray(d["h%d_mc_src" % bd])
        pic = np.ascontiguousarray(d["h%d_qpel_in" % bd])
        O.ffo_h264_qpel_bd(bd, 1, 0, 10, pat(pic, 6, 8), pat(src, 6, 8), pic.strides[0])
        assert np.array_equal(pic, d["h%d_qpel_out" % bd]), bd
        pic = np.ascontiguousarray(d["h%d_chroma_in" % bd])
        O.ffo_h264_chroma_mc_bd(bd, 0, 8, pat(pic, 2, 4), pat(src, 2, 4), pic.strides[0], 8, 3, 5)
        assert np.array_equal(pic, d["h%d_chroma_out" % bd]), bd
        pic = np.ascontiguousarray(d["h%d_bw_in" % bd])
        O.ffo_h264_biweight_bd(bd, 16, pat(pic, 1, 4), pat(src, 1, 4), pic.strides[0], 16, 5, 37, -21, 9)
        assert np.array_equal(pic, d["h%d_bw_out" % bd]), bd


def test_round4_alpha_both_sides_golden():
    """planar YUVA -> planar YUVA: the reference's four planes == the oracle's base conversion, and the oracle's luma of the same conversion
    with A in Y's place for the alpha plane (lum_h_scale / lum_planar_vscale on plane 3) — what libffhip's second pass computes"""
    from ffmpeg_amd import swscale as S
    d = load("round4")
    base = {33: 0, 78: 4, 79: 5}                        # yuva420p / 422p / 444p -> yuv420p / 422p / 444p
    for k in range(int(d["a_n"])):
        sf, sw, sh, df, dw, dh, flags = (int(v) for v in d["a%d_meta" % k])
        src = [np.ascontiguousarray(d["a%d_src%d" % (k, p)]) for p in range(4)]
        ht = S.HostTables(sw, sh, sf, dw, dh, df, flags)
        assert ht.t.dst_alpha_fill == 2 and ht.t.srcFormat == base[sf] and ht.t.dstFormat == base[df]
        t = ffi.make_otables(sw, sh, base[sf], dw, dh, base[df], flags, ht.banks(), ht.coeffs(), full=ht.full())
        for planes, wanted in ((src[:3], (0, 1, 2)), ([src[3], src[1], src[2]], (3,))):
            out = ffi.alloc_frame(base[df], dw, dh)
            sp, ss = ffi.planes(planes)
            dp, ds = ffi.planes(out)
            assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
            for j, p in enumerate(wanted):
                assert np.array_equal(out[j if p < 3 else 0], d["a%d_dst%d" % (k, p)]), (k, p)


def test_round4_vp9_loopfilter_422_440_golden():
    """the superblock filter with the two sub-sampling shifts apart: oracle == the reference's output, and the product's tables
    (ffhip_vp9_lf_sb_tables' luma part + ffhip_vp9_lf_sb_ctables) executed in kernel order leave the same samples"""
    import vp9_lf_gen as VG
    from ffmpeg_amd import _lib
    L = _lib.lib()
    O = ffi.oracle()
    d = load("round4")
    lim, mblim = np.ascontiguousarray(d["lf_lim"]), np.ascontiguousarray(d["lf_mblim"])
    changed = 0
    for n in range(int(d["lf_n"])):
        bd, ss_h, ss_v, row, col = (int(v) for v in d["lf%d_par" % n])
        cw, chh = 64 >> ss_h, 64 >> ss_v
        pos = ((64, 64), (chh, cw), (chh, cw))
        level, mask = np.ascontiguousarray(d["lf%d_level" % n]), np.ascontiguousarray(d["lf%d_mask" % n])
        a = [d["lf%d_in%d" % (n, k)].copy() for k in range(3)]
        b = [p.copy() for p in a]
        addr = lambda pl: [p.ctypes.data + r * p.strides[0] + c * p.itemsize for p, (r, c) in zip(pl, pos)]
        O.ffo_vp9_loopfilter_sb(bd, ss_h, ss_v, ptr(level, u8p), ptr(mask, u8p), row, col, *(C.cast(x, u8p) for x in addr(a)), a[0].strides[0],
                                a[1].strides[0], ptr(lim, u8p), ptr(mblim, u8p))
        f = np.zeros((), VG.FILTER_DT)
        f["level"], f["mask"] = level, mask
        fb = np.frombuffer(f.tobytes(), np.uint8).copy()
        tab, ctab = np.zeros(320, np.uint32), np.zeros(128, np.uint32)
        assert L.ffhip_vp9_lf_sb_tables(tab.ctypes.data, fb.ctypes.data, row, col, ss_h, ss_v, lim.ctypes.data, mblim.ctypes.data) == 0
        assert L.ffhip_vp9_lf_sb_ctables(ctab.ctypes.data, fb.ctypes.data, row, col, ss_h, ss_v, lim.ctypes.data, mblim.ctypes.data) == 0
        VG.run_tables(O, tab, bd, addr(b), [b[0].strides[0], b[1].strides[0]])
        VG.run_ctables(O, ctab, bd, addr(b)[1:], b[1].strides[0], ss_h, ss_v)
        for k in range(3):
            assert np.array_equal(a[k], d["lf%d_out%d" % (n, k)]), (n, k)
            assert np.array_equal(b[k], d["lf%d_out%d" % (n, k)]), (n, k, "tables")
            changed += int((a[k] != d["lf%d_in%d" % (n, k)]).sum())
    assert changed > 500
