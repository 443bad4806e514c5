#!/usr/bin/env python3
"""tools/make_golden.py — writes tests/golden/*.npz: seeded inputs and the outputs of the REAL reference
(oracle/_ref/libffref.so, compiled from /root/reference by oracle/refbuild/Makefile) for every row of the hot
path.  Run in the build container (the reference does not travel); the fixtures do, and pin both the oracle
(tests/test_golden.py, CPU) and the HIP kernels (tests/test_gpu_golden.py) where the reference is absent."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffi  # noqa: E402
from ffi import PIX, ptr, u8p, i8p, i16p, i32p, f32p  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
R = ffi.ref()


def at(a, off):
    return C.cast(a.ctypes.data + int(off), u8p)


def sws():
    d = {}
    rng = np.random.default_rng(1001)
    cases = [("yuv420p", 64, 16, "rgb24", 64, 16, 4), ("yuv420p", 62, 8, "bgr24", 62, 8, 4),
             ("nv12", 96, 54, "nv12", 192, 108, 4), ("nv21", 64, 40, "yuv420p", 160, 88, 4),
             ("yuv420p", 64, 48, "nv12", 128, 96, 4), ("nv12", 80, 48, "nv12", 48, 32, 4),
             ("yuv420p", 48, 32, "rgb24", 96, 64, 4), ("yuv420p", 48, 32, "bgr24", 48, 32, 4 | 0x40000 | 0x80000),
             # appended later (the earlier cases keep their seeded inputs): 32-bit packed targets
             ("yuv420p", 64, 16, "bgra", 64, 16, 4), ("yuv420p", 62, 8, "argb", 62, 8, 4),
             ("nv12", 48, 32, "rgba", 96, 64, 4), ("yuv420p", 48, 32, "abgr", 48, 32, 4 | 0x40000 | 0x80000),
             # round 2: 4:2:2 / 4:4:4 planar on either side
             ("yuv422p", 64, 40, "yuv422p", 160, 88, 4), ("yuv444p", 48, 32, "yuv420p", 96, 64, 4), ("nv12", 64, 48, "yuv444p", 40, 30, 4)]
    for i, (sf, sw, sh, df, dw, dh, fl) in enumerate(cases):
        src = ffi.alloc_frame(PIX[sf], sw, sh, rng)
        ctx = R.ffref_sws_create(sw, sh, PIX[sf], dw, dh, PIX[df], fl, 1)
        dst = ffi.alloc_frame(PIX[df], dw, dh)
        sp, ss = ffi.planes(src)
        dp, ds = ffi.planes(dst)
        assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, dp, ds) == dh
        d["c%d_meta" % i] = np.array([PIX[sf], sw, sh, PIX[df], dw, dh, fl, R.ffref_sws_is_unscaled(ctx)], np.int64)
        for p, a in enumerate(src):
            d["c%d_src%d" % (i, p)] = a
        for p, a in enumerate(dst):
            d["c%d_dst%d" % (i, p)] = a
        if not R.ffref_sws_is_unscaled(ctx):
            for name, (f, pos, fs, n) in ffi.ref_tables(ctx).items():
                d["c%d_%s_f" % (i, name)] = f
                d["c%d_%s_p" % (i, name)] = pos
                d["c%d_%s_s" % (i, name)] = np.array([fs, n], np.int32)
        R.ffref_sws_free(ctx)
    d["ncases"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(OUT, "sws.npz"), **d)


def h264():
    d = {}
    rng = np.random.default_rng(1002)
    stride = 32
    for which, size in ((0, 4), (1, 8), (2, 4), (3, 8)):
        n = 24
        coefs = rng.integers(-1500, 1500, (n, size * size)).astype(np.int16)
        coefs[::5] = rng.integers(-32768, 32768, coefs[::5].shape).astype(np.int16)
        coefs[1::4, 1:] = 0
        dst = rng.integers(0, 256, (n, size, stride), dtype=np.uint8)
        od, oc = dst.copy(), coefs.copy()
        for i in range(n):
            R.ffref_h264_idct(which, ptr(od[i]), ptr(oc[i], i16p), stride)
        d["idct%d_in_dst" % which], d["idct%d_in_coef" % which] = dst, coefs
        d["idct%d_out_dst" % which], d["idct%d_out_coef" % which] = od, oc
    lf_in, lf_out, lf_par = [], [], []
    for which in range(8):
        for alpha, beta, t in ((255, 18, 13), (127, 16, 6), (45, 10, 3), (17, 6, 1), (9, 3, 0)):
            base = int(rng.integers(30, 220))
            img = np.clip(base + rng.integers(-7, 8, (32, 32)), 0, 255).astype(np.uint8)
            tc0 = np.array([t, -1, 0, max(t - 1, 0)], np.int8)
            o = img.copy()
            R.ffref_h264_loop_filter(which, at(o, 8 * 32 + 8), 32, alpha, beta, ptr(tc0.copy(), i8p))
            lf_in.append(img); lf_out.append(o); lf_par.append([which, alpha, beta] + list(tc0))
    d["lf_in"], d["lf_out"], d["lf_par"] = np.stack(lf_in), np.stack(lf_out), np.array(lf_par, np.int32)
    q_out, q_par = [], []
    src = rng.integers(0, 256, (32, 64), dtype=np.uint8)
    dst = rng.integers(0, 256, (32, 64), dtype=np.uint8)
    for avg in (0, 1):
        for size_idx in range(3):
            for mc in range(16):
                o = dst.copy()
                R.ffref_h264_qpel(avg, size_idx, mc, at(o, 6 * 64 + 8), at(src, 6 * 64 + 8), 64)
                assert np.array_equal(o[:6], dst[:6]) and np.array_equal(o[22:], dst[22:])
                q_out.append(o[6:22, 8:24].copy()); q_par.append([avg, size_idx, mc])
    d["qpel_src"], d["qpel_dst"], d["qpel_out"], d["qpel_par"] = src, dst, np.stack(q_out), np.array(q_par, np.int32)
    # chroma 1/8-pel MC and weighted prediction
    csrc = rng.integers(0, 256, (24, 32), dtype=np.uint8)
    cdst = rng.integers(0, 256, (24, 32), dtype=np.uint8)
    c_out, c_par = [], []
    for avg in (0, 1):
        for idx in range(3):
            for (x, y) in ((0, 0), (5, 0), (0, 3), (7, 7), (2, 6), (4, 4)):
                o = cdst.copy()
                R.ffref_h264_chroma(avg, idx, at(o, 2 * 32 + 8), at(csrc, 2 * 32 + 8), 32, 8, x, y)
                c_out.append(o[2:10, 8:16].copy()); c_par.append([avg, idx, x, y])
    d["chroma_src"], d["chroma_dst"], d["chroma_out"], d["chroma_par"] = csrc, cdst, np.stack(c_out), np.array(c_par, np.int32)
    w_out, w_par = [], []
    for idx in range(4):
        for ld, wt, ws, of in ((0, 1, 1, 0), (5, 37, -12, 9), (7, -128, 127, -128), (2, 127, 127, 127), (6, 64, 64, 0)):
            o = cdst.copy()
            R.ffref_h264_weight(idx, at(o, 2 * 32 + 8), 32, 16, ld, wt, of)
            w_out.append(o[2:18, 8:24].copy()); w_par.append([0, idx, ld, wt, ws, of])
            o = cdst.copy()
            R.ffref_h264_biweight(idx, at(o, 2 * 32 + 8), at(csrc, 2 * 32 + 8), 32, 16, ld, wt, ws, of)
            w_out.append(o[2:18, 8:24].copy()); w_par.append([1, idx, ld, wt, ws, of])
    d["weight_out"], d["weight_par"] = np.stack(w_out), np.array(w_par, np.int32)
    np.savez_compressed(os.path.join(OUT, "h264.npz"), **d)


def h264_misc():
    """idct_add8 / dc_dequant_idct / add_pixels_clear: inputs + the reference's outputs (tests/golden/h264_misc.npz)"""
    from test_oracle_vs_ref import h264_misc_cases, h264_misc_apply
    R = ffi.ref()
    bo, stride, cases = h264_misc_cases(np.random.default_rng(0xF0F0_0013), n=12)
    d = {"bo": bo, "stride": np.int32(stride), "n": np.int32(len(cases))}
    names = ("cb", "cr", "blocks", "out", "dc", "cdc", "px4", "res4", "px8", "res8")
    for i, k in enumerate(cases):
        for key, v in k.items():
            d["in%d_%s" % (i, key)] = np.asarray(v)
        for nm, v in zip(names, h264_misc_apply(R, "ffref", bo, stride, k)):
            d["out%d_%s" % (i, nm)] = v
    np.savez_compressed(os.path.join(OUT, "h264_misc.npz"), **d)


def me():
    d = {}
    rng = np.random.default_rng(1003)
    a = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-20, 21, (64, 64)), 0, 255).astype(np.uint8)
    pos, vals = [], []
    for _ in range(40):
        y1, x1, y2, x2 = [int(v) for v in rng.integers(0, 40, 4)]
        x1 &= ~15
        pa, pb = at(a, y1 * 64 + x1), at(b, y2 * 64 + x2)
        pos.append([y1, x1, y2, x2])
        vals.append([R.ffref_me_cmp(0, 0, pa, pb, 64, 16), R.ffref_me_cmp(0, 0, pa, pb, 64, 8), R.ffref_me_cmp(0, 1, pa, pb, 64, 8),
                     R.ffref_me_cmp(1, 0, pa, pb, 64, 16), R.ffref_me_cmp(1, 0, pa, pb, 64, 8), R.ffref_me_cmp(1, 1, pa, pb, 64, 8)])
    d["cmp_a"], d["cmp_b"], d["cmp_pos"], d["cmp_vals"] = a, b, np.array(pos, np.int32), np.array(vals, np.int32)
    w, h = 96, 64
    ref_img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    cur = np.roll(ref_img, (2, -3), (0, 1)).copy()
    cur[:16, :16] = ref_img[:16, :16]
    cur[20:, 40:] = rng.integers(0, 4, cur[20:, 40:].shape, dtype=np.uint8)
    for Rr in (3, 7):
        mv, cost = [], []
        for by in range(h // 16):
            for bx in range(w // 16):
                m = np.zeros(2, np.int32)
                c = R.ffref_me_search_esa(ptr(cur), ptr(ref_img), w, w, h, 16, Rr, bx * 16, by * 16, ptr(m, i32p))
                mv.append(m.copy()); cost.append(c)
        d["esa_mv_r%d" % Rr], d["esa_cost_r%d" % Rr] = np.array(mv, np.int32), np.array(cost, np.uint64)
    d["esa_cur"], d["esa_ref"] = cur, ref_img
    np.savez_compressed(os.path.join(OUT, "me.npz"), **d)


def tx():
    d = {}
    rng = np.random.default_rng(1004)
    for len_ in (64, 1024):
        for inv, scale in ((0, 1.0), (0, 32768.0), (1, 1.0 / len_)):
            x = rng.uniform(-1, 1, (3, len_ if inv else 2 * len_)).astype(np.float32)
            rc = R.ffref_tx_create(1, inv, len_, scale, 0)
            out = np.zeros((3, len_), np.float32)
            for t in range(3):
                R.ffref_tx_run(rc, ptr(out[t], f32p), ptr(x[t].copy(), f32p), 4)
            R.ffref_tx_free(rc)
            key = "mdct%d_%d_%r" % (len_, inv, scale)
            d[key + "_in"], d[key + "_out"] = x, out
    # 15xM prime-factor lengths (CELT 120 / 960, AAC-960 1920), own generator so that the entries above stay as they were
    rng = np.random.default_rng(1006)
    for len_ in (120, 960, 1920):
        for inv, scale in ((0, 1.0), (1, 1.0 / len_)):
            x = rng.uniform(-1, 1, (3, len_ if inv else 2 * len_)).astype(np.float32)
            rc = R.ffref_tx_create(1, inv, len_, scale, 0)
            out = np.zeros((3, len_), np.float32)
            for t in range(3):
                R.ffref_tx_run(rc, ptr(out[t], f32p), ptr(x[t].copy(), f32p), 4)
            R.ffref_tx_free(rc)
            key = "mdct%d_%d_%r" % (len_, inv, scale)
            d[key + "_in"], d[key + "_out"] = x, out
    d["keys"] = np.array(sorted({k.rsplit("_", 1)[0] for k in d}))
    np.savez_compressed(os.path.join(OUT, "tx.npz"), **d)


def fft():
    """AV_TX_FLOAT_FFT, power-of-two, both directions: 3 transforms per (len, inv)"""
    d = {}
    rng = np.random.default_rng(1005)
    for len_ in (8, 256, 1024):
        for inv in (0, 1):
            x = rng.uniform(-1, 1, (3, 2 * len_)).astype(np.float32)
            rc = R.ffref_tx_create(0, inv, len_, 1.0, 0)
            out = np.zeros((3, 2 * len_), np.float32)
            for t in range(3):
                R.ffref_tx_run(rc, ptr(out[t], f32p), ptr(x[t].copy(), f32p), 8)
            R.ffref_tx_free(rc)
            d["fft%d_%d_in" % (len_, inv)], d["fft%d_%d_out" % (len_, inv)] = x, out
    # AV_TX_FLOAT_RDFT (type 6): r2c takes len reals -> len/2 + 1 complex, c2r the other way round
    rng = np.random.default_rng(1007)
    for len_ in (16, 1024):
        for inv in (0, 1):
            x = rng.uniform(-1, 1, (3, len_ + 2 if inv else len_)).astype(np.float32)
            if inv:
                x[:, 1] = x[:, -1] = 0
            rc = R.ffref_tx_create(6, inv, len_, 1.0, 0)
            out = np.zeros((3, len_ if inv else len_ + 2), np.float32)
            for t in range(3):
                R.ffref_tx_run(rc, ptr(out[t], f32p), ptr(x[t].copy(), f32p), 4)
            R.ffref_tx_free(rc)
            d["rdft%d_%d_in" % (len_, inv)], d["rdft%d_%d_out" % (len_, inv)] = x, out
    # AV_TX_FLOAT_DCT (type 9): DCT-II of n reals; DCT-III initialised with n / 2 (ff_tx_dct_init doubles it), input padded by 2
    rng = np.random.default_rng(1009)
    for n in (16, 1024):
        for inv in (0, 1):
            x = np.zeros((3, n + 2), np.float32)
            x[:, :n] = rng.uniform(-1, 1, (3, n)).astype(np.float32)
            rc = R.ffref_tx_create(9, inv, n >> inv, 1.0, 0)
            out = np.zeros((3, n + 2), np.float32)
            for t in range(3):
                R.ffref_tx_run(rc, ptr(out[t], f32p), ptr(x[t].copy(), f32p), 4)
            R.ffref_tx_free(rc)
            d["dct%d_%d_in" % (n, inv)], d["dct%d_%d_out" % (n, inv)] = x[:, :n].copy(), out[:, :n].copy()
    # the half-complex forward RDFTs (flags AV_TX_REAL_TO_REAL = 1 << 3 / AV_TX_REAL_TO_IMAGINARY = 1 << 4): len / 2 + 1 real parts resp.
    # len / 2 imaginary parts; the reference's FFT lands in dst first, hence the wide buffer
    rng = np.random.default_rng(1011)
    for len_ in (16, 1024):
        for mode in (1, 2):
            x = rng.uniform(-1, 1, (3, len_)).astype(np.float32)
            rc = R.ffref_tx_create(6, 0, len_, 1.0, 1 << (2 + mode))
            nout = len_ // 2 + (mode == 1)
            out = np.zeros((3, len_ + 2), np.float32)
            for t in range(3):
                R.ffref_tx_run(rc, ptr(out[t], f32p), ptr(x[t].copy(), f32p), 4)
            R.ffref_tx_free(rc)
            d["rdfth%d_%d_in" % (len_, mode)], d["rdfth%d_%d_out" % (len_, mode)] = x, out[:, :nout].copy()
    np.savez_compressed(os.path.join(OUT, "fft.npz"), **d)


def hevc():
    """HEVC inverse transforms: per size 24 blocks (dense / small / sparse / saturating) x col_limits, DC, 4x4 luma DST,
    add_residual on a padded picture"""
    rng = np.random.default_rng(9)
    d = {}
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        blocks, limits = [], []
        for t in range(24):
            kind = t % 4
            if kind == 0:
                c = rng.integers(-32768, 32768, (n, n))
            elif kind == 1:
                c = rng.integers(-512, 512, (n, n))
            elif kind == 2:
                c = np.zeros((n, n), np.int64)
                k = int(rng.integers(1, n + 1))
                c[:k, :k] = rng.integers(-2048, 2048, (k, k))
            else:
                c = rng.choice(np.array([-32768, 32767, 0, 1, -1]), (n, n))
            blocks.append(c.astype(np.int16))
            limits.append(int(rng.integers(0, 2 * n + 6)) if t % 5 else 1000)
        blocks = np.stack(blocks)
        d["in%d" % lg] = blocks
        d["lim%d" % lg] = np.array(limits, np.int32)
        out = blocks.copy()
        for t in range(24):
            R.ffref_hevc_idct(lg - 2, ptr(out[t], i16p), limits[t])
        d["idct%d" % lg] = out
        out = blocks.copy()
        for t in range(24):
            R.ffref_hevc_idct_dc(lg - 2, ptr(out[t], i16p))
        d["dc%d" % lg] = out
        pic = rng.integers(0, 256, (24, n + 2, 48), dtype=np.uint8)
        d["pic%d" % lg] = pic
        o = pic.copy()
        for t in range(24):
            R.ffref_hevc_add_residual(lg - 2, at(o[t], 48 + 5), ptr(blocks[t], i16p), 48)
        d["add%d" % lg] = o
    out = d["in2"].copy()
    for t in range(24):
        R.ffref_hevc_transform_4x4_luma(ptr(out[t], i16p))
    d["dst4"] = out
    # loop filters: 160 calls on 16x16 neighbourhoods, par = which(0 h_luma 1 v_luma 2 h_chroma 3 v_chroma), beta, tc0, tc1, no_p0/1, no_q0/1
    from test_oracle_vs_ref import hevc_lf_case
    bufs, pars = [], []
    for rep in range(160):
        buf, beta, tc, no_p, no_q = hevc_lf_case(rng, rep % 4 != 0)
        bufs.append(buf)
        pars.append([rep % 4, beta, tc[0], tc[1], no_p[0], no_p[1], no_q[0], no_q[1]])
    d["lf_in"], d["lf_par"] = np.stack(bufs), np.array(pars, np.int32)
    o = d["lf_in"].copy()
    for i, (which, beta, t0, t1, p0, p1, q0, q1) in enumerate(pars):
        off = 4 * 16 + 8 if which & 1 else 8 * 16 + 4
        R.ffref_hevc_loop_filter(int(which), at(o[i], off), 16, int(beta), ptr(np.array([t0, t1], np.int32), i32p),
                                 ptr(np.array([p0, p1], np.uint8)), ptr(np.array([q0, q1], np.uint8)))
    d["lf_out"] = o
    # SAO: 24 blocks, par = edge, cls, w, h, off0..4; source = padded 192-byte rows (block at row 1, col 1)
    srcs, pars, outs = [], [], []
    for rep in range(24):
        w = int(rng.choice([8, 16, 32, 48, 64])); h = int(rng.choice([8, 16, 32]))
        src = rng.integers(0, 256, (34, 192), dtype=np.uint8)
        if rep % 3 == 0:
            src[:] = np.clip(128 + rng.integers(-6, 7, src.shape), 0, 255)
        off = rng.integers(-7, 8, 5).astype(np.int16)
        edge = rep & 1
        cls = int(rng.integers(0, 4 if edge else 32))
        if edge:
            off[0] = 0
        dst = np.zeros((32, 64), np.uint8)
        idx = {8: 0, 16: 1, 32: 2, 48: 3, 64: 4}[w]
        if edge:
            R.ffref_hevc_sao_edge(idx, ptr(dst), at(src, 193), 64, ptr(off, i16p), cls, w, h)
        else:
            R.ffref_hevc_sao_band(idx, ptr(dst), at(src, 193), 64, 192, ptr(off, i16p), cls, w, h)
        srcs.append(src); outs.append(dst); pars.append([edge, cls, w, h] + [int(v) for v in off])
    d["sao_src"], d["sao_out"], d["sao_par"] = np.stack(srcs), np.stack(outs), np.array(pars, np.int32)
    # uni-directional MC: one 96x96 reference, 64 blocks, par = chroma, w, h, mx, my, y0, x0; outputs: int16 [64x64] and pixels [64x64]
    mref = rng.integers(0, 256, (96, 96), dtype=np.uint8)
    widths = [2, 4, 6, 8, 12, 16, 24, 32, 48, 64]
    pars, o16, o8 = [], [], []
    for rep in range(64):
        chroma = rep & 1
        w = widths[rep % 10]; h = int(rng.choice([2, 4, 8, 16]))
        mx, my = (int(v) for v in rng.integers(0, 8 if chroma else 4, 2))
        y0, x0 = int(rng.integers(4, 96 - h - 5)), int(rng.integers(4, 96 - w - 5))
        a16, a8 = np.zeros((64, 64), np.int16), np.zeros((64, 64), np.uint8)
        R.ffref_hevc_mc(chroma, 0, a16.ctypes.data, 0, at(mref, y0 * 96 + x0), 96, h, mx, my, w)
        R.ffref_hevc_mc(chroma, 1, a8.ctypes.data, 64, at(mref, y0 * 96 + x0), 96, h, mx, my, w)
        pars.append([chroma, w, h, mx, my, y0, x0]); o16.append(a16); o8.append(a8)
    d["mc_ref"], d["mc_par"], d["mc_out16"], d["mc_out8"] = mref, np.array(pars, np.int32), np.stack(o16), np.stack(o8)
    # weighted / bi-directional MC on the same reference: par = chroma, mode (2 uni_w, 3 bi, 4 bi_w), w, h, mx, my, y0, x0, denom, wx0, wx1, ox;
    # one shared 64x64 src2 (the other list's 14-bit intermediates)
    wrng = np.random.default_rng(111)
    src2 = wrng.integers(-8192, 16384, (64, 64)).astype(np.int16)
    pars, outs = [], []
    for rep in range(60):
        chroma, mode = rep & 1, 2 + (rep // 2) % 3
        w = widths[rep % 10]; h = int(wrng.choice([2, 4, 8, 16]))
        mx, my = (int(v) for v in wrng.integers(0, 8 if chroma else 4, 2))
        y0, x0 = int(wrng.integers(4, 96 - h - 5)), int(wrng.integers(4, 96 - w - 5))
        if rep % 3 == 0:
            den, wx0, wx1, ox = int(wrng.choice([0, 7, 12])), int(wrng.choice([0, 128, 255])), int(wrng.choice([0, 128, 255])), int(wrng.choice([0, 255]))
        else:
            den = int(wrng.integers(0, 8))
            wx0, wx1, ox = (1 << den) + int(wrng.integers(-128, 128)), (1 << den) + int(wrng.integers(-128, 128)), int(wrng.integers(-256, 255))
        a8 = np.zeros((64, 64), np.uint8)
        R.ffref_hevc_mc_w(chroma, mode, ptr(a8), 64, at(mref, y0 * 96 + x0), 96, ptr(src2, i16p), h, den, wx0, wx1, ox, mx, my, w)
        pars.append([chroma, mode, w, h, mx, my, y0, x0, den, wx0, wx1, ox]); outs.append(a8)
    d["mcw_src2"], d["mcw_par"], d["mcw_out"] = src2, np.array(pars, np.int32), np.stack(outs)
    np.savez_compressed(os.path.join(OUT, "hevc.npz"), **d)


def vp9():
    """VP9 itxfm_add: per (tx, txtp) 6 blocks (dense / sparse / dc-only / large / wrap-around / sparse); outputs: picture and block"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_vs_ref import vp9_block
    rng = np.random.default_rng(13)
    d = {}
    for tx in range(5):
        n = 4 if tx == 4 else 4 << tx
        blks, dsts, outs, oblk, par = [], [], [], [], []
        for txtp in range(4):
            for rep in range(6):
                kind = rep % 5
                blk = vp9_block(rng, n, kind)
                eob = 1 if kind == 2 else n * n
                dst = rng.integers(0, 256, (n, n), dtype=np.uint8)
                o, b = dst.copy(), blk.copy()
                R.ffref_vp9_itxfm_add(tx, txtp, ptr(o), n, ptr(b, i16p), eob)
                blks.append(blk); dsts.append(dst); outs.append(o); oblk.append(b); par.append([txtp, eob])
        d["tx%d_blk" % tx], d["tx%d_dst" % tx] = np.stack(blks), np.stack(dsts)
        d["tx%d_out" % tx], d["tx%d_oblk" % tx], d["tx%d_par" % tx] = np.stack(outs), np.stack(oblk), np.array(par, np.int32)
    # motion compensation: one 96x96 reference, 80 blocks, par = filter, avg, w, h, mx, my, y0, x0; dst in / out [64x64]
    mref = rng.integers(0, 256, (96, 96), dtype=np.uint8)
    pars, dout = [], []
    base = rng.integers(0, 256, (64, 64), dtype=np.uint8)    # every call starts from this destination (avg reads it)
    for rep in range(80):
        f, avg = rep % 4, (rep // 4) & 1
        w = [4, 8, 16, 32, 64][rep % 5]; h = int(rng.choice([2, 4, 8, 16]))
        mx, my = (int(v) for v in rng.integers(0, 16, 2))
        if rep % 6 == 0:
            mx = 0
        if rep % 9 == 0:
            my = 0
        y0, x0 = int(rng.integers(4, 96 - h - 5)), int(rng.integers(4, 96 - w - 5)) if w < 64 else int(rng.integers(4, 27))
        a = base.copy()
        R.ffref_vp9_mc(f, avg, ptr(a), 64, at(mref, y0 * 96 + x0), 96, w, h, mx, my)
        assert np.array_equal(a[h:], base[h:]) and np.array_equal(a[:, w:], base[:, w:])
        a[h:] = 0; a[:, w:] = 0                               # only the block is stored
        dout.append(a); pars.append([f, avg, w, h, mx, my, y0, x0])
    d["mc_ref"], d["mc_par"], d["mc_in"], d["mc_out"] = mref, np.array(pars, np.int32), base, np.stack(dout)
    np.savez_compressed(os.path.join(OUT, "vp9.npz"), **d)


def h264pred():
    """H264PredContext: one 136x200 picture; per batch kind a grid of independent blocks (every mode x flag combination), the
    reference's output blocks and, for the lossless kinds, the coefficients in"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_vs_ref import H264_PRED_KINDS, h264_pred_grid, h264_pred_apply
    rng = np.random.default_rng(14)
    pic = rng.integers(0, 256, (136, 200), dtype=np.uint8)
    pic[:, 100:] = np.clip(np.add.outer(np.arange(136) * 3, np.arange(100) * -2) + 120 + rng.integers(-4, 5, (136, 100)), 0, 255)
    pic[64:, :60] = rng.choice(np.array([0, 255], np.uint8), (72, 60))
    d = {"pic": pic}
    for kind, (n, nmodes, _) in enumerate(H264_PRED_KINDS):
        recs = h264_pred_grid(rng, kind, 136, 200, count=min(nmodes * 4, 64))
        coeffs = None
        if kind >= 4:
            coeffs = rng.integers(-300, 301, len(recs) * n * n).astype(np.int16)
            coeffs[:n * n] = rng.choice(np.array([-32768, 32767], np.int16), n * n)
            d["k%d_coef" % kind] = coeffs.copy()
        out = pic.copy()
        h264_pred_apply(R, "ffref", kind, out, recs, coeffs)
        assert coeffs is None or not coeffs.any()
        d["k%d_rec" % kind] = recs
        d["k%d_out" % kind] = np.stack([out[y:y + n, x:x + n] for x, y, *_ in recs.tolist()])
        mask = np.ones(pic.shape, bool)
        for x, y, *_ in recs.tolist():
            mask[y:y + n, x:x + n] = False
        assert np.array_equal(out[mask], pic[mask])
    np.savez_compressed(os.path.join(OUT, "h264pred.npz"), **d)


def aac():
    """AACDecDSP.imdct_and_windowing: the decoder's four window tables and 14 consecutive frames of one channel that walk through
    every window-sequence transition and shape switch; outputs and the overlap state from the reference"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_vs_ref import aac_ref_windows
    rng = np.random.default_rng(15)
    win = aac_ref_windows()
    seq = np.array([0, 0, 1, 2, 2, 3, 0, 1, 2, 3, 1, 2, 3, 0], np.int32)
    kb = np.array([0, 1, 1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 1, 1], np.int32)
    nf = len(seq)
    coeffs = np.round(rng.standard_normal((nf, 1024)) * 2000.0 / (1 + np.arange(1024) / 64.0)).astype(np.float32)
    saved = (rng.standard_normal(512) * 0.05).astype(np.float32)
    d = {"sine_1024": win[0], "sine_128": win[1], "kbd_long_1024": win[2], "kbd_short_128": win[3], "seq": seq, "kb": kb, "coeffs": coeffs,
         "saved_in": saved.copy(), "prev": np.array([3, 1], np.int32)}
    out = np.zeros((nf, 1024), np.float32)
    prev = (3, 1)
    for f in range(nf):
        s2 = np.array([seq[f], prev[0]], np.int32); k2 = np.array([kb[f], prev[1]], np.int32)
        assert R.ffref_aac_imdct_and_windowing(ptr(np.ascontiguousarray(coeffs[f]), f32p), ptr(s2, i32p), ptr(k2, i32p), ptr(saved, f32p),
                                               ptr(out[f], f32p)) == 0
        prev = (int(seq[f]), int(kb[f]))
    d["out"], d["saved_out"] = out, saved
    np.savez_compressed(os.path.join(OUT, "aac.npz"), **d)


def fdsp():
    """AVFloatDSPContext vector ops: len 1024 and 37, operands across magnitudes (bit patterns stored as uint32)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_vs_ref import fdsp_operands
    rng = np.random.default_rng(12)
    d = {}
    for op in range(7):
        for n in (1024, 37):
            dst, s0, s1, s2, mul = fdsp_operands(rng, op, n)
            k = "op%d_n%d_" % (op, n)
            d[k + "dst"], d[k + "s0"], d[k + "s1"], d[k + "s2"], d[k + "mul"] = dst.copy(), s0.copy(), s1, s2, np.array([mul], np.float32)
            R.ffref_fdsp(op, ptr(dst, f32p), ptr(s0, f32p), ptr(s1, f32p), ptr(s2, f32p), mul, n)
            d[k + "out"], d[k + "out0"] = dst.view(np.uint32), s0.view(np.uint32)
    np.savez_compressed(os.path.join(OUT, "fdsp.npz"), **d)


SWS_UOPS_CASES = [   # conversions the reference compiles into ONE micro-op list with planes in natural order
    ("yuv444p", "rgb24", (70, 9, 70, 9), {}), ("rgb24", "yuv444p", (70, 9, 70, 9), {}), ("rgb565le", "rgb24", (66, 5, 66, 5), {}),
    ("rgb24", "rgb565le", (66, 5, 66, 5), {}), ("gray", "monob", (64, 6, 64, 6), {}), ("yuv444p10le", "rgb48le", (34, 4, 34, 4), {}),
    ("gbrpf32le", "gbrpf32le", (48, 6, 96, 6), {}), ("yuv444p", "yuv444p", (64, 12, 64, 24), {"scaler": 1}),
    ("gray", "gray", (40, 8, 96, 8), {}), ("rgba", "argb", (33, 3, 33, 3), {}), ("yuv444p", "rgb24", (64, 8, 128, 8), {}),
]


def sws_uops():
    """SwsOpBackend row: the micro-op lists of real conversions as the reference's graph cuts them, a seeded
    source picture, and backend_c's output for it"""
    import swsops as S
    S.declare_ref(R)
    O = ffi.oracle()
    d = {"ncases": np.array([len(SWS_UOPS_CASES)], np.int32)}
    for j, (sf, df, size, kw) in enumerate(SWS_UOPS_CASES):
        lists = S.capture_lists(R, O, sf, df, size, **kw)
        assert len(lists) == 1, (sf, df, size, len(lists))
        for k, v in S.pack_lists(lists).items():
            d["c%d_%s" % (j, k)] = v
        rng = np.random.default_rng(900 + j)
        sw, sh, dw, dh = size
        src, dst = S.Picture(R, sf, sw, sh, slack=1024), S.Picture(R, df, dw, dh, slack=1024)
        for rows in src.payload(R):
            if sf == "gbrpf32le":
                rows[:] = rng.random((rows.shape[0], rows.shape[1] // 4)).astype(np.float32).view(np.uint8)
            else:
                rows[:] = rng.integers(0, 256, rows.shape, dtype=np.uint8)
        S.unbind(R)
        assert S.convert(R, S.BACKEND_C | S.BACKEND_MEMCPY, src, dst, **kw) >= 0
        d["c%d_size" % j] = np.array(size, np.int32)
        d["c%d_name" % j] = np.frombuffer(("%s %s" % (sf, df)).encode(), np.uint8)
        for i, rows in enumerate(src.payload(R)):
            d["c%d_src%d" % (j, i)] = rows.copy()
        for i, rows in enumerate(dst.payload(R)):
            d["c%d_dst%d" % (j, i)] = rows.copy()
    np.savez_compressed(os.path.join(OUT, "sws_uops.npz"), **d)


def round2():
    """round 2's additions, one file: the prime-factor MDCT lengths other than 15xM, ff_vp9_loopfilter_sb on superblocks, the AAC
    stereo tools / long-term prediction / 960-sample windowing — inputs and the REAL reference's outputs"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import aac_gen as A
    import vp9_lf_gen as G
    from test_oracle_vs_ref import aac_tns_case
    u16p, i8p_ = C.POINTER(C.c_uint16), C.POINTER(C.c_int8)
    d = {}
    # ---- ff_tx_mdct_pfa_{3,5,7,9}xM
    rng = np.random.default_rng(2001)
    keys = []
    for len_ in (96, 1536, 640, 112, 1152):
        for inv, scale in ((0, 1.0), (1, 1.0 / len_)):
            x = rng.uniform(-1, 1, (2, len_ if inv else 2 * len_)).astype(np.float32)
            rc = R.ffref_tx_create(1, inv, len_, scale, 0)
            out = np.zeros((2, len_), np.float32)
            for t in range(2):
                R.ffref_tx_run(rc, ptr(out[t], f32p), ptr(x[t].copy(), f32p), 4)
            R.ffref_tx_free(rc)
            key = "mdct%d_%d_%r" % (len_, inv, scale)
            keys.append(key)
            d[key + "_in"], d[key + "_out"] = x, out
    d["mdct_keys"] = np.array(keys)
    # ---- ff_vp9_loopfilter_sb: three superblocks per depth (first / later row and column), stream-like masks
    rng = np.random.default_rng(2002)
    lim, mblim = G.filter_lut(3)
    d["lf_lim"], d["lf_mblim"] = lim, mblim
    n = 0
    for bd in (8, 10):
        for row, col in ((0, 0), (8, 8), (0, 16)):
            f = G.structured(rng, row // 8, col // 8, 23, 24)
            dt = np.uint8 if bd == 8 else np.uint16
            pl = []
            for sz in (192, 96, 96):
                base = np.cumsum(rng.integers(-2, 3, (sz, sz + 4)), axis=1) + 128
                pl.append(np.clip((base << (bd - 8)) + rng.integers(0, (1 << (bd - 8)) + 1, base.shape), 0, (1 << bd) - 1).astype(dt))
            before = [p.copy() for p in pl]
            level, mask = np.ascontiguousarray(f["level"]), np.ascontiguousarray(f["mask"])
            R.ffref_vp9_loopfilter_sb(bd, 1, 1, ptr(level, u8p), ptr(mask, u8p), row, col,
                                      *(C.cast(p.ctypes.data + k * p.strides[0] + k * p.itemsize, u8p) for p, k in zip(pl, (64, 32, 32))),
                                      pl[0].strides[0], pl[1].strides[0], ptr(lim, u8p), ptr(mblim, u8p))
            d["lf%d_par" % n] = np.array([bd, row, col], np.int32)
            d["lf%d_level" % n], d["lf%d_mask" % n] = level, mask
            for k in range(3):
                d["lf%d_in%d" % (n, k)], d["lf%d_out%d" % (n, k)] = before[k], pl[k]
            n += 1
    d["lf_n"] = np.array(n)
    # ---- AAC: mid/side + intensity (a long and a short pair), apply_ltp, update_ltp, imdct_and_windowing_960
    rng = np.random.default_rng(2003)
    for k, short in enumerate((0, 1)):
        c = A.cpe(rng, short)
        a0, a1 = A.spectrum(rng), A.spectrum(rng)
        for name in ("group_len", "ms_mask", "band_type0", "band_type1", "sf1", "swb"):
            d["st%d_%s" % (k, name)] = c[name]
        d["st%d_par" % k] = np.array([c["num_window_groups"], c["max_sfb"], c["ms_present"]], np.int32)
        d["st%d_in0" % k], d["st%d_in1" % k] = a0.copy(), a1.copy()
        R.ffref_aac_apply_mid_side_stereo(ptr(a0, f32p), ptr(a1, f32p), c["num_window_groups"], ptr(c["group_len"], u8p), c["max_sfb"],
                                          ptr(c["ms_mask"], u8p), ptr(c["band_type0"], i32p), ptr(c["band_type1"], i32p), ptr(c["swb"], u16p))
        R.ffref_aac_apply_intensity_stereo(ptr(a0, f32p), ptr(a1, f32p), c["num_window_groups"], ptr(c["group_len"], u8p), c["max_sfb"],
                                           c["ms_present"], ptr(c["ms_mask"], u8p), ptr(c["band_type1"], i32p), ptr(c["sf1"], f32p),
                                           ptr(c["swb"], u16p))
        d["st%d_out0" % k], d["st%d_out1" % k] = a0, a1
    win = [np.ctypeslib.as_array(R.ffref_aac_window(k), (sz,)).copy() for k, sz in ((0, 1024), (1, 128), (2, 1024), (3, 128))]
    for k in range(4):
        d["win%d" % k] = win[k]
    l = A.ltp(rng, A.LONG_START)
    t = aac_tns_case(rng, 0)
    t["swb"], t["num_swb"], t["max_sfb"] = l["swb"], l["num_swb"], l["max_sfb"]
    co = A.spectrum(rng)
    d["ltp_in"], d["ltp_state"], d["ltp_used"], d["ltp_swb"] = co.copy(), l["ltp_state"], l["used"], l["swb"]
    d["ltp_par"] = np.array([l["lag"], l["max_sfb"], l["num_swb"], t["tns_max_bands"]], np.int32)
    d["ltp_coef"], d["ltp_seq"], d["ltp_kb"] = np.float32(l["coef"]), l["seq"], l["kb"]
    for name in ("n_filt", "length", "direction", "order", "coef"):
        d["ltp_tns_" + name] = t[name]
    pf = np.zeros(1024, np.float32)
    assert R.ffref_aac_apply_ltp(ptr(co, f32p), ptr(l["ltp_state"], f32p), l["lag"], l["coef"], ptr(l["used"], i8p_), ptr(l["seq"], i32p),
                                 ptr(l["kb"], i32p), l["max_sfb"], l["num_swb"], t["tns_max_bands"], ptr(l["swb"], u16p), 1, ptr(t["n_filt"], i32p),
                                 ptr(t["length"], i32p), ptr(t["direction"], i32p), ptr(t["order"], i32p), ptr(t["coef"], f32p), ptr(pf, f32p)) == 0
    d["ltp_out"], d["ltp_pred"] = co, pf
    for k, (seq0, kb0) in enumerate(((0, 0), (1, 1), (2, 0))):
        buf, saved, out = A.spectrum(rng), A.spectrum(rng)[:512].copy(), A.spectrum(rng)
        st = (rng.standard_normal(3072) * 100).astype(np.float32)
        d["ul%d_in" % k], d["ul%d_buf" % k], d["ul%d_saved" % k], d["ul%d_output" % k] = st.copy(), buf, saved, out
        d["ul%d_par" % k] = np.array([seq0, kb0], np.int32)
        assert R.ffref_aac_update_ltp(ptr(st, f32p), ptr(buf, f32p), ptr(saved, f32p), ptr(out, f32p), seq0, kb0) == 0
        d["ul%d_out" % k] = st
    w960 = [np.ctypeslib.as_array(R.ffref_aac_window_len(960, k), (sz,)).copy() for k, sz in ((0, 960), (1, 120), (2, 960), (3, 120))]
    for k in range(4):
        d["w960_%d" % k] = w960[k]
    seq = np.array([0, 1, 2, 2, 3, 0, 1, 2, 3, 0], np.int32)
    kb = np.array([0, 1, 1, 0, 0, 1, 0, 0, 1, 1], np.int32)
    coeffs = np.round(rng.standard_normal((len(seq), 1024)) * 2000.0 / (1 + np.arange(1024) / 64.0)).astype(np.float32)
    saved = (rng.standard_normal(480) * 0.05).astype(np.float32)
    d["a960_seq"], d["a960_kb"], d["a960_coeffs"], d["a960_saved_in"] = seq, kb, coeffs, saved.copy()
    out = np.zeros((len(seq), 960), np.float32)
    prev = (0, 0)
    for f in range(len(seq)):
        s2, k2 = np.array([seq[f], prev[0]], np.int32), np.array([kb[f], prev[1]], np.int32)
        assert R.ffref_aac_imdct_and_windowing_len(960, ptr(np.ascontiguousarray(coeffs[f]), f32p), ptr(s2, i32p), ptr(k2, i32p), ptr(saved, f32p),
                                                   ptr(out[f], f32p)) == 0
        prev = (int(seq[f]), int(kb[f]))
    d["a960_out"], d["a960_saved_out"] = out, saved
    np.savez_compressed(os.path.join(OUT, "round2.npz"), **d)


def round3():
    """round 3's additions, one file: the scaler above 8 bits, range conversion, full-range and 4:2:2 sources to packed RGB, the
    prime-factor FFT and FFTs beyond one wave, the h264 tables at 10 / 12 bits — inputs and the REAL reference's outputs"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_oracle_vs_ref_sws_hbd as H
    import test_oracle_vs_ref_h264_hbd as B
    d = {}
    # ---- sws: (src name, w, h, dst name, w, h, flags, src_range, dst_range); 8-bit names from ffi.PIX, deeper ones from H.FMT
    rng = np.random.default_rng(3001)
    cases = [("p010le", 64, 36, "p010le", 128, 72, 4, 0, 0), ("yuv420p10le", 128, 72, "yuv420p", 64, 36, 4, 0, 0),
             ("yuv420p16le", 64, 36, "yuv420p16le", 96, 54, 4, 0, 0), ("yuv420p10le", 96, 48, "yuv420p10le", 48, 24, 2, 0, 0),
             ("yuv420p10le", 64, 36, "yuv420p10le", 96, 54, 4, 1, 0), ("yuv420p16le", 64, 36, "p016le", 96, 54, 2, 0, 1),
             ("yuvj420p", 64, 40, "yuv420p", 160, 88, 4, 1, 0), ("yuv420p", 96, 54, "yuvj420p", 96, 54, 2, 0, 1),
             ("yuvj420p", 64, 40, "rgb24", 160, 88, 4, 1, 1), ("yuvj420p", 64, 16, "bgra", 64, 16, 4, 1, 1),
             ("yuv422p", 64, 40, "rgb24", 160, 88, 4, 0, 0), ("yuv422p", 64, 16, "rgb24", 64, 16, 4, 0, 0)]
    names = []
    for k, (sn, sw, sh, dn, dw, dh, flags, sr, dr) in enumerate(cases):
        hb = sn in H.FMT and H.FMT[sn][1] > 8 or dn in H.FMT and H.FMT[dn][1] > 8
        sfmt = H.FMT[sn][0] if sn in H.FMT else ffi.PIX[sn]
        dfmt = H.FMT[dn][0] if dn in H.FMT else ffi.PIX[dn]
        base = {"yuvj420p": "yuv420p"}
        if hb:
            src = H.make_frame(sn, sw, sh, rng, pad=2)
            want = H.make_frame(dn, dw, dh, None, pad=0)
            sp, ss = H.planes_of(src)
            wp, ws = H.planes_of(want)
            ctx = R.ffref_sws_create_ranges(sw, sh, sfmt, dw, dh, dfmt, flags, 1, sr, dr)
        else:
            src = ffi.alloc_frame(ffi.PIX[base.get(sn, sn)], sw, sh, rng, pad=2)
            for pl in src:
                pl[::5, : pl.shape[1] // 2] = 255
                pl[3::7, pl.shape[1] // 3:] = 0
            want = ffi.alloc_frame(ffi.PIX[base.get(dn, dn)], dw, dh)
            sp, ss = ffi.planes(src)
            wp, ws = ffi.planes(want)
            ctx = R.ffref_sws_create(sw, sh, sfmt, dw, dh, dfmt, flags, 1)
        assert ctx and R.ffref_sws_scale(ctx, sp, ss, 0, sh, wp, ws) == dh
        R.ffref_sws_free(ctx)
        names.append("%s %d %d %s %d %d %d %d %d" % (sn, sw, sh, dn, dw, dh, flags, sr, dr))
        for i, a in enumerate(src):
            d["sws%d_src%d" % (k, i)] = a
        for i, a in enumerate(want):
            d["sws%d_out%d" % (k, i)] = a
    d["sws_cases"] = np.array(names)
    # ---- av_tx: ff_tx_fft_pfa lengths, both directions, and the first length beyond one wave
    rng = np.random.default_rng(3002)
    keys = []
    for len_, inv in ((120, 0), (120, 1), (960, 0), (960, 1), (96, 1), (1280, 0), (4096, 0)):
        x = (rng.standard_normal((2, 2 * len_)) * 10.0 ** rng.integers(-2, 3, (2, 1))).astype(np.float32)
        rc = R.ffref_tx_create(0, inv, len_, 1.0, 0)
        out = np.zeros((2, 2 * len_), np.float32)
        for t in range(2):
            R.ffref_tx_run(rc, ptr(out[t], f32p), ptr(x[t].copy(), f32p), 8)
        R.ffref_tx_free(rc)
        key = "fft%d_%d" % (len_, inv)
        keys.append(key)
        d[key + "_in"], d[key + "_out"] = x, out
    d["fft_keys"] = np.array(keys)
    # ---- h264 at 10 and 12 bits: idct 4x4 / 8x8, one luma and one chroma loop filter, a J-position qpel, chroma MC, (bi)weight
    O = ffi.oracle()
    B._sigs(R, O)
    R.ffref_h264_set_bit_depth.argtypes = [C.c_int]
    rng = np.random.default_rng(3003)
    for bd in (10, 12):
        R.ffref_h264_set_bit_depth(bd)
        for kind in (0, 1):
            n = 8 if kind else 4
            c = B.coefs(rng, n * n, bd)
            pic = B.pixels(rng, (n + 4, 40), bd, True)
            d["h%d_idct%d_c" % (bd, kind)], d["h%d_idct%d_in" % (bd, kind)] = c.copy(), pic.copy()
            R.ffref_h264_idct(kind, B.at(pic, 2, 5), B.bptr(c), pic.strides[0])
            d["h%d_idct%d_out" % (bd, kind)] = pic
        for kind, inner in ((0, 4), (5, 4), (2, 2), (7, 2)):
            pic = np.clip(300 + rng.integers(-12 << (bd - 8), (12 << (bd - 8)) + 1, (40, 40)), 0, (1 << bd) - 1).astype(np.uint16)
            tc = np.array([0, 1, 3, -1], np.int8)
            d["h%d_lf%d_in" % (bd, kind)] = pic.copy()
            R.ffref_h264_loop_filter_variant(kind, 0, B.at(pic, 12, 12), pic.strides[0], 40, 9, tc.ctypes.data_as(C.POINTER(C.c_int8)))
            d["h%d_lf%d_out" % (bd, kind)] = pic
        src, pic = B.pixels(rng, (32, 40), bd, True), B.pixels(rng, (32, 40), bd)
        d["h%d_mc_src" % bd], d["h%d_qpel_in" % bd] = src, pic.copy()
        R.ffref_h264_qpel(1, 0, 10, B.at(pic, 6, 8), B.at(src, 6, 8), pic.strides[0])
        d["h%d_qpel_out" % bd] = pic
        pic = B.pixels(rng, (32, 40), bd)
        d["h%d_chroma_in" % bd] = pic.copy()
        R.ffref_h264_chroma(0, 0, B.at(pic, 2, 4), B.at(src, 2, 4), pic.strides[0], 8, 3, 5)
        d["h%d_chroma_out" % bd] = pic
        pic = B.pixels(rng, (32, 40), bd, True)
        d["h%d_bw_in" % bd] = pic.copy()
        R.ffref_h264_biweight(0, B.at(pic, 1, 4), B.at(src, 1, 4), pic.strides[0], 16, 5, 37, -21, 9)
        d["h%d_bw_out" % bd] = pic
    R.ffref_h264_set_bit_depth(8)
    np.savez_compressed(os.path.join(OUT, "round3.npz"), **d)


def round4():
    """round 4's additions whose oracle side is a restatement: a scaled alpha plane (planar YUVA on both sides) and the VP9 superblock loop
    filter with the two sub-sampling shifts apart (4:2:2, 4:4:0) — inputs and the REAL reference's outputs"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import vp9_lf_gen as G
    d = {}
    rng = np.random.default_rng(4001)
    cases = [("yuva420p", 64, 36, "yuva420p", 128, 72, 4), ("yuva444p", 48, 32, "yuva420p", 100, 50, 2), ("yuva422p", 66, 38, "yuva420p", 33, 19, 2)]
    for k, (sn, sw, sh, dn, dw, dh, flags) in enumerate(cases):
        src = ffi.alloc_frame(ffi.PIX[sn], sw, sh, rng, pad=3)
        ctx = R.ffref_sws_create(sw, sh, ffi.PIX[sn], dw, dh, ffi.PIX[dn], flags, 1)
        assert ctx
        out = ffi.alloc_frame(ffi.PIX[dn], dw, dh)
        sp, ss = ffi.planes(src)
        dp, ds = ffi.planes(out)
        assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, dp, ds) == dh
        R.ffref_sws_free(ctx)
        d["a%d_meta" % k] = np.array([ffi.PIX[sn], sw, sh, ffi.PIX[dn], dw, dh, flags], np.int32)
        for p in range(4):
            d["a%d_src%d" % (k, p)], d["a%d_dst%d" % (k, p)] = src[p], out[p]
    d["a_n"] = np.int32(len(cases))
    lim, mblim = G.filter_lut(3)
    d["lf_lim"], d["lf_mblim"] = lim, mblim
    n = 0
    for bd in (8, 10):
        for ss_h, ss_v in ((1, 0), (0, 1)):
            for row, col in ((0, 0), (8, 8)):
                dt = np.uint8 if bd == 8 else np.uint16
                cw, chh = 64 >> ss_h, 64 >> ss_v
                f = G.structured(rng, row // 8, col // 8, 23, 22, ss_h, ss_v)
                mk = lambda h, w: np.clip(np.cumsum(rng.integers(-2, 3, (h, w + 7)), axis=1) * (1 << (bd - 8)) + (1 << (bd - 1)), 0, (1 << bd) - 1).astype(dt)
                pl = [mk(192, 192), mk(3 * chh, 3 * cw), mk(3 * chh, 3 * cw)]
                d["lf%d_par" % n] = np.array([bd, ss_h, ss_v, row, col], np.int32)
                d["lf%d_level" % n], d["lf%d_mask" % n] = np.ascontiguousarray(f["level"]), np.ascontiguousarray(f["mask"])
                for k in range(3):
                    d["lf%d_in%d" % (n, k)] = pl[k].copy()
                pos = ((64, 64), (chh, cw), (chh, cw))
                R.ffref_vp9_loopfilter_sb(bd, ss_h, ss_v, ptr(d["lf%d_level" % n], u8p), ptr(d["lf%d_mask" % n], u8p), row, col,
                                          *(C.cast(p.ctypes.data + r * p.strides[0] + c * p.itemsize, u8p) for p, (r, c) in zip(pl, pos)),
                                          pl[0].strides[0], pl[1].strides[0], ptr(lim, u8p), ptr(mblim, u8p))
                for k in range(3):
                    d["lf%d_out%d" % (n, k)] = pl[k]
                n += 1
    d["lf_n"] = np.int32(n)
    np.savez_compressed(os.path.join(OUT, "round4.npz"), **d)


def h264pred422():
    """pred8x8[] at chroma_format_idc 2 (the 8 wide x 16 tall forms): at 8 bits a grid of independent blocks of one 136x200
    picture, every mode three times or more (batch kind 7); at 10 bits one 40x48 patch per mode with the block at (8, 16)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_vs_ref import h264_pred_grid, h264_pred_apply
    R.ffref_h264_pred_set_format.argtypes = [C.c_int, C.c_int]
    rng = np.random.default_rng(31)
    pic = rng.integers(0, 256, (136, 200), dtype=np.uint8)
    pic[:, 100:] = np.clip(np.add.outer(np.arange(136) * 3, np.arange(100) * -2) + 120 + rng.integers(-4, 5, (136, 100)), 0, 255)
    pic[64:, :60] = rng.choice(np.array([0, 255], np.uint8), (72, 60))
    d = {"pic": pic}
    R.ffref_h264_pred_set_format(8, 2)
    recs = h264_pred_grid(rng, 7, 136, 200)
    out = pic.copy()
    h264_pred_apply(R, "ffref", 7, out, recs)
    d["k7_rec"] = recs
    d["k7_out"] = np.stack([out[y:y + 16, x:x + 8] for x, y, *_ in recs.tolist()])
    mask = np.ones(pic.shape, bool)
    for x, y, *_ in recs.tolist():
        mask[y:y + 16, x:x + 8] = False
    assert np.array_equal(out[mask], pic[mask]) and len(recs) >= 33
    R.ffref_h264_pred_set_format(10, 2)
    p10 = rng.integers(0, 1024, (11, 40, 48)).astype(np.uint16)
    p10[3] = np.clip(np.add.outer(np.arange(40) * 30, np.arange(48) * -17) + 500 + rng.integers(-9, 10, (40, 48)), 0, 1023)   # the plane mode on a slope
    d["p10_in"] = p10.copy()
    for mode in range(11):
        R.ffref_h264_pred8x8(mode, C.cast(p10[mode].ctypes.data + (8 * 48 + 16) * 2, u8p), 96)
    d["p10_out"] = p10
    R.ffref_h264_pred_set_format(8, 1)
    np.savez_compressed(os.path.join(OUT, "h264pred422.npz"), **d)


TXW_TYPES = {"d_fft": 2, "d_mdct": 3, "i_fft": 4, "i_mdct": 5}   # AVTXType values (libavutil/tx.h:48-69)
TXW_CASES = [(kind, inv, len_) for kind in ("d_fft", "d_mdct", "i_fft", "i_mdct") for inv in (0, 1)
             for len_ in ((4, 8, 16, 32, 64, 256) if kind.endswith("fft") else (16, 32, 64, 256, 1024))]


def txw_input(kind, inv, len_, nt, rng):
    """seeded input rows of one wide-type case: doubles ~ N(0, 1) * 10^k, int32 over the whole range for the FFT's wrapping sums and at
    the codecs' headroom (|x| < 2^24) for the MDCT"""
    n_in = 2 * len_ if (kind.endswith("fft") or not inv) else len_
    if kind[0] == "d":
        return (rng.standard_normal((nt, n_in)) * 10.0 ** rng.integers(-3, 4, (nt, 1))).astype(np.float64)
    hi = 2 ** 31 - 1 if kind.endswith("fft") else 2 ** 24
    x = rng.integers(-hi, hi, (nt, n_in), dtype=np.int64).astype(np.int32)
    x[0, : min(4, n_in)] = [-2 ** 31, 2 ** 31 - 1, -1, 0][: min(4, n_in)]
    return x


def tx_wide():
    """AV_TX_DOUBLE_* / AV_TX_INT32_* FFT and MDCT (powers of two), both directions: inputs + the reference's outputs
    (tests/golden/tx_wide.npz)"""
    d = {}
    rng = np.random.default_rng(606)
    for kind, inv, len_ in TXW_CASES:
        is_int, mdct = kind[0] == "i", kind.endswith("mdct")
        scale = (1.0 / len_ if inv else 1.0) if not is_int else (1.0 if not inv else 1.0 / 64)
        rc = (R.ffref_tx_create(TXW_TYPES[kind], inv, len_, scale, 0) if is_int else R.ffref_tx_create_d(TXW_TYPES[kind], inv, len_, scale, 0))
        assert rc, (kind, inv, len_)
        x = txw_input(kind, inv, len_, 2, rng)
        out = np.zeros((2, len_ if mdct else 2 * len_), x.dtype)
        for t in range(2):
            xi = x[t].copy()
            R.ffref_tx_run(rc, out[t].ctypes.data, xi.ctypes.data, x.dtype.itemsize * (1 if mdct else 2))
        R.ffref_tx_free(rc)
        key = "%s_%d_%d" % (kind, inv, len_)
        d[key + "_in"], d[key + "_out"], d[key + "_scale"] = x, out, np.array([scale])
    np.savez_compressed(os.path.join(OUT, "tx_wide.npz"), **d)


DCST1_CASES = [(12, 64, 1.0 / 64), (15, 64, 1.0 / 64), (12, 4, 1.0), (15, 4, -1.5), (12, 10, 0.75), (15, 66, 2.0), (12, 256, 1.0), (15, 100, 0.75),
               (12, 1024, 1.0 / 1024), (15, 1024, 1.0)]


def tx_dcst1():
    """AV_TX_FLOAT_DCT_I / AV_TX_FLOAT_DST_I, forward: inputs + the reference's outputs (tests/golden/tx_dcst1.npz)"""
    d = {}
    rng = np.random.default_rng(707)
    for typ, n, scale in DCST1_CASES:
        rc = R.ffref_tx_create(typ, 0, n, scale, 0)
        assert rc, (typ, n)
        x = (rng.standard_normal((3, n)) * 10.0 ** rng.integers(-2, 3, (3, 1))).astype(np.float32)
        out = np.zeros((3, n), np.float32)
        for t in range(3):
            xi = np.zeros(2 * n + 8, np.float32)
            xi[:n] = x[t]
            o = np.zeros(2 * n + 8, np.float32)   # (the reference's RDFT uses the output buffer as its work array)
            R.ffref_tx_run(rc, o.ctypes.data, xi.ctypes.data, 4)
            out[t] = o[:n]
        R.ffref_tx_free(rc)
        key = "t%d_%d" % (typ, n)
        d[key + "_in"], d[key + "_out"], d[key + "_scale"] = x, out, np.array([scale], np.float32)
    np.savez_compressed(os.path.join(OUT, "tx_dcst1.npz"), **d)


def sws_rgbin():
    """packed 8-bit RGB sources into YUV targets through the reference's sws_scale(): inputs + outputs (tests/golden/sws_rgbin.npz)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_oracle_vs_ref_sws_rgbin as T
    d = {}
    cases = [("rgb24", 64, 36, "yuv420p", 64, 36, 4), ("bgra", 96, 54, "nv12", 64, 36, 2), ("argb", 65, 37, "yuv420p", 65, 37, 4),
             ("abgr", 64, 36, "yuv444p", 100, 36, 4), ("bgr24", 64, 36, "nv12", 64, 36, 4), ("rgba", 64, 36, "yuv420p", 64, 36, 4 | 0x4000),
             ("rgb24", 48, 32, "yuv422p", 96, 64, 4)]
    rng = np.random.default_rng(808)
    for i, (sf, sw, sh, df, dw, dh, fl) in enumerate(cases):
        rgb = T.make_rgb(sf, sw, sh, rng, pad=0)
        want = T.alloc_dst(df, dw, dh, pad=0)
        ctx = R.ffref_sws_create(sw, sh, PIX[sf], dw, dh, T.DST[df][0], fl, 1)
        sp, ss = T.planes_of([rgb])
        wp, ws = T.planes_of(want)
        assert R.ffref_sws_scale(ctx, sp, ss, 0, sh, wp, ws) == dh
        R.ffref_sws_free(ctx)
        d["c%d_meta" % i] = np.array([PIX[sf], sw, sh, T.DST[df][0], dw, dh, fl], np.int64)
        d["c%d_src" % i] = rgb
        for p, a in enumerate(want):
            d["c%d_dst%d" % (i, p)] = a
    d["ncases"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(OUT, "sws_rgbin.npz"), **d)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        sws(); h264(); h264_misc(); me(); tx(); fft(); hevc(); fdsp(); vp9(); h264pred(); aac(); sws_uops(); round2(); round3(); h264pred422(); round4(); tx_wide(); sws_rgbin(); tx_dcst1()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
