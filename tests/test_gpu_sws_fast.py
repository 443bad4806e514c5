"""GPU parity of the column-walking fast path of the scaler (sws_colwalk.hip) vs the oracle, bit-exact.

Every case must actually take the fast path (ctx.fast_path) and is run in each measured variant
(FFHIP_CW_LUMA_GROUPS / FFHIP_CW_DEPTH / FFHIP_CW_PLAIN / FFHIP_CW_STRIP) and, as a cross-check of the
two kernels against each other, with the fast path disabled (FFHIP_SWS_FAST=0)."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import PIX, ptr

pytestmark = pytest.mark.gpu

VARIANTS = [
    {},
    {"FFHIP_CW_LUMA_GROUPS": "1"},
    {"FFHIP_CW_DEPTH": "6"},
    {"FFHIP_CW_LUMA_GROUPS": "1", "FFHIP_CW_DEPTH": "6", "FFHIP_CW_STRIP": "37"},
    {"FFHIP_CW_PLAIN": "1"},
    {"FFHIP_CW_OPT": "0"},
    {"FFHIP_CW_OPT": "0", "FFHIP_CW_DEPTH": "6", "FFHIP_CW_LUMA_GROUPS": "1"},
    {"FFHIP_CW_STRIP": "128"},
    {"FFHIP_CW_DUP": "0"},
    {"FFHIP_SWS_FAST": "0"},
]


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _upload_aligned(arrs, n, rng):
    """planes -> [n, rows, pitch] cuda tensors with a 64-byte aligned pitch and DIFFERENT frames"""
    torch = _torch()
    out, host = [], []
    for a in arrs:
        pitch = (a.shape[1] + 63) // 64 * 64
        h = rng.integers(0, 256, (n, a.shape[0], pitch), dtype=np.uint8)
        h[0, :, :a.shape[1]] = a
        host.append(h)
        out.append(torch.from_numpy(h).to("cuda:0"))
    return out, host


def _run(sf, sw, sh, df, dw, dh, flags, banks=None, env=None, monkeypatch=None, n=3, seed=1, need_mfma=False, need="fast"):
    from ffmpeg_amd import swscale as S
    torch = _torch()
    rng = np.random.default_rng(seed)
    for k in ("FFHIP_SWS_UP2", "FFHIP_UP2_FSHIFT", "FFHIP_UP2_STRIP", "FFHIP_UP2_DEPTH", "FFHIP_UP2_VAR", "FFHIP_UP2_XCD",
              "FFHIP_SWS_DOWN2", "FFHIP_DN2_XCD", "FFHIP_DN2_STRIP", "FFHIP_SWS_UP2RGB", "FFHIP_UP2RGB_STEPS", "FFHIP_UP2RGB_FPP", "FFHIP_SWS_RGB2", "FFHIP_SWS_EQRGB", "FFHIP_EQRGB_STEPS",
              "FFHIP_EQRGB_FPP", "FFHIP_SWS_DOWN32", "FFHIP_SWS_UP32"):
        monkeypatch.delenv(k, raising=False)
    # the general kernels at exact 2:1 / 1:2 sizes need the fast paths switched off: a knob, i.e. libffhip_measure.so (conftest.py);
    # every other size runs the product library
    if need != "down2" and sw == 2 * dw and sh == 2 * dh:
        monkeypatch.setenv("FFHIP_SWS_DOWN2", "0")  # test_down2* covers sws_down2.hip
    if need != "down32" and 2 * sw == 3 * dw and 2 * sh == 3 * dh:
        monkeypatch.setenv("FFHIP_SWS_DOWN32", "0")  # test_down32* covers sws_down32.hip
    if need != "up32" and ((2 * dw == 3 * sw and 2 * dh == 3 * sh) or (3 * dw == 4 * sw and 3 * dh == 4 * sh)):
        monkeypatch.setenv("FFHIP_SWS_UP32", "0")  # test_up32* covers the 8-bit twin of sws_up32.hip
    if need != "up2" and dw == 2 * sw and dh == 2 * sh:
        monkeypatch.setenv("FFHIP_SWS_UP2", "0")   # these tests are about the general kernels; test_up2* covers sws_up2.hip
    if need != "up2rgb" and dw == 2 * sw and dh == 2 * sh:
        monkeypatch.setenv("FFHIP_SWS_UP2RGB", "0")  # ... and test_up2rgb* sws_up2rgb.hip
    for k in ("FFHIP_CW_LUMA_GROUPS", "FFHIP_CW_DEPTH", "FFHIP_CW_PLAIN", "FFHIP_CW_STRIP", "FFHIP_SWS_FAST", "FFHIP_CW_OPT",
              "FFHIP_SWS_MFMA", "FFHIP_MF_STRIP", "FFHIP_CWRGB_DIRECT", "FFHIP_CWRGB_STRIP", "FFHIP_SWS_WIDE", "FFHIP_CW_DUP", "FFHIP_LW_STRIP", "FFHIP_LW_AHEAD"):
        monkeypatch.delenv(k, raising=False)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    if banks is None:
        banks = ht.banks()
    t = ffi.make_otables(sw, sh, PIX[sf], dw, dh, PIX[df], flags, banks, ht.coeffs())
    tabs = None
    if banks is not None:
        from ffmpeg_amd import _lib
        tabs = _lib.SwsTables()
        C.memmove(C.byref(tabs), C.byref(ht.t), C.sizeof(tabs))
        keep = []
        for name in ("hLum", "hChr", "vLum", "vChr"):
            f, p, fs, nn = banks[name]
            f = np.ascontiguousarray(f, np.int16); p = np.ascontiguousarray(p, np.int32)
            keep += [f, p]
            setattr(tabs, name, _lib.SwsFilter(ptr(f, ffi.i16p), ptr(p, ffi.i32p), fs, nn))
    ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], flags, tables=tabs)
    if need == "up2":
        assert ctx.up2_path, "case does not reach the exact-2x kernel"
    elif need == "wide":
        assert ctx.wide_path, "case does not reach the wide-bank walker"
    elif need == "down2":
        assert ctx.down2_path, "case does not reach the exact-2:1 kernel"
    elif need == "up2rgb":
        assert ctx.up2rgb_path, "case does not reach the exact-2x kernel with the RGB writer"
    elif need == "down32":
        assert ctx.paths & 4096, "case does not reach the exact-3:2 kernel"
    elif need == "up32":
        assert ctx.paths & 8192, "case does not reach the exact-3:2 / 4:3 up-scaler"
    elif need == "fast":
        assert ctx.fast_path, "case does not reach the column walker"
    if (env or {}).get("FFHIP_SWS_MFMA") == "1" and need_mfma:
        assert ctx.mfma_path, "case does not reach the matrix-core variant"
    first = ffi.alloc_frame(PIX[sf], sw, sh, rng)
    dsrc, hsrc = _upload_aligned(first, n, rng)
    shapes = S.plane_shapes(PIX[df], dw, dh)
    ddst = [torch.full((n, r, (c + 63) // 64 * 64), 0xA5, dtype=torch.uint8, device="cuda:0") for r, c in shapes]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for f in range(n):
        src = [h[f, :, :a.shape[1]] for h, a in zip(hsrc, first)]
        src = [np.ascontiguousarray(a) for a in src]
        want = ffi.alloc_frame(PIX[df], dw, dh)
        sp, ss = ffi.planes(src)
        dp, ds = ffi.planes(want)
        assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
        for p, a in enumerate(want):
            got = ddst[p][f].cpu().numpy()
            assert np.array_equal(got[:, :a.shape[1]], a), "frame %d plane %d: %d mismatches" % (
                f, p, (got[:, :a.shape[1]] != a).sum())
            assert (got[:, a.shape[1]:] == 0xA5).all(), "wrote past the row end"
    ctx.close()


CASES = [
    ("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_BICUBIC),
    ("nv21", 192, 108, "nv21", 384, 216, ffi.SWS_BICUBIC),
    ("nv12", 192, 108, "nv21", 384, 216, ffi.SWS_BICUBIC),
    ("nv12", 64, 40, "yuv420p", 192, 104, ffi.SWS_BICUBIC),          # interleaved -> planar, 3x / 2.6x
    ("yuv420p", 128, 72, "nv12", 256, 144, ffi.SWS_BICUBIC),         # planar -> interleaved
    ("yuv420p", 128, 72, "nv21", 384, 216, ffi.SWS_BICUBIC),
    ("yuv420p", 128, 72, "yuv420p", 256, 144, ffi.SWS_BICUBIC),      # planar -> planar: three single-plane jobs
    ("nv12", 1048, 600, "nv12", 2096, 1416, ffi.SWS_BICUBIC),        # several column blocks and strips, ragged last block
    ("nv12", 32, 16, "nv12", 64, 32, ffi.SWS_BICUBIC),               # a single partial wave
    ("nv12", 128, 72, "nv12", 192, 108, ffi.SWS_BICUBIC),            # 1.5x: byte-aligned source spans
    ("yuv420p", 240, 136, "yuv420p", 320, 180, ffi.SWS_BICUBIC),     # 1.33x
    ("yuv420p", 202, 120, "nv12", 456, 270, ffi.SWS_BICUBIC),        # 2.26x, srcW % 4 != 0, odd chroma width
    ("nv21", 90, 50, "yuv420p", 200, 110, ffi.SWS_BILINEAR),         # padded banks at a fractional ratio
    ("yuv422p", 128, 72, "yuv422p", 256, 160, ffi.SWS_BICUBIC),      # 4:2:2 / 4:4:4: other chroma plane sizes, same kernels
    ("yuv444p", 128, 72, "yuv420p", 272, 144, ffi.SWS_BICUBIC),
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join("%s=%s" % (k[6:], v) for k, v in e.items()) or "default")
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_fast_path(case, env, monkeypatch):
    _run(*case, env=env, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF)


def test_fast_path_full_size(monkeypatch):
    """BASELINE configs[1] frame size, two different frames"""
    _run("nv12", 1920, 1080, "nv12", 3840, 2160, ffi.SWS_BICUBIC, monkeypatch=monkeypatch, n=2, seed=77)
    _run("nv12", 1920, 1080, "nv12", 3840, 2160, ffi.SWS_BICUBIC, env={"FFHIP_CW_LUMA_GROUPS": "1", "FFHIP_CW_DEPTH": "6"},
         monkeypatch=monkeypatch, n=2, seed=78)


# ---------------------------------------------------------------------------------------------
# exact 2x up-scaling on the static-schedule kernel (sws_up2.hip): BASELINE configs[1]'s shape
# ---------------------------------------------------------------------------------------------
UP2_CASES = [
    ("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_BICUBIC),
    ("nv21", 192, 108, "nv21", 384, 216, ffi.SWS_BICUBIC),
    ("nv12", 192, 108, "nv21", 384, 216, ffi.SWS_BICUBIC),           # the channels swap bytes on the way
    ("nv21", 192, 108, "nv12", 384, 216, ffi.SWS_BICUBIC),
    ("yuv420p", 128, 72, "yuv420p", 256, 144, ffi.SWS_BICUBIC),      # three single-plane jobs
    ("nv12", 1048, 600, "nv12", 2096, 1200, ffi.SWS_BICUBIC),        # several column blocks and strips, ragged last block
    ("nv12", 32, 16, "nv12", 64, 32, ffi.SWS_BICUBIC),               # a single partial wave; chroma rows of 4 groups
    ("nv12", 16, 8, "nv12", 32, 16, ffi.SWS_BICUBIC),                # the smallest planes the kernel takes
    ("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_BILINEAR),          # 2-tap banks zero-padded to 4
    ("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_POINT),             # 1-tap banks
    ("yuv420p", 136, 72, "yuv420p", 272, 144, ffi.SWS_BICUBLIN),     # bicubic luma, bilinear chroma
    ("nv12", 520, 90, "nv12", 1040, 180, ffi.SWS_BICUBIC),           # 130 luma groups: 2.03 waves per row
    ("yuv422p", 128, 72, "yuv422p", 256, 144, ffi.SWS_BICUBIC),      # 4:2:2: chroma planes 64 x 72 -> 128 x 144
    ("yuv444p", 128, 72, "yuv444p", 256, 144, ffi.SWS_BICUBIC),      # 4:4:4: three planes of the luma's size
]
UP2_VARIANTS = [
    {},
    {"FFHIP_UP2_FSHIFT": "0"},
    {"FFHIP_UP2_FSHIFT": "1", "FFHIP_UP2_DEPTH": "6"},
    {"FFHIP_UP2_FSHIFT": "2", "FFHIP_UP2_STRIP": "12"},
    {"FFHIP_UP2_STRIP": "6", "FFHIP_UP2_DEPTH": "6"},
    {"FFHIP_UP2_STRIP": "1000"},
    {"FFHIP_UP2_XCD": "0", "FFHIP_UP2_FSHIFT": "1"},
    {"FFHIP_UP2_XCD": "0", "FFHIP_UP2_FSHIFT": "2", "FFHIP_UP2_DEPTH": "6"},
]


@pytest.mark.parametrize("env", UP2_VARIANTS, ids=lambda e: ",".join("%s=%s" % (k[10:], v) for k, v in e.items()) or "default")
@pytest.mark.parametrize("case", UP2_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_up2(case, env, monkeypatch):
    _run(*case, env=env, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, need="up2")


@pytest.mark.parametrize("n", [1, 2, 5])
def test_up2_odd_batches(n, monkeypatch):
    """a wave shared by 2 or 4 frames with the batch ending inside it"""
    for fs in ("1", "2"):
        _run("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_BICUBIC, env={"FFHIP_UP2_FSHIFT": fs}, monkeypatch=monkeypatch, n=n,
             seed=n, need="up2")


def test_up2_full_size(monkeypatch):
    """BASELINE configs[1] frame size, different frames"""
    _run("nv12", 1920, 1080, "nv12", 3840, 2160, ffi.SWS_BICUBIC, monkeypatch=monkeypatch, n=2, seed=177, need="up2")
    _run("yuv420p", 1920, 1080, "yuv420p", 3840, 2160, ffi.SWS_BICUBIC, env={"FFHIP_UP2_DEPTH": "3"}, monkeypatch=monkeypatch,
         n=3, seed=178, need="up2")


def test_up2_saturating_content(monkeypatch):
    """rows of 0 / 255 runs drive the bicubic sums past 32767 (min(., 32767) in hScale8To15_c) and below 0 (clip)"""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    monkeypatch.delenv("FFHIP_SWS_UP2", raising=False)
    sw, sh = 256, 64
    ht = S.HostTables(sw, sh, PIX["nv12"], 2 * sw, 2 * sh, PIX["nv12"], ffi.SWS_BICUBIC)
    t = ffi.make_otables(sw, sh, PIX["nv12"], 2 * sw, 2 * sh, PIX["nv12"], ffi.SWS_BICUBIC, ht.banks(), ht.coeffs())
    ctx = S.SwsContext(sw, sh, PIX["nv12"], 2 * sw, 2 * sh, PIX["nv12"], ffi.SWS_BICUBIC)
    assert ctx.up2_path
    rng = np.random.default_rng(9)
    y = np.where(rng.integers(0, 2, (sh, sw)) > 0, 255, 0).astype(np.uint8)
    y[::2] = np.repeat(np.where(rng.integers(0, 2, (sh // 2, sw // 4)) > 0, 255, 0).astype(np.uint8), 4, axis=1)
    uv = np.where(rng.integers(0, 2, (sh // 2, sw)) > 0, 255, 0).astype(np.uint8)
    dsrc = [torch.from_numpy(y[None].copy()).to("cuda:0"), torch.from_numpy(uv[None].copy()).to("cuda:0")]
    ddst = [torch.zeros((1, 2 * sh, 2 * sw), dtype=torch.uint8, device="cuda:0"),
            torch.zeros((1, sh, 2 * sw), dtype=torch.uint8, device="cuda:0")]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    want = ffi.alloc_frame(PIX["nv12"], 2 * sw, 2 * sh)
    sp, ss = ffi.planes([y, uv])
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == 2 * sh
    for p, a in enumerate(want):
        assert np.array_equal(ddst[p][0].cpu().numpy(), a)
    ctx.close()


def _adversarial_banks(rng, srcW, srcH, dstW, dstH, extreme):
    """4x4-tap banks with arbitrary (eligible) positions and full-range int16 coefficients: the
    checkasm recipe (tests/checkasm/sw_scale.c:356-458) pushed through the fused kernel."""
    def hbank(n, sw):
        pos = np.zeros(n, np.int32)
        for g in range(0, n, 4):
            base = int(rng.integers(0, (sw - 8) // 4 + 1)) * 4
            lo = int(rng.integers(0, 5))
            pos[g:g + 4] = base + np.sort(rng.integers(lo, 5, 4))
        pos = np.minimum(pos, sw - 4)
        if extreme:
            f = rng.integers(-32768, 32768, (n, 4)).astype(np.int16)
            f[::3] = -((1 << 14) // 3)
            f[::3, 0] = (1 << 15) - 1
            f[1::7] = 32767
            f[2::7] = -32768
        else:
            f = rng.integers(-3000, 9000, (n, 4)).astype(np.int16)
        return (f.reshape(-1), pos, 4, n)

    def vbank(n, sh):
        steps = rng.integers(0, 3, n)
        pos = np.minimum(np.cumsum(steps), sh - 4).astype(np.int32)
        if extreme:
            f = rng.integers(-32768, 32768, (n, 4)).astype(np.int16)
            f[::5] = 32767
            f[1::5] = -32768
        else:
            f = rng.integers(-1500, 4000, (n, 4)).astype(np.int16)
        return (f.reshape(-1), pos, 4, n)
    return {"hLum": hbank(dstW, srcW), "hChr": hbank(dstW // 2, srcW // 2), "vLum": vbank(dstH, srcH),
            "vChr": vbank(dstH // 2, srcH // 2)}


@pytest.mark.parametrize("extreme", [0, 1])
@pytest.mark.parametrize("env", [{}, {"FFHIP_CW_LUMA_GROUPS": "1", "FFHIP_CW_DEPTH": "6"}, {"FFHIP_CW_PLAIN": "1"},
                                 {"FFHIP_CW_OPT": "0"}], ids=["default", "g1d6", "plain", "noopt"])
@pytest.mark.parametrize("fmts", [("nv12", "nv12"), ("yuv420p", "nv21"), ("nv21", "yuv420p"), ("yuv420p", "yuv420p")])
def test_fast_path_adversarial_tables(fmts, env, extreme, monkeypatch):
    sw, sh, dw, dh = 200, 120, 520, 300
    rng = np.random.default_rng(extreme * 100 + len(env))
    banks = _adversarial_banks(rng, sw, sh, dw, dh, extreme)
    _run(fmts[0], sw, sh, fmts[1], dw, dh, ffi.SWS_BICUBIC, banks=banks, env=env, monkeypatch=monkeypatch, n=2,
         seed=extreme + 5)


# ---------------------------------------------------------------------------------------------
# unscaled yuv420p -> rgb24: the LDS-transposing streaming kernel, aligned pitches, ragged widths
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["", "plain", "old"])
@pytest.mark.parametrize("w,h", [(16, 2), (1000, 6), (1048, 4), (2050, 4), (3840, 8), (1920, 1080), (4112, 2), (30, 4)])
@pytest.mark.parametrize("dst", ["rgb24", "bgr24"])
def test_unscaled_rgb_transposed(w, h, dst, variant, monkeypatch):
    from ffmpeg_amd import swscale as S
    import test_gpu_sws as T
    torch = _torch()
    if variant:
        monkeypatch.setenv("FFHIP_YUV2RGB_VARIANT", variant)
    else:
        monkeypatch.delenv("FFHIP_YUV2RGB_VARIANT", raising=False)
    rng = np.random.default_rng(w * 7 + h)
    n = 2
    first = ffi.alloc_frame(PIX["yuv420p"], w, h, rng)
    dsrc, hsrc = _upload_aligned(first, n, rng)
    pitch = (3 * w + 63) // 64 * 64
    ddst = [torch.full((n, h, pitch), 0x5A, dtype=torch.uint8, device="cuda:0")]
    ctx = S.SwsContext(w, h, PIX["yuv420p"], w, h, PIX[dst], S.SWS_BICUBIC)
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    got = ddst[0].cpu().numpy()
    for f in range(n):
        src = [np.ascontiguousarray(hh[f, :, :a.shape[1]]) for hh, a in zip(hsrc, first)]
        want = T._oracle_unscaled(src, w, h, dst == "bgr24")
        wv = 3 * (w & ~1)
        assert np.array_equal(got[f, :, :wv], want[:, :wv]), "frame %d: %d mismatches" % (f, (got[f, :, :wv] != want[:, :wv]).sum())
        assert (got[f, :, 3 * w:] == 0x5A).all()
    ctx.close()


# ---------------------------------------------------------------------------------------------
# the matrix-core horizontal pass (k_sws_mfma): same results, bit for bit
# ---------------------------------------------------------------------------------------------
MFMA_CASES = [
    ("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_BICUBIC),
    ("nv21", 192, 108, "nv21", 384, 216, ffi.SWS_BICUBIC),
    ("nv12", 192, 108, "nv21", 384, 216, ffi.SWS_BICUBIC),
    ("yuv420p", 128, 72, "yuv420p", 256, 144, ffi.SWS_BICUBIC),
    ("nv12", 1048, 600, "nv12", 2096, 1416, ffi.SWS_BICUBIC),        # several column blocks / chunks, ragged last block
    ("nv12", 64, 40, "nv12", 192, 104, ffi.SWS_BICUBIC),             # 3x / 2.6x
    ("nv12", 1920, 1080, "nv12", 3840, 2160, ffi.SWS_BICUBIC),       # BASELINE configs[1]
]


@pytest.mark.parametrize("strip", ["", "100", "37"])
@pytest.mark.parametrize("case", MFMA_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_mfma_path(case, strip, monkeypatch):
    env = {"FFHIP_SWS_MFMA": "1"}
    if strip:
        env["FFHIP_MF_STRIP"] = strip
    _run(*case, env=env, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, n=2, need_mfma=True)


@pytest.mark.parametrize("fmts", [("nv12", "nv12"), ("nv21", "nv12"), ("yuv420p", "yuv420p")])
def test_mfma_path_adversarial_tables(fmts, monkeypatch):
    """arbitrary eligible positions / coefficients (the non-extreme recipe: horizontal sums cannot wrap int16);
    banks whose tiles do not fit fall back to the column walker - either way the result is the oracle's"""
    sw, sh, dw, dh = 200, 120, 520, 300
    rng = np.random.default_rng(77)
    banks = _adversarial_banks(rng, sw, sh, dw, dh, 0)
    _run(fmts[0], sw, sh, fmts[1], dw, dh, ffi.SWS_BICUBIC, banks=banks, env={"FFHIP_SWS_MFMA": "1"}, monkeypatch=monkeypatch,
         n=2, seed=6)


# ---------------------------------------------------------------------------------------------
# banks with fewer than 4 taps ride the fast path zero-padded: bilinear / point up-scaling, 1:1 re-packing
# ---------------------------------------------------------------------------------------------
PADDED_CASES = [
    ("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_BILINEAR),          # 2 x 2 taps
    ("nv21", 96, 64, "nv21", 200, 136, ffi.SWS_BILINEAR),
    ("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_POINT),             # 1 x 1 tap
    ("yuv420p", 128, 72, "nv12", 128, 72, ffi.SWS_BICUBIC),          # 1:1 planar -> NV12 (identity banks)
    ("nv12", 640, 360, "yuv420p", 640, 360, ffi.SWS_BICUBIC),        # 1:1 NV12 -> planar
    ("nv12", 640, 360, "nv12", 640, 360, ffi.SWS_BICUBIC),
    ("yuv420p", 128, 72, "yuv420p", 256, 216, ffi.SWS_BICUBLIN),     # bicubic luma, bilinear chroma
    ("nv12", 128, 72, "nv12", 384, 72, ffi.SWS_BICUBIC),             # horizontal only: 1-tap vertical bank
    ("yuv420p", 384, 216, "yuv420p", 192, 108, ffi.SWS_AREA),        # 2x area down-scaling: 2 taps at stride 2
]


@pytest.mark.parametrize("env", [{}, {"FFHIP_CW_OPT": "0"}, {"FFHIP_SWS_MFMA": "1"}], ids=["default", "noopt", "mfma"])
@pytest.mark.parametrize("case", PADDED_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_fast_path_padded_banks(case, env, monkeypatch):
    _run(*case, env=env, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, n=2)


# ---------------------------------------------------------------------------------------------
# packed rgb24 / bgr24 targets on the column walker (k_sws_colwalk_rgb): yuv2rgb_X_c_template
RGB_CASES = [
    ("yuv420p", 128, 72, "rgb24", 256, 144, ffi.SWS_BICUBIC),
    ("yuv420p", 128, 72, "bgr24", 256, 144, ffi.SWS_BICUBIC),
    ("nv12", 192, 108, "rgb24", 384, 216, ffi.SWS_BICUBIC),
    ("nv21", 192, 108, "bgr24", 384, 216, ffi.SWS_BICUBIC),
    ("nv12", 64, 40, "rgb24", 192, 104, ffi.SWS_BICUBIC),             # 3x / 2.6x
    ("yuv420p", 1048, 600, "rgb24", 2096, 1416, ffi.SWS_BICUBIC),     # several column blocks and strips, ragged last block
    ("nv12", 32, 16, "bgr24", 64, 32, ffi.SWS_BICUBIC),               # a single partial wave
    ("yuv420p", 96, 64, "rgb24", 192, 200, ffi.SWS_BICUBIC),          # anisotropic: 2x / 3.125x
    ("yuv420p", 96, 64, "rgb24", 136, 200, ffi.SWS_BICUBIC),          # 1.42x: byte-aligned source spans
    ("nv12", 128, 72, "bgr24", 192, 108, ffi.SWS_BICUBIC),            # 1.5x
    ("yuv420p", 202, 120, "rgb24", 456, 270, ffi.SWS_BICUBIC),        # srcW % 4 != 0, odd chroma width
    ("yuv420p", 176, 144, "bgr24", 176, 144, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND | ffi.SWS_BITEXACT),  # unscaled: 1-tap luma, 4-tap chroma
    ("nv12", 192, 108, "rgb24", 192, 216, ffi.SWS_BICUBIC),           # vertical only: 1-tap horizontal banks
    ("yuv420p", 128, 72, "rgba", 256, 144, ffi.SWS_BICUBIC),          # 32-bit packed targets, alpha = 255
    ("nv12", 192, 108, "bgra", 384, 216, ffi.SWS_BICUBIC),
    ("nv21", 128, 72, "argb", 192, 108, ffi.SWS_BICUBIC),
    ("yuv420p", 1048, 600, "abgr", 2096, 1416, ffi.SWS_BICUBIC),
    # the other two templates of packed_vscale on the same walker: yuv2rgb_2 (2-tap banks, rows sum to 4096: no rounding term)
    ("yuv420p", 128, 72, "rgb24", 256, 144, ffi.SWS_BILINEAR),
    ("nv12", 192, 108, "bgra", 384, 216, ffi.SWS_BILINEAR),
    ("yuv420p", 96, 64, "bgr24", 136, 200, ffi.SWS_BILINEAR),
    ("yuv420p", 1048, 600, "rgb24", 2096, 1416, ffi.SWS_BILINEAR),
    # ... and yuv2rgb_1 (1-tap luma; chroma 1 or 2 taps): horizontal-only scaling of 4:2:0, point sampling
    ("yuv420p", 128, 72, "rgb24", 256, 72, ffi.SWS_BILINEAR),
    ("nv12", 128, 72, "rgb24", 256, 72, ffi.SWS_BICUBIC),
    ("yuv420p", 128, 72, "rgb24", 256, 144, ffi.SWS_POINT),
]


@pytest.mark.parametrize("env", [{}, {"FFHIP_CWRGB_DIRECT": "1"}, {"FFHIP_SWS_FAST": "0"}, {"FFHIP_CWRGB_STRIP": "64"}, {"FFHIP_CWRGB_STRIP": "9"}],
                         ids=["default", "direct", "tiled", "strip64", "strip9"])
@pytest.mark.parametrize("case", RGB_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_fast_path_rgb(case, env, monkeypatch):
    _run(*case, env=env, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF)


def test_fast_path_rgb_full_size(monkeypatch):
    _run("yuv420p", 1920, 1080, "rgb24", 3840, 2160, ffi.SWS_BICUBIC, monkeypatch=monkeypatch, n=2, seed=79)


# exact 2x of 4:2:0 (planar, NV12, NV21) into packed RGB: the static-schedule kernel with the writer fused (k_sws_up2_rgb, sws_up2rgb.hip):
# yuv2rgb_X (bicubic: 4 x 4 taps) and yuv2rgb_2 (bilinear: no rounding term), every packed layout, one group per row .. several
# column blocks with a ragged last one, one strip .. several (the strip length is a knob of the measure build), a single chroma window
UP2RGB_CASES = [
    ("yuv420p", 16, 8, "rgb24", 32, 16, ffi.SWS_BICUBIC),             # two groups: both lanes sit at a row end
    ("yuv420p", 128, 72, "rgb24", 256, 144, ffi.SWS_BICUBIC),
    ("yuv420p", 128, 72, "bgr24", 256, 144, ffi.SWS_BICUBIC),
    ("yuv420p", 136, 70, "argb", 272, 140, ffi.SWS_BICUBIC),          # 34 groups, odd chroma height
    ("yuv420p", 128, 72, "rgba", 256, 144, ffi.SWS_BICUBIC),
    ("yuv420p", 264, 50, "abgr", 528, 100, ffi.SWS_BICUBIC),          # a full wave + a lane pair
    ("yuv420p", 128, 72, "bgra", 256, 144, ffi.SWS_BICUBIC),
    ("yuv420p", 1048, 600, "rgb24", 2096, 1416 - 216, ffi.SWS_BICUBIC),  # several column blocks and strips, ragged last block
    ("yuv420p", 128, 72, "rgb24", 256, 144, ffi.SWS_BILINEAR),
    ("yuv420p", 520, 130, "bgr24", 1040, 260, ffi.SWS_BILINEAR),
    ("yuv420p", 128, 72, "rgba", 256, 144, ffi.SWS_BILINEAR),
    # byte-interleaved chroma: a lane's six (u, v) pairs are the 12 bytes under its luma span
    ("nv12", 16, 8, "bgr24", 32, 16, ffi.SWS_BICUBIC),
    ("nv12", 192, 108, "rgb24", 384, 216, ffi.SWS_BICUBIC),
    ("nv21", 192, 108, "bgr24", 384, 216, ffi.SWS_BICUBIC),
    ("nv12", 264, 50, "bgra", 528, 100, ffi.SWS_BICUBIC),
    ("nv21", 1048, 600, "argb", 2096, 1200, ffi.SWS_BICUBIC),
    ("nv12", 520, 130, "rgb24", 1040, 260, ffi.SWS_BILINEAR),
]


@pytest.mark.parametrize("env", [{}, {"FFHIP_UP2RGB_STEPS": "6"}, {"FFHIP_UP2RGB_STEPS": "500"}, {"FFHIP_SWS_UP2RGB": "v1"}, {"FFHIP_SWS_UP2RGB": "v2"},
                                 {"FFHIP_SWS_UP2RGB": "v3"}, {"FFHIP_UP2RGB_FPP": "1"}, {"FFHIP_UP2RGB_FPP": "4"}],
                         ids=["default", "strip6", "one_strip", "plain_stores", "direct", "pieces8", "fpp1", "fpp4"])
@pytest.mark.parametrize("case", UP2RGB_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_up2rgb(case, env, monkeypatch):
    _run(*case, env=env, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, need="up2rgb")


def test_up2rgb_full_size(monkeypatch):
    _run("yuv420p", 1920, 1080, "rgb24", 3840, 2160, ffi.SWS_BICUBIC, monkeypatch=monkeypatch, n=2, seed=81, need="up2rgb")


def test_up2rgb_is_not_taken_where_it_does_not_apply():
    from ffmpeg_amd import swscale as S
    for sf, sw, sh, df, dw, dh in (("yuv444p", 128, 72, "rgb24", 256, 144), ("yuv420p", 132, 72, "rgb24", 264, 144),
                                   ("yuv420p", 128, 72, "rgb24", 256, 216), ("yuv422p", 128, 72, "rgb24", 256, 144)):
        ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], ffi.SWS_BICUBIC)
        assert not ctx.up2rgb_path, (sf, sw, sh, df, dw, dh)


# 4:2:0 into packed RGB at the SOURCE'S SIZE through the scaler (sws_eqrgb.hip): what sws_scale() runs for NV12 -> rgb24 (no table
# converter exists for semi-planar sources) and for yuv420p under SWS_ACCURATE_RND: one-tap luma, the chroma lines brought up to a line
# per output line by the 4-tap vertical bank
EQRGB_CASES = [
    ("nv12", 64, 8, "rgb24"), ("nv12", 192, 108, "rgb24"), ("nv21", 192, 108, "bgr24"), ("nv12", 264, 50, "bgra"), ("nv21", 136, 70, "argb"),
    ("nv12", 1048, 600, "rgba"), ("nv12", 1048, 600, "rgb24"), ("yuv420p", 192, 108, "rgb24"), ("yuv420p", 520, 130, "abgr"),
    ("yuv420p", 1048, 600, "bgr24"),
]


@pytest.mark.parametrize("env", [{}, {"FFHIP_EQRGB_STEPS": "3"}, {"FFHIP_EQRGB_STEPS": "900"}, {"FFHIP_EQRGB_FPP": "1"}, {"FFHIP_EQRGB_FPP": "4"}],
                         ids=["default", "strip3", "one_strip", "fpp1", "fpp4"])
@pytest.mark.parametrize("case", EQRGB_CASES, ids=lambda c: "%s_%dx%d_%s" % c)
def test_eqrgb(case, env, monkeypatch):
    from ffmpeg_amd import swscale as S
    sf, w, h, df = case
    fl = ffi.SWS_BICUBIC | (ffi.SWS_ACCURATE_RND if sf == "yuv420p" else 0)   # (plain yuv420p gets the table converter: test_gpu_sws.py)
    ctx = S.SwsContext(w, h, PIX[sf], w, h, PIX[df], fl)
    assert ctx.paths & 128, "case does not reach the equal-size kernel (paths %d)" % ctx.paths
    ctx.close()
    _run(sf, w, h, df, w, h, fl, env=env, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, need="any")


@pytest.mark.parametrize("flags", [ffi.SWS_BILINEAR, ffi.SWS_POINT, ffi.SWS_BICUBIC | ffi.SWS_BITEXACT], ids=["bilinear", "point", "bitexact"])
def test_eqrgb_other_vertical_chroma_banks(flags, monkeypatch):
    """2-tap (bilinear: the reference runs yuv2rgb_1 with the row's chroma weight, libswscale/vscale.c:140-160, output.c:1883-1939) and
    1-tap vertical chroma banks, when they fit the regular windows: same kernel, or the column walker — either way the reference's bytes"""
    _run("nv12", 192, 108, "rgb24", 192, 108, flags, monkeypatch=monkeypatch, seed=87, need="any")
    _run("nv21", 264, 50, "bgra", 264, 50, flags, monkeypatch=monkeypatch, seed=88, need="any")


def test_eqrgb_full_size_and_switched_off(monkeypatch):
    _run("nv12", 1920, 1080, "rgb24", 1920, 1080, ffi.SWS_BICUBIC, monkeypatch=monkeypatch, n=5, seed=85, need="any")
    _run("nv12", 192, 108, "rgb24", 192, 108, ffi.SWS_BICUBIC, env={"FFHIP_SWS_EQRGB": "0"}, monkeypatch=monkeypatch, seed=86, need="fast")


def test_eqrgb_is_not_taken_where_it_does_not_apply():
    from ffmpeg_amd import swscale as S
    for sf, w, h, df, fl in (("nv12", 196, 108, "rgb24", ffi.SWS_BICUBIC),
                             ("yuv422p", 192, 108, "rgb24", ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND)):
        ctx = S.SwsContext(w, h, PIX[sf], w, h, PIX[df], fl)
        assert not ctx.paths & 128, (sf, w, h, df, fl)


# down-scaling (and every other ratio whose banks the column walker does not take) INTO packed RGB: two stages — the wide-bank walker
# on the context's own banks with the luma stored as unclipped int16 (FFHipLwJob.y16), then the tables' closed form (sws_y16rgb.hip).
# Random bytes through a bicubic down-scale overshoot 0..255 all the time: the int16 plane is what keeps the two stages exact
RGB2_CASES = [
    ("nv12", 384, 216, "rgb24", 192, 104, ffi.SWS_BICUBIC),            # ~2:1, 8 x 8 taps, interleaved chroma in
    ("nv21", 384, 216, "bgr24", 192, 104, ffi.SWS_BICUBIC),
    ("yuv420p", 384, 216, "rgb24", 192, 104, ffi.SWS_BICUBIC),
    ("yuv420p", 640, 360, "bgra", 216, 120, ffi.SWS_BICUBIC),          # ~3:1: 12 taps padded to 16
    ("nv12", 640, 368, "argb", 160, 92, ffi.SWS_BICUBIC),              # 4:1: 16 x 16 taps
    ("yuv420p", 480, 270, "rgba", 320, 180, ffi.SWS_BICUBIC),          # 1.5:1
    ("nv12", 2096, 1416, "rgb24", 1048, 600, ffi.SWS_BICUBIC),         # several column blocks and strips, ragged last block
    ("yuv422p", 384, 216, "abgr", 192, 104, ffi.SWS_BICUBIC),          # 4:2:2 source: chroma down vertically as well
    ("yuv420p", 384, 216, "rgb24", 200, 216, ffi.SWS_BICUBIC),         # horizontal only — vertical luma bank of 1 tap: NOT this path
    ("yuv420p", 640, 360, "rgb24", 160, 92, ffi.SWS_AREA),             # area: 4..5 taps
    ("yuv420p", 384, 216, "rgb24", 192, 104, ffi.SWS_BILINEAR),        # bilinear down: 4 taps that do not fit the column walker's spans
    # exact 2:1: the luma job on the static-schedule kernel (k_sws_down2 with int16 stores), the chroma on the wide walker
    ("nv12", 384, 216, "rgb24", 192, 108, ffi.SWS_BICUBIC),
    ("yuv420p", 2576, 96, "bgra", 1288, 48, ffi.SWS_BICUBIC),          # several column blocks, ragged last one
    ("nv21", 384, 216, "abgr", 192, 108, ffi.SWS_BICUBIC),
    ("yuv422p", 384, 216, "bgr24", 192, 108, ffi.SWS_BICUBIC),
    # even widths that are not multiples of 8: the row's last group is ragged in both stages
    ("nv12", 384, 216, "rgb24", 170, 96, ffi.SWS_BICUBIC),
    ("yuv420p", 384, 216, "bgra", 172, 96, ffi.SWS_BICUBIC),
    ("nv21", 384, 216, "bgr24", 174, 100, ffi.SWS_BICUBIC),
    ("yuv420p", 1080, 480, "argb", 540, 240, ffi.SWS_BICUBIC),         # 2:1 with 540 = 8 * 67 + 4 columns: the luma on k_sws_down2
    ("nv12", 1920, 1080, "rgb24", 854, 480, ffi.SWS_BICUBIC),
    # 17..32-tap banks: a thumbnail, a network input
    ("nv12", 768, 432, "rgb24", 128, 72, ffi.SWS_BICUBIC),
    ("yuv420p", 1920, 1080, "bgr24", 320, 180, ffi.SWS_BICUBIC),
    ("nv12", 1920, 1080, "rgb24", 224, 224, ffi.SWS_BICUBIC),          # 36 taps across
]


@pytest.mark.parametrize("case", RGB2_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_rgb_two_stage(case, monkeypatch):
    from ffmpeg_amd import swscale as S
    sf, sw, sh, df, dw, dh, fl = case
    ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], fl)
    two = bool(ctx.paths & 4) and not ctx.fast_path
    assert two == (dh != sh), "paths %d" % ctx.paths
    ctx.close()
    # (FFHIP_SWS_DOWN2=1: _run() switches the exact-2:1 kernel off for the tests of the general kernels; these want it)
    _run(*case, env={"FFHIP_SWS_DOWN2": "1"}, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, need="any")


@pytest.mark.parametrize("case", [c for c in RGB2_CASES if c[1] == 2 * c[4] and c[2] == 2 * c[5]], ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_rgb_two_stage_exact_half_chroma_on_the_wide_walker(case, monkeypatch):
    """exact 2:1 from a 4:2:0 source: the product runs luma AND chroma (no vertical filter: FFHipDn2Job.v1) in one k_sws_down2 launch;
    FFHIP_SWS_DOWN2=l keeps the chroma on the wide walker beside the luma kernel (the form before) — both must give the reference's bytes"""
    _run(*case, env={"FFHIP_SWS_DOWN2": "l"}, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFF, need="any")


@pytest.mark.parametrize("sf,df,dw,dh", [("nv12", "rgb24", 200, 108), ("nv21", "bgra", 1288, 48), ("yuv420p", "bgr24", 1288, 48), ("yuv420p", "argb", 200, 108),
                                         ("nv12", "abgr", 12, 8), ("yuv420p", "rgba", 24, 8)])
def test_rgb_two_stage_exact_half_one_launch(sf, df, dw, dh, monkeypatch):
    """the single-launch first stage at widths with ragged last lane blocks, the smallest widths it takes, every layout"""
    _run(sf, 2 * dw, 2 * dh, df, dw, dh, ffi.SWS_BICUBIC, env={"FFHIP_SWS_DOWN2": "1"}, monkeypatch=monkeypatch, seed=dw + dh, need="any")


@pytest.mark.parametrize("sf,df,dw,dh", [("yuv420p", "rgb24", 200, 108), ("yuv420p", "bgra", 1288, 48), ("yuv420p", "abgr", 12, 8), ("yuv420p", "bgr24", 1032, 20),
                                         ("yuv420p", "rgba", 260, 16), ("yuv420p", "argb", 2048, 12)])
def test_rgb_exact_half_fused_planar(sf, df, dw, dh, monkeypatch):
    """the fused kernel on planar chroma (12 bytes of each plane per lane and row): lane blocks' borders, ragged blocks, every layout"""
    _run(sf, 2 * dw, 2 * dh, df, dw, dh, ffi.SWS_BICUBIC, env={"FFHIP_SWS_DOWN2": "1"}, monkeypatch=monkeypatch, seed=dw + 5 * dh, need="any")


@pytest.mark.parametrize("sf,df,dw,dh", [("nv12", "rgb24", 200, 108), ("nv21", "bgra", 1288, 48), ("nv12", "bgr24", 192, 108), ("nv21", "argb", 12, 8),
                                         ("nv12", "abgr", 1032, 20), ("nv12", "rgba", 2048, 12), ("yuv420p", "rgb24", 200, 108), ("yuv420p", "bgra", 1288, 48)])
def test_rgb_exact_half_two_stages_kept(sf, df, dw, dh, monkeypatch):
    """the product runs the FUSED k_sws_down2_rgb; FFHIP_SWS_DOWN2=t keeps the two-stage form (one first-stage launch +
    k_y16_rgb with interleaved chroma): both must give the reference's bytes"""
    _run(sf, 2 * dw, 2 * dh, df, dw, dh, ffi.SWS_BICUBIC, env={"FFHIP_SWS_DOWN2": "t"}, monkeypatch=monkeypatch, seed=dw + 3 * dh, need="any")


def test_rgb_two_stage_full_size(monkeypatch):
    _run("nv12", 3840, 2160, "rgb24", 1920, 1080, ffi.SWS_BICUBIC, env={"FFHIP_SWS_DOWN2": "1"}, monkeypatch=monkeypatch, n=2, seed=83, need="any")


def test_rgb_two_stage_switched_off_is_the_tiled_kernel(monkeypatch):
    _run("nv12", 384, 216, "rgb24", 192, 104, ffi.SWS_BICUBIC, env={"FFHIP_SWS_RGB2": "0"}, monkeypatch=monkeypatch, seed=5, need="any")


def test_rgb_two_stage_exact_half_one_kernel_after_the_other(monkeypatch):
    """(the product runs the luma kernel beside the chroma kernel on a second stream; FFHIP_SWS_RGB2=s queues them on one)"""
    _run("nv12", 384, 216, "rgb24", 192, 108, ffi.SWS_BICUBIC, env={"FFHIP_SWS_DOWN2": "1", "FFHIP_SWS_RGB2": "s"}, monkeypatch=monkeypatch, seed=7,
         need="any")


def test_rgb_two_stage_exact_half_with_the_luma_on_the_wide_walker(monkeypatch):
    _run("nv12", 384, 216, "rgb24", 192, 108, ffi.SWS_BICUBIC, env={"FFHIP_SWS_DOWN2": "0"}, monkeypatch=monkeypatch, seed=6, need="any")


@pytest.mark.parametrize("fmts", [("nv12", "rgb24"), ("yuv420p", "bgr24"), ("nv21", "rgb24")])
def test_fast_path_rgb_adversarial_tables(fmts, monkeypatch):
    """arbitrary eligible positions, wide coefficients; chroma vertical bank of dstH rows as packed targets need"""
    sw, sh, dw, dh = 200, 120, 520, 300
    rng = np.random.default_rng(11)
    banks = _adversarial_banks(rng, sw, sh, dw, dh, 0)
    full = _adversarial_banks(rng, sw, sh, dw, 2 * dh, 0)
    banks["vChr"] = full["vChr"]
    # unit-gain coefficients with negative lobes: packed output indexes the reference's 2048-entry luma ramp with the
    # unclamped sample, so gains far above 1 would leave the table (undefined in the reference itself)
    for name, one in (("hLum", 1 << 14), ("hChr", 1 << 14), ("vLum", 1 << 12), ("vChr", 1 << 12)):
        _, pos, fs, n = banks[name]
        g = rng.dirichlet(np.ones(4), n) * 1.2 - 0.05
        f = np.round(g * one).astype(np.int32)
        f[:, 3] += one - f.sum(1)
        banks[name] = (f.astype(np.int16).reshape(-1), pos, fs, n)
    _run(fmts[0], sw, sh, fmts[1], dw, dh, ffi.SWS_BICUBIC, banks=banks, monkeypatch=monkeypatch, n=2, seed=6)


# ---------------------------------------------------------------------------------------------
# wide banks (5..16 taps: down-scaling) on the LDS-backed walker (sws_lwalk.hip)
WIDE_CASES = [
    ("nv12", 384, 216, "nv12", 192, 108, ffi.SWS_BICUBIC),           # 2x down: 8 x 8 taps
    ("nv21", 384, 216, "nv21", 192, 108, ffi.SWS_BICUBIC),
    ("yuv420p", 384, 216, "yuv420p", 192, 108, ffi.SWS_BICUBIC),     # three single-plane jobs
    ("nv12", 384, 216, "yuv420p", 192, 108, ffi.SWS_BICUBIC),        # interleaved -> planar
    ("yuv420p", 384, 216, "nv21", 192, 108, ffi.SWS_BICUBIC),        # planar -> interleaved
    ("nv12", 640, 360, "nv12", 216, 120, ffi.SWS_BICUBIC),           # ~3x down: 12 taps padded to 16
    ("yuv420p", 640, 368, "nv12", 160, 92, ffi.SWS_BICUBIC),         # 4x down: 16 x 16 taps
    ("nv12", 480, 270, "nv12", 320, 180, ffi.SWS_BICUBIC),           # 1.5x down: 6 taps padded to 8
    ("nv12", 2096, 1416, "nv12", 1048, 600, ffi.SWS_BICUBIC),        # several column blocks and strips, ragged last block
    ("nv12", 384, 216, "nv12", 192, 108, ffi.SWS_BILINEAR),          # 4-tap bank whose windows do not fit the column walker
    ("yuv420p", 640, 360, "yuv420p", 160, 90 + 2, ffi.SWS_AREA),       # 4x area: 4..5 taps at stride 4
    ("nv12", 384, 216, "nv12", 512, 108, ffi.SWS_BICUBIC),           # up horizontally, down vertically
    ("yuv420p", 202, 120, "yuv420p", 104, 60, ffi.SWS_BICUBIC),      # odd chroma width (planar)
    ("yuv422p", 480, 270, "yuv444p", 320, 180, ffi.SWS_BICUBIC),     # 4:2:2 -> 4:4:4: chroma up horizontally, down vertically
    # exact 2:1 across several column blocks
    ("yuv420p", 2560, 96, "yuv420p", 1280, 48, ffi.SWS_BICUBIC),     # planes + a planar U/V pair
    ("nv21", 2560, 96, "nv21", 1280, 48, ffi.SWS_BICUBIC),           # interleaved pair, swapped
    ("nv12", 2560, 96, "yuv420p", 1280, 40, ffi.SWS_BICUBIC),        # interleaved in, planar out, another vertical ratio
    ("yuv420p", 2576, 96, "nv12", 1288, 48, ffi.SWS_BICUBIC),        # ragged last block
    # widths that are not multiples of 4 (854 x 480 from 1080p, ...): the last lane of a row stores its one to three columns one by one
    ("nv12", 384, 216, "nv12", 170, 96, ffi.SWS_BICUBIC),             # chroma 85 wide: one (u, v) pair in the last lane
    ("yuv420p", 384, 216, "yuv420p", 170, 96, ffi.SWS_BICUBIC),
    ("nv12", 384, 216, "yuv420p", 171 + 3, 96, ffi.SWS_BICUBIC),      # 174: luma 2 left over, chroma 87
    ("yuv420p", 384, 216, "nv21", 166, 90, ffi.SWS_BICUBIC),
    ("yuv444p", 384, 216, "yuv444p", 173, 97, ffi.SWS_BICUBIC),       # odd in both directions
    ("nv12", 1920, 1080, "nv12", 854, 480, ffi.SWS_BICUBIC),
    # banks of 17..32 taps (round 5): ratios down to about 1/8
    ("nv12", 768, 432, "nv12", 128, 72, ffi.SWS_BICUBIC),             # 6:1: 24 x 24 taps
    ("yuv420p", 640, 360, "yuv420p", 96, 54, ffi.SWS_BICUBIC),        # 6.67:1: 27 taps, planar
    ("nv12", 768, 216, "yuv420p", 128, 108, ffi.SWS_BICUBIC),         # 6:1 across, 2:1 down: 32 x 8 taps
    ("yuv420p", 192, 432, "nv12", 96, 72, ffi.SWS_BICUBIC),           # 2:1 across, 6:1 down: 8 x 32 taps
    ("nv12", 1920, 1080, "nv12", 426, 240, ffi.SWS_BICUBIC),          # 4.5:1, a ragged width
    ("yuv420p", 1920, 1080, "yuv420p", 256, 144, ffi.SWS_BILINEAR),   # 7.5:1 bilinear: 16 taps
    # 33..64 taps across (round 5): a 1080p frame into a 224-wide network input
    ("nv12", 1920, 1080, "nv12", 224, 224, ffi.SWS_BICUBIC),          # 8.6:1 across (36 taps), 4.8:1 down (20)
    ("yuv420p", 1536, 216, "yuv420p", 128, 72, ffi.SWS_BICUBIC),      # 12:1 across (48 taps), 3:1 down
]


@pytest.mark.parametrize("ahead", ["0", "1"], ids=["inrow", "ahead"])
@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_wide_path(case, ahead, monkeypatch):
    _run(*case, env={"FFHIP_LW_AHEAD": ahead}, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, need="wide")


@pytest.mark.parametrize("case", CASES[:7], ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_wide_path_on_narrow_banks(case, monkeypatch):
    """the wide walker forced onto 4-tap up-scaling banks (padded to 8 x 8)"""
    _run(*case, env={"FFHIP_SWS_WIDE": "1", "FFHIP_SWS_FAST": "0"}, monkeypatch=monkeypatch, seed=3, need="wide")


def test_wide_path_full_size(monkeypatch):
    _run("nv12", 3840, 2160, "nv12", 1920, 1080, ffi.SWS_BICUBIC, monkeypatch=monkeypatch, n=2, seed=80, need="wide")


# ---------------------------------------------------------------------------------------------
# exact 3:2 down-scaling on the static-schedule kernel (sws_down32.hip, round 5)
DOWN32_CASES = [
    ("nv12", 384, 216, "nv12", 256, 144, ffi.SWS_BICUBIC),           # 6 x 6 taps, one column block
    ("nv21", 384, 216, "nv21", 256, 144, ffi.SWS_BICUBIC),
    ("nv12", 384, 216, "nv21", 256, 144, ffi.SWS_BICUBIC),           # the pair turned round on the way
    ("yuv420p", 384, 216, "yuv420p", 256, 144, ffi.SWS_BICUBIC),     # three plane jobs
    ("yuv420p", 1560, 96, "yuv420p", 1040, 64, ffi.SWS_BICUBIC),     # several lane blocks, ragged last one (130 / 65 groups)
    ("nv12", 1560, 96, "nv12", 1040, 64, ffi.SWS_BICUBIC),
    ("nv12", 384, 216, "nv12", 256, 144, ffi.SWS_BILINEAR),          # 4 taps inside the 6-tap window
    ("yuv420p", 384, 216, "yuv420p", 256, 144, ffi.SWS_POINT),
    ("yuv422p", 288, 108, "yuv422p", 192, 72, ffi.SWS_BICUBIC),      # a chroma row per luma row
    ("yuv444p", 144, 54, "yuv444p", 96, 36, ffi.SWS_BICUBIC),        # the smallest rows it takes: three groups
    ("nv12", 384, 222, "nv12", 256, 148, ffi.SWS_BICUBIC),           # strips that end inside a period
    ("nv12", 1920, 1080, "nv12", 1280, 720, ffi.SWS_BICUBIC),
]


@pytest.mark.parametrize("case", DOWN32_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_down32(case, monkeypatch):
    _run(*case, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, need="down32")


def test_down32_is_not_taken_for_other_shapes():
    from ffmpeg_amd import swscale as S
    for sf, sw, sh, df, dw, dh in (("nv12", 384, 216, "yuv420p", 256, 144), ("nv12", 378, 216, "nv12", 252, 144), ("nv12", 384, 216, "nv12", 256, 108)):
        ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], ffi.SWS_BICUBIC)   # layouts differ; 252 is not a multiple of 8; 2:1 down
        assert not ctx.paths & 4096, (sf, sw, sh, df, dw, dh)
        ctx.close()


# ---------------------------------------------------------------------------------------------
# exact 3:2 and 4:3 up-scaling on the static-schedule kernel's 8-bit twin (sws_up32.hip, round 6)
UP32_CASES = [
    ("nv12", 144, 96, "nv12", 216, 144, ffi.SWS_BICUBIC),            # three groups per chroma row... (108 columns / 6 = 18) one column block
    ("nv21", 144, 96, "nv21", 216, 144, ffi.SWS_BICUBIC),
    ("yuv420p", 144, 96, "yuv420p", 216, 144, ffi.SWS_BICUBIC),      # three plane jobs
    ("yuv420p", 48, 24, "yuv420p", 72, 36, ffi.SWS_BICUBIC),         # the smallest rows it takes: three groups per chroma row
    ("yuv420p", 1040, 64, "yuv420p", 1560, 96, ffi.SWS_BICUBIC),     # several lane blocks, ragged last one (130 / 65 groups)
    ("nv12", 1040, 64, "nv12", 1560, 96, ffi.SWS_BICUBIC),
    ("nv12", 144, 96, "nv12", 216, 144, ffi.SWS_BILINEAR),           # 2 taps inside the 4-tap window
    ("yuv420p", 144, 96, "yuv420p", 216, 144, ffi.SWS_POINT),
    ("yuv422p", 144, 48, "yuv422p", 216, 72, ffi.SWS_BICUBIC),
    ("yuv444p", 72, 48, "yuv444p", 108, 72, ffi.SWS_BICUBIC),
    ("nv12", 144, 300, "nv12", 216, 450, ffi.SWS_BICUBIC),           # several strips of rows
    ("nv12", 1280, 720, "nv12", 1920, 1080, ffi.SWS_BICUBIC),
    # 4:3: period (3 in, 4 out)
    ("nv12", 144, 108, "nv12", 192, 144, ffi.SWS_BICUBIC),
    ("yuv420p", 144, 108, "yuv420p", 192, 144, ffi.SWS_BICUBIC),
    ("yuv420p", 72, 36, "yuv420p", 96, 48, ffi.SWS_BILINEAR),        # three groups per chroma row
    ("nv21", 792, 66, "nv21", 1056, 88, ffi.SWS_BICUBIC),            # 66 luma groups, 66 chroma groups: a ragged block each
    ("yuv444p", 72, 54, "yuv444p", 96, 72, ffi.SWS_POINT),
    ("nv12", 144, 330, "nv12", 192, 440, ffi.SWS_BICUBIC),
    ("nv12", 1920, 1080, "nv12", 2560, 1440, ffi.SWS_BICUBIC),
]


@pytest.mark.parametrize("case", UP32_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_up32(case, monkeypatch):
    _run(*case, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, need="up32")


def test_up32_is_not_taken_for_other_shapes():
    from ffmpeg_amd import swscale as S
    for sf, sw, sh, df, dw, dh in (("nv12", 144, 96, "yuv420p", 216, 144), ("nv12", 140, 96, "nv12", 210, 144), ("nv12", 144, 96, "nv21", 216, 144)):
        ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], ffi.SWS_BICUBIC)   # layouts differ; 210 is not a multiple of 12; the pair turned round
        assert not ctx.paths & 8192, (sf, sw, sh, df, dw, dh)
        ctx.close()


# ---------------------------------------------------------------------------------------------
# exact 2:1 down-scaling on the static-schedule kernel (sws_down2.hip)
DOWN2_CASES = [
    ("nv12", 384, 216, "nv12", 192, 108, ffi.SWS_BICUBIC),           # 8 x 8 taps, one column block, two strips
    ("nv21", 384, 216, "nv21", 192, 108, ffi.SWS_BICUBIC),
    ("nv12", 384, 216, "nv21", 192, 108, ffi.SWS_BICUBIC),           # the channels change places
    ("nv21", 384, 216, "nv12", 192, 108, ffi.SWS_BICUBIC),
    ("yuv420p", 384, 216, "yuv420p", 192, 108, ffi.SWS_BICUBIC),     # three single-plane jobs
    ("nv12", 384, 216, "nv12", 192, 108, ffi.SWS_BILINEAR),          # 4 taps inside the 8-sample windows
    ("nv12", 48, 16, "nv12", 24, 8, ffi.SWS_BICUBIC),                # the smallest: 6 / 3 groups, both edge lanes adjacent
    ("yuv420p", 48, 48, "yuv420p", 24, 24, ffi.SWS_BICUBIC),         # chroma 12 wide: 3 groups
    ("nv12", 2096, 1416, "nv12", 1048, 708, ffi.SWS_BICUBIC),        # several column blocks and strips, ragged last block
    ("yuv420p", 2576, 96, "yuv420p", 1288, 48, ffi.SWS_BICUBIC),     # ragged last block, planar
    ("nv12", 512, 492, "nv12", 256, 246, ffi.SWS_BICUBIC),           # dstH % 4 != 0: the last step group is cut
    ("nv12", 1032, 16, "nv12", 516, 8, ffi.SWS_BICUBIC),             # chroma: as many source rows as taps, every window touches an edge
    ("nv12", 520, 500, "nv12", 260, 250, ffi.SWS_BICUBIC),           # chroma 130 wide: a width the wide walker does not take
    ("nv12", 1032, 8, "nv12", 516, 4, ffi.SWS_BICUBIC),              # fewer chroma rows than taps
    ("yuv422p", 384, 216, "yuv422p", 192, 108, ffi.SWS_BICUBIC),     # 4:2:2 / 4:4:4 planes
    ("yuv444p", 384, 216, "yuv444p", 192, 108, ffi.SWS_BICUBIC),
]


@pytest.mark.parametrize("env", [{}, {"FFHIP_DN2_XCD": "0"}, {"FFHIP_DN2_STRIP": "8"}],
                         ids=lambda e: ",".join("%s=%s" % (k[6:], v) for k, v in e.items()) or "default")
@pytest.mark.parametrize("case", DOWN2_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_down2(case, env, monkeypatch):
    _run(*case, env=env, monkeypatch=monkeypatch, seed=abs(hash(case)) & 0xFFFF, need="down2")


def test_down2_full_size(monkeypatch):
    _run("nv12", 3840, 2160, "nv12", 1920, 1080, ffi.SWS_BICUBIC, monkeypatch=monkeypatch, n=2, seed=81, need="down2")


def test_down2_not_taken(monkeypatch):
    """shapes the kernel does not serve: interleaved -> planar, a width off the 4-column grid, another ratio"""
    from ffmpeg_amd import swscale as S
    for sf, sw, sh, df, dw, dh in (("nv12", 384, 216, "yuv420p", 192, 108), ("nv12", 388, 216, "nv12", 194, 108),
                                   ("nv12", 384, 216, "nv12", 192, 72)):
        ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], ffi.SWS_BICUBIC)
        assert not ctx.down2_path
        ctx.close()
