"""GPU parity: HIP swscale kernels (through the C-ABI of libffhip.so) vs the oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import PIX, ptr

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _upload(arrs, device="cuda:0", n=1):
    """list of 2-D numpy planes -> list of [n, rows, pitch] cuda tensors (pitch = np stride)"""
    torch = _torch()
    out = []
    for a in arrs:
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        out.append(t.unsqueeze(0).repeat(n, 1, 1).contiguous())
    return out


def _oracle_unscaled(src, w, h, bgr, coeffs=ffi.DEFAULT_COEFFS):
    O = ffi.oracle()
    luts = ffi.OLuts()
    k = ffi.OYuv2RgbCoeffs(*[coeffs[n] for n in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
    O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
    out = np.zeros((h, (3 if int(bgr) < 2 else 4) * w), np.uint8)   # bgr: the oracle's packed layout number (0 rgb24 .. 5 bgra)
    sp, ss = ffi.planes(src)
    O.ffo_yuv420p_to_rgb24(C.byref(luts), w, sp, ss, 0, h, ptr(out), out.strides[0], int(bgr))
    return out


@pytest.mark.parametrize("w,h,pad", [(64, 16, 0), (1920, 1080, 0), (1078, 6, 0), (1076, 4, 3), (30, 2, 1),
                                     (3840, 2160, 0), (2, 2, 0), (18, 4, 0)])
@pytest.mark.parametrize("dst", ["rgb24", "bgr24", "argb", "rgba", "abgr", "bgra"])
def test_unscaled_yuv420p_rgb24(w, h, pad, dst):
    from ffmpeg_amd import swscale as S
    torch = _torch()
    rng = np.random.default_rng(w + h)
    src = ffi.alloc_frame(PIX["yuv420p"], w, h, rng, pad=pad)
    want = _oracle_unscaled(src, w, h, ffi.RGB_LAYOUT[PIX[dst]])
    bw = want.shape[1]
    ctx = S.SwsContext(w, h, PIX["yuv420p"], w, h, PIX[dst], S.SWS_BICUBIC)
    dsrc = _upload(src)
    ddst = [torch.zeros((1, h, bw + pad), dtype=torch.uint8, device="cuda:0")]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    got = ddst[0][0, :, :bw].cpu().numpy()
    assert np.array_equal(got, want)
    # host-pointer SwsFunc face, including a 2-line aligned slice
    hd = np.zeros((h, bw + pad), np.uint8)
    assert ctx.scale(src, [hd]) == h
    assert np.array_equal(hd[:, :bw], want)
    if h >= 8:
        hd2 = np.zeros_like(hd)
        y0, hh = 2, 4
        sl = [src[0][y0:], src[1][y0 // 2:], src[2][y0 // 2:]]
        assert ctx.scale(sl, [hd2], y0, hh) == hh
        assert np.array_equal(hd2[y0:y0 + hh, :bw], want[y0:y0 + hh]) and not hd2[:y0].any() and not hd2[y0 + hh:].any()
    ctx.close()


@pytest.mark.parametrize("dst", ["rgb24", "bgr24"])
def test_unscaled_launch_tuner_keeps_the_bytes(dst):
    """round 6: large launches of the table converter (>= 64 MiB of 24-bit pixels) choose their workgroup numbering per context — the first
    eight alternate plain and eighth-per-XCD between events, the rest take the faster of the two (ffhip_sws_tuned_numbering).  Whatever
    the box decides, every launch writes the oracle's bytes: a 4-frame 3840x2160 batch, eleven calls."""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    w, h, n = 3840, 2160, 4
    rng = np.random.default_rng(77)
    src = ffi.alloc_frame(PIX["yuv420p"], w, h, rng)
    want = torch.from_numpy(_oracle_unscaled(src, w, h, ffi.RGB_LAYOUT[PIX[dst]])).cuda()
    ctx = S.SwsContext(w, h, PIX["yuv420p"], w, h, PIX[dst], S.SWS_BICUBIC)
    assert ctx.tuned_numbering == -1
    dsrc = _upload(src, n=n)
    seen = []
    for call in range(11):
        ddst = [torch.zeros((n, h, 3 * w), dtype=torch.uint8, device="cuda:0")]
        ctx.scale_batch(dsrc, ddst)
        torch.cuda.synchronize()
        for f in range(n):
            assert torch.equal(ddst[0][f], want), "call %d frame %d" % (call, f)
        seen.append(ctx.tuned_numbering)
    assert seen[:8] == [-1] * 8 and seen[-1] in (0, 1), seen      # decided on the ninth large launch
    # a small launch of the same context is not part of the tuning and not affected by it
    one = [torch.zeros((1, h, 3 * w), dtype=torch.uint8, device="cuda:0")]
    ctx.scale_batch([t[:1] for t in dsrc], one)
    torch.cuda.synchronize()
    assert torch.equal(one[0][0], want)
    ctx.close()


def test_host_frames_that_come_back_are_still_right():
    """round 6: the host-pointer face copies tight planes linearly (copy2d(), sws_api.hip: twice the PCIe rate of the 2-D copy it used).
    The bytes must be the oracle's on every call for the SAME host buffers handed over again and again, for buffers that are then freed
    and replaced (other addresses or, as malloc likes to, the same), and for content that changes between calls in place."""
    from ffmpeg_amd import swscale as S
    w, h = 1920, 1080
    ctx = S.SwsContext(w, h, PIX["yuv420p"], w, h, PIX["rgb24"], S.SWS_BICUBIC)
    for round_ in range(3):
        rng = np.random.default_rng(500 + round_)
        src = ffi.alloc_frame(PIX["yuv420p"], w, h, rng)
        hd = np.zeros((h, 3 * w), np.uint8)
        for call in range(6):
            if call == 4:                    # new content in the (by now registered) buffers
                for pl in src:
                    pl[:] = rng.integers(0, 256, pl.shape, dtype=np.uint8)
            want = _oracle_unscaled(src, w, h, ffi.RGB_LAYOUT[PIX["rgb24"]])
            hd[:] = 0
            assert ctx.scale(src, [hd]) == h
            assert np.array_equal(hd, want), "round %d call %d" % (round_, call)
        del src, hd
    ctx.close()


FORM_CASES = [(sf, df) for sf in ("yuv422p", "yuva420p") for df in ("rgb24", "bgr24", "argb", "rgba", "abgr", "bgra", "gbrp")] + [("yuv420p", "gbrp")]


@pytest.mark.parametrize("w,h,pad,n", [(64, 16, 0, 1), (1920, 1080, 0, 2), (354, 10, 3, 3), (30, 4, 1, 1), (2, 2, 0, 1)])
@pytest.mark.parametrize("sf,df", FORM_CASES)
def test_unscaled_converter_forms(sf, df, w, h, pad, n):
    """the table converter's 4:2:2 sources (each luma row with the chroma row of its own), yuva420p (the alpha plane into the alpha byte
    of the 32-bit targets, unread otherwise) and the planar gbrp target (yuv2rgb.c:238-320, 524-553) — batch face, host face, slices"""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    O = ffi.oracle()
    rng = np.random.default_rng(w + h + len(sf) + 3 * len(df))
    src = ffi.alloc_frame(PIX[sf], w, h, rng, pad=pad)
    luts = ffi.OLuts()
    k = ffi.OYuv2RgbCoeffs(*[ffi.DEFAULT_COEFFS[c] for c in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
    O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
    want = ffi.alloc_frame(PIX[df], w, h)
    alpha = sf == "yuva420p" and df in ("argb", "rgba", "abgr", "bgra")
    sp, ss = ffi.planes(src)
    O.ffo_yuv2rgb_unscaled(C.byref(luts), w, sp, ss, 0, h, ffi.planes(want)[0], ffi.planes(want)[1], ffi.RGB_LAYOUT[PIX[df]],
                           int(sf == "yuv422p"), int(alpha))
    ctx = S.SwsContext(w, h, PIX[sf], w, h, PIX[df], S.SWS_BICUBIC)
    dsrc = _upload(src if alpha or sf != "yuva420p" else src[:3], n=n)
    ddst = [torch.zeros((n, a.shape[0], a.shape[1] + pad), dtype=torch.uint8, device="cuda:0") for a in want]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for f in range(n):
        for p, a in enumerate(want):
            assert np.array_equal(ddst[p][f, :, :a.shape[1]].cpu().numpy(), a), (f, p)
    # the SwsFunc face on host pointers: the frame, then 2-line aligned slices
    hd = [np.zeros((a.shape[0], a.shape[1] + pad), np.uint8) for a in want]
    assert ctx.scale(src, hd) == h
    for p, a in enumerate(want):
        assert np.array_equal(hd[p][:, :a.shape[1]], a)
    if h >= 8:
        hd2 = [np.zeros_like(a) for a in hd]
        vs = 0 if sf == "yuv422p" else 1
        for y0, hh in ((0, 2), (2, 4), (6, h - 6)):
            sl = [src[0][y0:], src[1][y0 >> vs:], src[2][y0 >> vs:]] + ([src[3][y0:]] if len(src) > 3 else [])
            assert ctx.scale(sl, hd2, y0, hh) == hh
        for p, a in enumerate(want):
            assert np.array_equal(hd2[p][:, :a.shape[1]], a)
    ctx.close()


def _rgba_cases():
    import test_sws_unscaled_forms_cpu as F
    return F.RGBA_ALPHA_CASES


@pytest.mark.parametrize("sf,sw,sh,df,dw,dh,flags", _rgba_cases())
def test_scaled_source_alpha_into_packed_rgba(sf, sw, sh, df, dw, dh, flags):
    """a YUVA source scaled to the four 32-bit RGB orders: the source's alpha plane goes through the luma banks into the alpha byte
    (yuv2rgba32_{1,2,X}_c and the _full twins; the oracle's form is pinned to the reference in tests/test_sws_unscaled_forms_cpu.py) —
    batch face (3 frames), host face, source slices"""
    import test_sws_unscaled_forms_cpu as F
    from ffmpeg_amd import swscale as S
    torch = _torch()
    rng = np.random.default_rng(sw + dw + len(df) + flags % 97)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=2)
    src[3][:4] = 255
    src[3][4:8] = 0
    want, _ = F.rgba_alpha_oracle(sf, sw, sh, df, dw, dh, flags, src)
    ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    n = 3
    dsrc = _upload(src, n=n)
    ddst = [torch.zeros((n, dh, 4 * dw + 8), dtype=torch.uint8, device="cuda:0")]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for f in range(n):
        got = ddst[0][f, :, :4 * dw].cpu().numpy()
        assert np.array_equal(got, want[0]), "frame %d: %d bytes differ" % (f, (got != want[0]).sum())
    hd = [np.zeros_like(want[0])]
    assert ctx.scale(src, hd) == dh
    assert np.array_equal(hd[0], want[0])
    if sh >= 12:
        hd = [np.zeros_like(want[0])]
        vs = 0 if sf in ("yuva422p", "yuva444p") else 1
        rets = []
        for y0, y1 in ((0, 4), (4, 20), (20, sh)):
            sl = [src[0][y0:], src[1][y0 >> vs:], src[2][y0 >> vs:], src[3][y0:]]
            rets.append(ctx.scale(sl, hd, y0, y1 - y0))
        assert rets == [0, 0, dh] and np.array_equal(hd[0], want[0])
    ctx.close()


def test_unscaled_colorspace_set_on_a_live_context():
    """sws_setColorspaceDetails() after the init (libswscale/utils.c:848-1000): ffhip_sws_yuv2rgb_coeffs + ffhip_sws_set_yuv2rgb"""
    from ffmpeg_amd import swscale as S, _lib
    torch = _torch()
    L = _lib.lib()
    w, h = 640, 360
    rng = np.random.default_rng(9)
    src = ffi.alloc_frame(PIX["yuv420p"], w, h, rng)
    ctx = S.SwsContext(w, h, PIX["yuv420p"], w, h, PIX["rgb24"], S.SWS_BICUBIC)
    ddst = [torch.zeros((1, h, 3 * w), dtype=torch.uint8, device="cuda:0")]
    bt709 = (C.c_int * 4)(117504, 138453, 13954, 34903)   # ff_yuv2rgb_coeffs[SWS_CS_ITU709], libswscale/yuv2rgb.c:47-66
    for inv, full, b, c, s in ((bt709, 1, 0, 1 << 16, 1 << 16), (bt709, 0, 3 << 11, (1 << 16) + 5000, (1 << 16) - 9000)):
        t = _lib.SwsTables()
        _lib.check(L.ffhip_sws_yuv2rgb_coeffs(C.byref(t), inv, full, b, c, s))
        _lib.check(L.ffhip_sws_set_yuv2rgb(ctx._c, C.byref(t)))
        coeffs = dict(cy=t.yuv2rgb_cy, oy=t.yuv2rgb_oy, crv=t.yuv2rgb_crv, cbu=t.yuv2rgb_cbu, cgu=t.yuv2rgb_cgu, cgv=t.yuv2rgb_cgv,
                      yoffs=t.yuv2rgb_yoffs)
        want = _oracle_unscaled(src, w, h, 0, coeffs)
        ctx.scale_batch(_upload(src), ddst)
        assert np.array_equal(ddst[0][0].cpu().numpy(), want)
    ctx.close()


def test_unscaled_exhaustive_uv():
    """all 65536 (U,V) pairs against a moving Y ramp"""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    w, h = 512, 256
    y = ((np.arange(w)[None, :] * 3 + np.arange(h)[:, None] * 7) & 255).astype(np.uint8)
    u = np.tile(np.arange(256, dtype=np.uint8), (h // 2, 1)).copy()
    v = np.tile((np.arange(128, dtype=np.uint8) * 2)[:, None], (1, w // 2)).copy()
    v[:, ::2] += 1
    src = [y, u, v]
    want = _oracle_unscaled(src, w, h, 0)
    ctx = S.SwsContext(w, h, 0, w, h, 2, 4)
    ddst = [torch.zeros((1, h, 3 * w), dtype=torch.uint8, device="cuda:0")]
    ctx.scale_batch(_upload(src), ddst)
    assert np.array_equal(ddst[0][0].cpu().numpy(), want)


SCALE_CASES = [
    ("nv12", 192, 108, "nv12", 384, 216, ffi.SWS_BICUBIC, 0),
    ("nv12", 1920, 1080, "nv12", 3840, 2160, ffi.SWS_BICUBIC, 0),     # BASELINE configs[1], one frame
    ("nv12", 160, 90, "nv12", 100, 62, ffi.SWS_BICUBIC, 3),
    ("nv21", 96, 64, "nv21", 200, 130, ffi.SWS_BILINEAR, 0),
    ("nv12", 96, 64, "nv21", 200, 130, ffi.SWS_BICUBIC, 1),
    ("yuv420p", 128, 72, "yuv420p", 256, 144, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 101, 77, "yuv420p", 333, 191, ffi.SWS_BICUBIC, 5),
    ("nv12", 128, 72, "yuv420p", 64, 36, ffi.SWS_AREA, 0),
    ("yuv420p", 128, 72, "nv12", 128, 90, ffi.SWS_POINT, 0),
    ("nv12", 1920, 1080, "nv12", 640, 360, ffi.SWS_BICUBIC, 0),       # 11-tap downscale
    ("nv12", 192, 108, "nv12", 384, 216, 0x200, 0),                   # lanczos, 6 taps
    ("nv12", 640, 360, "nv12", 640, 360, ffi.SWS_BICUBIC, 0),         # identity banks
    ("yuv420p", 176, 144, "rgb24", 352, 288, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 176, 144, "bgr24", 176, 144, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND | ffi.SWS_BITEXACT, 0),
    ("yuv420p", 176, 144, "rgb24", 176, 288, ffi.SWS_BILINEAR, 0),
    ("nv12", 176, 144, "rgb24", 352, 288, ffi.SWS_BILINEAR, 2),
    ("yuv420p", 352, 288, "rgb24", 120, 90, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 1920, 1080, "rgb24", 3840, 2160, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 176, 144, "rgba", 352, 288, ffi.SWS_BICUBIC, 0),
    ("nv12", 176, 144, "bgra", 352, 288, ffi.SWS_BILINEAR, 2),
    ("yuv420p", 352, 288, "argb", 120, 90, ffi.SWS_BICUBIC, 0),
    ("nv21", 176, 144, "abgr", 176, 144, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND, 0),
    ("yuv420p", 1920, 1080, "rgb24", 1920, 1080, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND, 0),
    # 4:2:2 / 4:4:4 planar: the same kernels on other chroma plane sizes
    ("yuv422p", 128, 72, "yuv422p", 200, 130, ffi.SWS_BICUBIC, 3),
    ("yuv444p", 101, 77, "yuv420p", 333, 191, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 128, 72, "yuv444p", 200, 100, ffi.SWS_BILINEAR, 0),
    ("yuv422p", 161, 90, "nv12", 100, 62, ffi.SWS_BICUBIC, 0),
    ("nv21", 96, 64, "yuv422p", 200, 130, ffi.SWS_BICUBIC, 0),
    ("yuv422p", 128, 72, "yuv444p", 128, 72, ffi.SWS_BICUBIC, 0),     # same size: only the chroma planes are scaled
    ("yuv444p", 1920, 1080, "yuv444p", 1280, 720, ffi.SWS_BICUBIC, 0),
    # 4:2:2 sources to packed RGB (round 3): chroma banks from the source's own chroma plane; equal sizes = one-tap banks
    ("yuv422p", 64, 16, "rgb24", 64, 16, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND, 0),   # (without ACCURATE_RND: the table converter, test_unscaled_converter_forms)
    ("yuv422p", 176, 144, "rgb24", 352, 288, ffi.SWS_BICUBIC, 0),
    ("yuv422p", 176, 144, "bgra", 176, 144, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND, 0),
    ("yuv422p", 352, 288, "argb", 120, 90, ffi.SWS_BILINEAR, 3),
    ("yuv422p", 1920, 1080, "rgb24", 1920, 1080, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND, 0),
    ("yuv422p", 960, 540, "bgr24", 1920, 1080, ffi.SWS_BICUBIC, 0),
    # SWS_FULL_CHR_H_INT (0x2000) on packed RGB targets: asked for, or forced by a 4:4:4 source or an odd width (utils.c:1270-1290):
    # a chroma sample per pixel and the yuv2rgb_full_{1,2,X} writers (output.c:1998-2310)
    ("yuv444p", 64, 36, "rgb24", 128, 72, ffi.SWS_BICUBIC, 0),
    ("yuv444p", 97, 53, "bgr24", 60, 41, ffi.SWS_BICUBIC, 3),
    ("yuv444p", 40, 30, "rgba", 40, 30, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND, 0),
    ("yuv420p", 64, 36, "rgb24", 128, 72, ffi.SWS_BICUBIC | 0x2000, 0),
    ("yuv420p", 66, 38, "bgra", 131, 73, ffi.SWS_BICUBIC, 0),
    ("yuv422p", 64, 36, "rgb24", 96, 54, ffi.SWS_BILINEAR | 0x2000, 0),
    ("nv12", 64, 36, "argb", 127, 71, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND, 0),
    ("yuv444p", 48, 32, "rgb24", 48, 64, ffi.SWS_BILINEAR, 0),
    ("yuv420p", 80, 60, "bgr24", 160, 60, ffi.SWS_POINT | 0x2000, 0),
    ("yuv444p", 1920, 1080, "rgb24", 1280, 720, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 960, 540, "rgb24", 1920, 1080, ffi.SWS_BICUBIC | 0x2000 | ffi.SWS_ACCURATE_RND, 0),
    ("yuv420p", 64, 36, "abgr", 64, 36, ffi.SWS_BICUBIC | 0x2000 | ffi.SWS_ACCURATE_RND, 0),
    # planar 4:4:4 at the source's size (round 5): four one-tap banks + yuv2rgb_full_1 -> the streaming kernel of sws_full444.hip, every
    # layout, ragged row ends (77 + 3 bytes of padding: dword-aligned lines), several lane blocks, the smallest width it takes
    ("yuv444p", 200, 30, "rgb24", 200, 30, ffi.SWS_BICUBIC, 0),
    ("yuv444p", 1032, 16, "bgr24", 1032, 16, ffi.SWS_BILINEAR, 0),
    ("yuv444p", 77, 20, "argb", 77, 20, ffi.SWS_BICUBIC, 3),
    ("yuv444p", 520, 12, "abgr", 520, 12, ffi.SWS_POINT, 0),
    ("yuv444p", 8, 8, "bgra", 8, 8, ffi.SWS_BICUBIC, 0),
    ("yuv444p", 1920, 1080, "rgba", 1920, 1080, ffi.SWS_BICUBIC, 0),
    ("yuv444p", 1921, 4, "rgb24", 1921, 4, ffi.SWS_BICUBIC, 3),
    # 4:2:0 between its planar and semi-planar layouts at the same size (round 5): one-tap banks -> the layout kernel of sws_copy420.hip;
    # rows whose last 16 bytes overlap their neighbours', lines that are not dword-aligned, several lane blocks, the smallest sizes it takes
    ("nv12", 64, 36, "yuv420p", 64, 36, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 64, 36, "nv12", 64, 36, ffi.SWS_BICUBIC, 0),
    ("nv21", 50, 22, "yuv420p", 50, 22, ffi.SWS_BILINEAR, 3),
    ("yuv420p", 50, 22, "nv21", 50, 22, ffi.SWS_POINT, 1),
    ("nv12", 38, 10, "nv21", 38, 10, ffi.SWS_BICUBIC, 0),
    ("nv21", 32, 8, "nv21", 32, 8, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 34, 6, "yuv420p", 34, 6, ffi.SWS_BICUBIC, 5),
    ("nv12", 2100, 24, "yuv420p", 2100, 24, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 2100, 24, "nv12", 2100, 24, ffi.SWS_BICUBIC, 2),
    ("nv12", 1920, 1080, "yuv420p", 1920, 1080, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 1920, 1080, "nv21", 1920, 1080, ffi.SWS_BICUBIC, 0),
    # planar 4:4:4 -> 4:2:0 at the same size: the luma copied, the chroma planes on the exact-2:1 kernel
    ("yuv444p", 64, 36, "yuv420p", 64, 36, ffi.SWS_BICUBIC, 0),
    ("yuv444p", 200, 50, "yuv420p", 200, 50, ffi.SWS_BILINEAR, 0),
    ("yuv444p", 1048, 24, "yuv420p", 1048, 24, ffi.SWS_BICUBIC, 0),
    ("yuv444p", 1920, 1080, "yuv420p", 1920, 1080, ffi.SWS_BICUBIC, 0),
    ("yuv444p", 66, 38, "yuv420p", 66, 38, ffi.SWS_BICUBIC, 2),      # 33 chroma columns: not this path
    # planar 4:2:0 -> 4:4:4 at the same size: the luma copied, the chroma planes on the exact-2x kernel
    ("yuv420p", 64, 36, "yuv444p", 64, 36, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 200, 50, "yuv444p", 200, 50, ffi.SWS_BILINEAR, 0),
    ("yuv420p", 1048, 24, "yuv444p", 1048, 24, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 1920, 1080, "yuv444p", 1920, 1080, ffi.SWS_BICUBIC, 0),
    ("yuv420p", 68, 38, "yuv444p", 68, 38, ffi.SWS_POINT, 0),
]


@pytest.mark.parametrize("case", SCALE_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x_p%d" % c)
def test_scaled(case):
    from ffmpeg_amd import swscale as S
    torch = _torch()
    sf, sw, sh, df, dw, dh, flags, pad = case
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=pad)
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    assert not ht.unscaled_yuv2rgb
    t = ffi.make_otables(sw, sh, PIX[sf], dw, dh, PIX[df], flags, ht.banks(), ht.coeffs(), full=ht.full())
    want = ffi.alloc_frame(PIX[df], dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    n = 3
    dsrc = _upload(src, n=n)
    ddst = [torch.zeros((n,) + a.shape[:1] + (a.shape[1] + pad,), dtype=torch.uint8, device="cuda:0") for a in want]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for f in range(n):
        for p, a in enumerate(want):
            got = ddst[p][f, :, :a.shape[1]].cpu().numpy()
            assert np.array_equal(got, a), "frame %d plane %d: %d mismatches" % (f, p, (got != a).sum())
    if sw * sh <= 400 * 400:
        hd = [np.zeros_like(a) for a in want]
        assert ctx.scale(src, hd) == dh
        for a, b in zip(hd, want):
            assert np.array_equal(a, b)
        # the result does not depend on how the source is sliced (tools/scale_slice_test.c): three in-order slices; output lines
        # are reported once the frame is complete
        if sh >= 12:
            hd = [np.zeros_like(a) for a in want]
            cuts = [0, 4, 4 + 2 * ((sh - 4) // 4), sh]
            rets = []
            for y0, y1 in zip(cuts[:-1], cuts[1:]):
                vs = 0 if sf in ("yuv422p", "yuv444p") else 1
                sl = [src[0][y0:]] + [a[y0 >> vs:] for a in src[1:]]
                rets.append(ctx.scale(sl, hd, y0, y1 - y0))
            assert rets == [0, 0, dh]
            for a, b in zip(hd, want):
                assert np.array_equal(a, b)
            with pytest.raises(RuntimeError, match="out of order"):
                ctx.scale([src[0][8:]] + [a[8 >> vs:] for a in src[1:]], hd, 8, 2)
    ctx.close()


def test_420_layouts_take_the_layout_kernel():
    from ffmpeg_amd import swscale as S
    for sf, df in (("nv12", "yuv420p"), ("yuv420p", "nv12"), ("nv21", "nv12")):
        ctx = S.SwsContext(64, 36, PIX[sf], 64, 36, PIX[df], ffi.SWS_BICUBIC)
        assert ctx.paths & 512, ctx.paths
        ctx.close()
    ctx = S.SwsContext(64, 36, PIX["nv12"], 64, 36, PIX["yuv422p"], ffi.SWS_BICUBIC)   # another subsampling: chroma is scaled
    assert not ctx.paths & 512
    ctx.close()
    ctx = S.SwsContext(24, 16, PIX["nv12"], 24, 16, PIX["yuv420p"], ffi.SWS_BICUBIC)   # chroma rows of 12 bytes: the older kernels
    assert not ctx.paths & 512
    ctx.close()


def test_444_equal_size_takes_the_streaming_kernel():
    from ffmpeg_amd import swscale as S
    for df in ("rgb24", "bgra"):
        ctx = S.SwsContext(200, 30, PIX["yuv444p"], 200, 30, PIX[df], ffi.SWS_BICUBIC)
        assert ctx.paths & 256, ctx.paths
        ctx.close()
    ctx = S.SwsContext(200, 30, PIX["yuv444p"], 100, 30, PIX["rgb24"], ffi.SWS_BICUBIC)   # scaled: the general full-chroma kernel
    assert not ctx.paths & 256
    ctx.close()


@pytest.mark.parametrize("name,base", [("yuvj420p", "yuv420p"), ("yuvj444p", "yuv444p")])
def test_full_range_twins(name, base):
    """yuvjXXXp on both sides runs as the base formats (no range conversion between equal ranges)"""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    sw, sh, dw, dh = 96, 54, 192, 108
    rng = np.random.default_rng(12)
    src = ffi.alloc_frame(PIX[base], sw, sh, rng)
    ht = S.HostTables(sw, sh, PIX[base], dw, dh, PIX[base], ffi.SWS_BICUBIC)
    t = ffi.make_otables(sw, sh, PIX[base], dw, dh, PIX[base], ffi.SWS_BICUBIC, ht.banks(), ht.coeffs())
    want = ffi.alloc_frame(PIX[base], dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    ctx = S.SwsContext(sw, sh, S.PIX_FMT[name], dw, dh, S.PIX_FMT[name], ffi.SWS_BICUBIC)
    dsrc = _upload(src, n=2)
    ddst = [torch.zeros((2,) + a.shape, dtype=torch.uint8, device="cuda:0") for a in want]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for p, a in enumerate(want):
        assert np.array_equal(ddst[p][1].cpu().numpy(), a)
    ctx.close()
    # a J format on one side only is a range conversion (test_range_conversion below; packed RGB targets: test_full_range_yuv_to_rgb)


def test_from_tables_dropin():
    """drop-in construction: the banks are handed over (as FFmpeg would), not generated by us"""
    from ffmpeg_amd import swscale as S, _lib
    torch = _torch()
    sw, sh, dw, dh = 96, 64, 192, 128
    ht = S.HostTables(sw, sh, 23, dw, dh, 23, 4)
    ctx = S.SwsContext(sw, sh, 23, dw, dh, 23, 4, tables=ht.t)
    rng = np.random.default_rng(5)
    src = ffi.alloc_frame(23, sw, sh, rng)
    t = ffi.make_otables(sw, sh, 23, dw, dh, 23, 4, ht.banks(), ht.coeffs())
    want = ffi.alloc_frame(23, dw, dh)
    sp, ss = ffi.planes(src); dp, ds = ffi.planes(want)
    ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds)
    ddst = [torch.zeros((1,) + a.shape, dtype=torch.uint8, device="cuda:0") for a in want]
    ctx.scale_batch(_upload(src), ddst)
    for p, a in enumerate(want):
        assert np.array_equal(ddst[p][0].cpu().numpy(), a)


@pytest.mark.parametrize("fs", [1, 4, 8, 16, 40])
def test_hscale_line_face(fs):
    """checkasm-style adversarial coefficients through the per-line face"""
    from ffmpeg_amd import _lib
    torch = _torch()
    L = _lib.lib()
    dstW, srcW, nlines = 512, 560, 3
    rng = np.random.default_rng(fs)
    src = rng.integers(0, 256, (nlines, srcW), dtype=np.uint8)
    filt = rng.integers(-(1 << 14), 1 << 14, (dstW, fs)).astype(np.int16)
    filt[::3] = -((1 << 14) // max(fs - 1, 1))
    filt[::3, 0] = (1 << 15) - 1
    pos = np.sort(rng.integers(0, srcW - fs, dstW)).astype(np.int32)
    want = np.zeros((nlines, dstW), np.int16)
    for l in range(nlines):
        ffi.oracle().ffo_hscale8to15(ptr(want[l], ffi.i16p), dstW, ptr(src[l]), ptr(filt, ffi.i16p), ptr(pos, ffi.i32p), fs)
    d_src = torch.from_numpy(src).cuda(); d_f = torch.from_numpy(filt).cuda(); d_p = torch.from_numpy(pos).cuda()
    d_out = torch.zeros((nlines, dstW), dtype=torch.int16, device="cuda:0")
    _lib.check(L.ffhip_sws_hscale8to15_dev(d_out.data_ptr(), dstW, dstW * 2, d_src.data_ptr(), srcW, nlines,
                                           d_f.data_ptr(), d_p.data_ptr(), fs, torch.cuda.current_stream().cuda_stream))
    assert np.array_equal(d_out.cpu().numpy(), want)


@pytest.mark.parametrize("fs", [1, 2, 4, 16])
def test_vscale_line_face(fs):
    from ffmpeg_amd import _lib
    torch = _torch()
    L = _lib.lib()
    dstW = 333
    rng = np.random.default_rng(fs + 9)
    lines = rng.integers(-32768, 32768, (fs, dstW + 3)).astype(np.int16)
    filt = rng.integers(-4096, 8192, fs).astype(np.int16)
    dither = rng.integers(0, 128, 8, dtype=np.uint8)
    rows = (ffi.i16p * fs)(*[ptr(lines[j], ffi.i16p) for j in range(fs)])
    for off in (0, 3):
        want = np.zeros(dstW, np.uint8)
        if fs == 1:
            ffi.oracle().ffo_yuv2plane1_8(rows[0], ptr(want), dstW, ptr(dither), off)
        else:
            ffi.oracle().ffo_yuv2planeX8(ptr(filt, ffi.i16p), fs, rows, ptr(want), dstW, ptr(dither), off)
        d_l = torch.from_numpy(lines).cuda(); d_f = torch.from_numpy(filt).cuda(); d_d = torch.from_numpy(dither).cuda()
        d_o = torch.zeros(dstW, dtype=torch.uint8, device="cuda:0")
        _lib.check(L.ffhip_sws_yuv2planeX8_dev(d_f.data_ptr(), fs, d_l.data_ptr(), lines.strides[0], d_o.data_ptr(), dstW,
                                               d_d.data_ptr(), off, torch.cuda.current_stream().cuda_stream))
        assert np.array_equal(d_o.cpu().numpy(), want)


RANGE_CASES = [("yuvj420p", 64, 40, "yuv420p", 160, 88, ffi.SWS_BICUBIC), ("yuv420p", 64, 40, "yuvj420p", 160, 88, ffi.SWS_BICUBIC),
               ("yuvj420p", 96, 54, "yuv420p", 96, 54, ffi.SWS_BICUBIC), ("yuv420p", 96, 54, "yuvj420p", 96, 54, ffi.SWS_BILINEAR),
               ("yuvj444p", 64, 40, "yuv422p", 48, 30, ffi.SWS_BICUBIC), ("yuv422p", 80, 40, "yuvj420p", 120, 90, ffi.SWS_BILINEAR),
               ("yuvj420p", 64, 40, "nv12", 128, 80, ffi.SWS_BICUBIC), ("nv12", 64, 40, "yuvj420p", 100, 60, ffi.SWS_BICUBIC),
               ("yuvj420p", 200, 120, "yuv420p", 50, 30, ffi.SWS_BICUBIC), ("yuvj420p", 1920, 1080, "yuv420p", 3840, 2160, ffi.SWS_BICUBIC),
               ("yuv420p", 1920, 1080, "yuvj420p", 1280, 720, ffi.SWS_BICUBIC), ("yuv420p", 320, 180, "yuvj420p", 640, 360, ffi.SWS_BICUBIC),
               ("yuvj420p", 176, 144, "yuv420p", 352, 288, ffi.SWS_BILINEAR), ("yuv444p", 64, 40, "yuvj444p", 128, 80, ffi.SWS_BICUBIC)]


@pytest.mark.parametrize("case", RANGE_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_range_conversion(case):
    """a full-range (J) format on one side: lum / chrRangeToJpeg_c, ...FromJpeg_c on the horizontal intermediates (libswscale/
    swscale.c:160-207) — HIP == oracle (pinned to the reference by tests/test_oracle_vs_ref.py::test_range_conversion), extreme samples
    included; these contexts run on the general tiled kernel, except exact 2x between like layouts: the static-schedule kernel with
    the range stage between its passes (k_sws_up2<., ., 0, 1>, round 4)"""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    sf, sw, sh, df, dw, dh, flags = case
    base = {"yuvj420p": "yuv420p", "yuvj422p": "yuv422p", "yuvj444p": "yuv444p"}
    bs, bd = base.get(sf, sf), base.get(df, df)
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    src = ffi.alloc_frame(PIX[bs], sw, sh, rng, pad=0)
    for pl in src:
        pl[::5, : pl.shape[1] // 2] = 255
        pl[3::7, pl.shape[1] // 3:] = 0
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    ranges = (int(sf in base), int(df in base))
    assert (ht.t.src_range, ht.t.dst_range) == ranges
    t = ffi.make_otables(sw, sh, PIX[bs], dw, dh, PIX[bd], flags, ht.banks(), ht.coeffs(), ranges=ranges)
    want = ffi.alloc_frame(PIX[bd], dw, dh)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    up2 = dw == 2 * sw and dh == 2 * sh and (bs == "nv12") == (bd == "nv12") and not (sw & 7) and sw >= 16
    assert bool(ctx.up2_path) == up2, (ctx.fast_path, ctx.up2_path)
    n = 2
    dsrc = _upload(src, n=n)
    ddst = [torch.zeros((n,) + a.shape, dtype=torch.uint8, device="cuda:0") for a in want]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for f in range(n):
        for p, a in enumerate(want):
            got = ddst[p][f].cpu().numpy()
            assert np.array_equal(got, a), "frame %d plane %d: %d mismatches" % (f, p, (got != a).sum())
    # ... and differs from the same conversion without the range change (the stage is not a no-op)
    ctx2 = S.SwsContext(sw, sh, PIX[bs], dw, dh, PIX[bd], flags)
    d2 = [torch.zeros_like(x) for x in ddst]
    ctx2.scale_batch(dsrc, d2)
    torch.cuda.synchronize()
    assert not torch.equal(d2[0], ddst[0])


@pytest.mark.parametrize("dst", ["rgb24", "bgra"])
@pytest.mark.parametrize("sw,sh,dw,dh,flags", [(64, 16, 64, 16, ffi.SWS_BICUBIC), (1920, 1080, 1920, 1080, ffi.SWS_BICUBIC), (64, 40, 160, 88, ffi.SWS_BICUBIC),
                                               (96, 54, 48, 28, ffi.SWS_BILINEAR), (176, 144, 352, 288, ffi.SWS_BICUBIC),
                                               (64, 40, 64, 40, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND)])
def test_full_range_yuv_to_rgb(dst, sw, sh, dw, dh, flags):
    """yuvj420p -> packed RGB: the full-range yuv2rgb coefficients (ff_yuv2rgb_c_init_tables' fullRange branch) in the unscaled kernel
    and in the scaled RGB kernels; HIP == oracle (pinned to the reference by tests/test_oracle_vs_ref.py::test_full_range_yuv_to_rgb)"""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    rng = np.random.default_rng(sw + dw + len(dst))
    src = ffi.alloc_frame(PIX["yuv420p"], sw, sh, rng, pad=0)
    for pl in src:
        pl[::5, : pl.shape[1] // 2] = 255
        pl[3::7, pl.shape[1] // 3:] = 0
    ht = S.HostTables(sw, sh, PIX["yuvj420p"], dw, dh, PIX[dst], flags)
    co = ht.coeffs()
    want = ffi.alloc_frame(PIX[dst], dw, dh)
    sp, ss = ffi.planes(src)
    O = ffi.oracle()
    if ht.unscaled_yuv2rgb:
        luts = ffi.OLuts()
        k = ffi.OYuv2RgbCoeffs(*[co[n] for n in ("cy", "oy", "crv", "cbu", "cgu", "cgv", "yoffs")])
        O.ffo_yuv2rgb_luts_init(C.byref(luts), C.byref(k))
        O.ffo_yuv420p_to_rgb24(C.byref(luts), sw, sp, ss, 0, sh, ptr(want[0]), want[0].strides[0], ffi.RGB_LAYOUT[PIX[dst]])
    else:
        t = ffi.make_otables(sw, sh, PIX["yuv420p"], dw, dh, PIX[dst], flags, ht.banks(), co)
        dp, ds = ffi.planes(want)
        assert O.ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    ctx = S.SwsContext(sw, sh, PIX["yuvj420p"], dw, dh, PIX[dst], flags)
    n = 2
    dsrc = _upload(src, n=n)
    ddst = [torch.zeros((n,) + want[0].shape, dtype=torch.uint8, device="cuda:0")]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for f in range(n):
        got = ddst[0][f].cpu().numpy()
        assert np.array_equal(got, want[0]), "frame %d: %d mismatches" % (f, (got != want[0]).sum())
    # ... and differs from the limited-range conversion of the same bytes
    ctx2 = S.SwsContext(sw, sh, PIX["yuv420p"], dw, dh, PIX[dst], flags)
    d2 = [torch.zeros_like(ddst[0])]
    ctx2.scale_batch(dsrc, d2)
    torch.cuda.synchronize()
    assert not torch.equal(d2[0], ddst[0])


ALPHA2_CASES = [("yuva420p", 64, 36, "yuva420p", 128, 72, ffi.SWS_BICUBIC, 0), ("yuva420p", 65, 37, "yuva444p", 40, 30, ffi.SWS_BILINEAR, 5),
                ("yuva444p", 64, 36, "yuva422p", 96, 54, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND, 0), ("yuva422p", 66, 38, "yuva420p", 33, 19, ffi.SWS_BILINEAR, 0),
                ("yuva420p", 960, 540, "yuva420p", 1920, 1080, ffi.SWS_BICUBIC, 0), ("yuva420p", 1920, 1080, "yuva420p", 960, 540, ffi.SWS_BICUBIC, 0),
                ("yuva420p", 640, 360, "yuva420p", 1000, 562, 0x200, 3)]        # SWS_LANCZOS: wide banks


@pytest.mark.parametrize("case", ALPHA2_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x_p%d" % c)
def test_alpha_on_both_sides(case):
    """planar YUVA -> planar YUVA: the alpha plane is scaled by the luma banks (lum_h_scale / lum_planar_vscale on plane 3, hscale.c:63-79,
    vscale.c:57-70; pinned to the reference in test_oracle_vs_ref.py::test_alpha_on_both_sides_is_the_luma_scaler): the oracle's luma of the
    base conversion with A in Y's place, through every kernel family (exact 2x, exact 2:1, the column walkers, the tiled kernel), the batch
    face and the host-pointer face"""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    sf, sw, sh, df, dw, dh, flags, pad = case
    base = {"yuva420p": "yuv420p", "yuva422p": "yuv422p", "yuva444p": "yuv444p"}
    bs, bd = base[sf], base[df]
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=pad)
    ht = S.HostTables(sw, sh, PIX[bs], dw, dh, PIX[bd], flags)
    t = ffi.make_otables(sw, sh, PIX[bs], dw, dh, PIX[bd], flags, ht.banks(), ht.coeffs(), full=ht.full())
    want = ffi.alloc_frame(PIX[bd], dw, dh)
    sp, ss = ffi.planes(src[:3])
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    wa = ffi.alloc_frame(PIX[bd], dw, dh)
    sp, ss = ffi.planes([src[3], src[1], src[2]])
    dp, ds = ffi.planes(wa)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    want.append(wa[0])
    ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    n = 3
    dsrc = _upload(src, n=n)
    ddst = [torch.full((n,) + a.shape[:1] + (a.shape[1] + pad,), 9, dtype=torch.uint8, device="cuda:0") for a in want]
    for _ in range(2):                      # the second call reuses the context's scratch planes
        ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for f in range(n):
        for p, a in enumerate(want):
            got = ddst[p][f].cpu().numpy()
            assert np.array_equal(got[:, :a.shape[1]], a), "frame %d plane %d: %d mismatches" % (f, p, (got[:, :a.shape[1]] != a).sum())
            assert (got[:, a.shape[1]:] == 9).all(), "plane %d: bytes beyond the row were written" % p
    assert not (want[3] == 255).all()
    with pytest.raises(RuntimeError, match="alpha"):
        ctx.scale_batch(dsrc[:3], ddst)
    hd = [np.zeros_like(a) for a in want]
    assert ctx.scale(src, hd) == dh
    for a, b in zip(hd, want):
        assert np.array_equal(a, b)
    ctx.close()


ALPHA_CASES = [("yuv420p", 64, 36, "yuva420p", 128, 72, ffi.SWS_BICUBIC, 0), ("yuv422p", 65, 37, "yuva444p", 40, 30, ffi.SWS_BILINEAR, 5),
               ("nv12", 64, 36, "yuva422p", 96, 54, ffi.SWS_BICUBIC | ffi.SWS_ACCURATE_RND, 0), ("yuva420p", 64, 36, "yuv420p", 128, 72, ffi.SWS_BICUBIC, 3),
               ("yuva444p", 64, 36, "rgb24", 100, 50, ffi.SWS_BICUBIC, 0), ("yuva422p", 66, 38, "nv12", 33, 19, ffi.SWS_BILINEAR, 0),
               ("yuv420p", 64, 36, "yuva420p", 64, 36, ffi.SWS_BICUBIC, 0), ("yuv420p", 960, 540, "yuva420p", 1920, 1080, ffi.SWS_BICUBIC, 0)]


@pytest.mark.parametrize("case", ALPHA_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x_p%d" % c)
def test_alpha_on_one_side(case):
    """yuva420p / 422p / 444p on one side: a source's alpha plane is not read, a target's is filled with 255 (swscale.c:536-553) and the
    other planes are the base formats' (pinned to the reference in test_oracle_vs_ref.py::test_alpha_on_one_side).  (Alpha on both sides:
    planar, test_alpha_on_both_sides; a source alpha plane into packed RGBA, test_scaled_source_alpha_into_packed_rgba.)"""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    sf, sw, sh, df, dw, dh, flags, pad = case
    base = {"yuva420p": "yuv420p", "yuva422p": "yuv422p", "yuva444p": "yuv444p"}
    bs, bd = base.get(sf, sf), base.get(df, df)
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng, pad=pad)
    ht = S.HostTables(sw, sh, PIX[bs], dw, dh, PIX[bd], flags)
    t = ffi.make_otables(sw, sh, PIX[bs], dw, dh, PIX[bd], flags, ht.banks(), ht.coeffs(), full=ht.full())
    want = ffi.alloc_frame(PIX[bd], dw, dh)
    sp, ss = ffi.planes(src[:3] if sf in base else src)
    dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    if df in base:
        want.append(np.full((dh, dw), 255, np.uint8))
    ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    n = 3
    dsrc = _upload(src, n=n)
    ddst = [torch.full((n,) + a.shape[:1] + (a.shape[1] + pad,), 9, dtype=torch.uint8, device="cuda:0") for a in want]
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    for f in range(n):
        for p, a in enumerate(want):
            got = ddst[p][f].cpu().numpy()
            assert np.array_equal(got[:, :a.shape[1]], a), "frame %d plane %d: %d mismatches" % (f, p, (got[:, :a.shape[1]] != a).sum())
            assert (got[:, a.shape[1]:] == 9).all(), "plane %d: bytes beyond the row were written" % p
    if df in base:
        with pytest.raises(RuntimeError, match="alpha"):
            ctx.scale_batch(dsrc, ddst[:3])
    hd = [np.zeros_like(a) for a in want]
    assert ctx.scale(src, hd) == dh
    for a, b in zip(hd, want):
        assert np.array_equal(a, b)
    ctx.close()
