"""CPU tier of the decoder-driven H.264 tests: the claims the GPU side builds on, checked against the reference compiled in place.

* FFHIP_MC_EMU's definition (include/ffhip.h) — "footprint sample (x, y) is read at row clamp(y), column clamp(x)" — IS
  emulated_edge_mc() (libavcodec/videodsp_template.c:24-100), for windows on the rim and windows entirely outside the picture;
* the persistent reference decoder of oracle/refbuild/ffref_shim_h264mb.c runs ff_h264_hl_decode_mb() on inter macroblocks whose
  motion vectors leave unpadded reference pictures, and ff_h264_filter_mb(), without touching memory outside the planes (the planes
  sit between guard bands that must stay intact)."""
import ctypes as C

import numpy as np
import pytest

import ffi
import h264_inter_gen as I

pytestmark = pytest.mark.skipif(not ffi.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("depth", [8, 10])
def test_emulated_edge_mc_is_coordinate_clamping(depth):
    R = ffi.ref()
    R.ffref_emulated_edge_mc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t] + [C.c_int] * 6
    R.ffref_emulated_edge_mc.restype = None
    rng = np.random.default_rng(depth)
    dt, px = (np.uint8, 1) if depth == 8 else (np.uint16, 2)
    w, h, stride = 48, 32, 64
    pic = rng.integers(0, 1 << depth, (h, stride), dtype=dt)
    for bw, bh in ((21, 21), (9, 9), (9, 17)):
        buf = np.zeros((bh, stride), dt)
        for _ in range(400):
            far = rng.random() < .4
            x = int(rng.integers(-3000, 3000)) if far else int(rng.integers(-30, w + 10))
            y = int(rng.integers(-3000, 3000)) if far else int(rng.integers(-30, h + 10))
            R.ffref_emulated_edge_mc(depth, buf.ctypes.data, pic.ctypes.data + (y * stride + x) * px, stride * px, stride * px, bw, bh, x, y, w, h)
            ys, xs = np.clip(np.arange(y, y + bh), 0, h - 1), np.clip(np.arange(x, x + bw), 0, w - 1)
            assert np.array_equal(buf[:, :bw], pic[np.ix_(ys, xs)]), (x, y, bw, bh)


@pytest.mark.parametrize("depth,weights", [(8, 0), (8, 1), (8, 2), (10, 1)])
def test_reference_decoder_inter_macroblocks_stay_inside_their_planes(depth, weights):
    R = ffi.ref()
    rng = np.random.default_rng(depth + weights)
    mb_w, mb_h, nref, G = 6, 4, 2, 8                       # G guard rows around every allocation
    dt, px = (np.uint8, 1) if depth == 8 else (np.uint16, 2)
    W, H = mb_w * 16, mb_h * 16
    sy, sc = W + 16, W // 2 + 16
    rows = [H, H // 2, H // 2]
    st = [sy, sc, sc]
    refs = [rng.integers(0, 1 << depth, (nref * rows[pl] + 2 * G, st[pl]), dtype=dt) for pl in range(3)]
    ref0 = [r.copy() for r in refs]
    cur = [np.full((rows[pl] + 2 * G, st[pl]), 7, dt) for pl in range(3)]
    d = I.Dec(R, "ffref_", depth, mb_w, mb_h, sy * px, sc * px, 0)
    for l in (0, 1):
        for i in range(nref):
            d.set_ref(l, i, [refs[pl].ctypes.data + (G + i * rows[pl]) * st[pl] * px for pl in range(3)])
    d.set_pwt(I.make_pwt(rng, weights, depth, nref))
    d.set_cur([cur[pl].ctypes.data + G * st[pl] * px for pl in range(3)])
    for my in range(mb_h):
        for mx in range(mb_w):
            d.decode_inter(I.make_inter_mb(rng, d.bits, mx, my, nref, 2000, depth=depth))
    for pl in range(3):
        assert np.array_equal(refs[pl], ref0[pl])                              # references untouched
        assert (cur[pl][:G] == 7).all() and (cur[pl][-G:] == 7).all()            # nothing written outside the picture's rows
        assert (cur[pl][G:-G, :(W if pl == 0 else W // 2)] != 7).mean() > .9     # every macroblock reconstructed
        assert (cur[pl][G:-G, (W if pl == 0 else W // 2):] == 7).all()           # row padding untouched
    for stt in I.make_filter_picture(rng, d.bits, mb_w, mb_h, depth, .2):
        d.filter_mb(stt["mb_x"], stt["mb_y"], stt)
    for pl in range(3):
        assert (cur[pl][:G] == 7).all() and (cur[pl][-G:] == 7).all()
    d.close()
