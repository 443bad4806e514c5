"""-m gpu: packed 8-bit RGB sources through libffhip's C ABI (ffhip_sws_getContext on rgb24 / bgr24 / rgba / bgra / argb / abgr into a
YUV target: kernels/sws_rgbin.hip in front of the 14-bit planar context) against the oracle's composite — pinned to the reference's
sws_scale() on the CPU tier (tests/test_oracle_vs_ref_sws_rgbin.py) — and against the reference's own outputs in
tests/golden/sws_rgbin.npz: the host face whole and in slices, the batched device face."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import ffi
from ffi import PIX
from test_oracle_vs_ref_sws_rgbin import CASES, DST, alloc_dst, make_rgb, oracle_rgb_scale

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sws_rgbin.npz")
BIG = [("bgra", 1920, 1080, "nv12", 1920, 1080, ffi.SWS_BICUBIC), ("rgb24", 1280, 720, "yuv420p", 1920, 1080, ffi.SWS_BICUBIC),
       ("rgba", 1920, 1080, "yuv420p", 1280, 720, ffi.SWS_BILINEAR), ("bgr24", 1918, 1078, "yuv444p", 1918, 1078, ffi.SWS_BICUBIC)]


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _crop(planes, dname, dw, dh):
    _, layout, hs, vs = DST[dname]
    cw, ch = -((-dw) >> hs), -((-dh) >> vs)
    w = [dw] + ([2 * cw] if layout == 2 else [cw, cw])
    return [p[:, :wi] for p, wi in zip(planes, w)]


@pytest.mark.parametrize("case", CASES + BIG, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_host_face_and_slices(case):
    from ffmpeg_amd import swscale as S
    _torch()
    sname, sw, sh, dname, dw, dh, flags = case
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    rgb = make_rgb(sname, sw, sh, rng)
    want, _ = oracle_rgb_scale(sname, rgb, sw, sh, dname, dw, dh, flags)
    ctx = S.SwsContext(sw, sh, PIX[sname], dw, dh, DST[dname][0], flags)
    got = alloc_dst(dname, dw, dh)
    assert ctx.scale([rgb], got) == dh
    for i, (a, b) in enumerate(zip(_crop(want, dname, dw, dh), _crop(got, dname, dw, dh))):
        assert np.array_equal(a, b), "plane %d: %d samples differ" % (i, (a != b).sum())
    # three source slices in order: the same frame
    got2 = alloc_dst(dname, dw, dh)
    cuts = [0, (sh // 3) & ~1, (2 * sh // 3) & ~1, sh]
    for a, b in zip(cuts[:-1], cuts[1:]):
        r = ctx.scale([rgb[a:]], got2, srcSliceY=a, srcSliceH=b - a)
        assert r == (dh if b == sh else 0)
    for a, b in zip(got, got2):
        assert np.array_equal(a, b)
    ctx.close()


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[4], CASES[9], BIG[0]], ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_batched_device_face(case):
    from ffmpeg_amd import swscale as S
    torch = _torch()
    sname, sw, sh, dname, dw, dh, flags = case
    n = 5 if sw < 1000 else 2
    rng = np.random.default_rng(abs(hash(case)) & 0xFFF)
    frames = [make_rgb(sname, sw, sh, rng, pad=0) for _ in range(n)]
    ctx = S.SwsContext(sw, sh, PIX[sname], dw, dh, DST[dname][0], flags)
    src = S.alloc_batch(PIX[sname], sw, sh, n, "cuda:0")
    dst = S.alloc_batch(DST[dname][0], dw, dh, n, "cuda:0", fill=7)
    for f in range(n):
        src[0][f, :, :frames[f].shape[1]] = torch.from_numpy(frames[f]).cuda()
    for rep in range(2):          # the second call reuses the context's converter planes
        ctx.scale_batch(src, dst)
    torch.cuda.synchronize()
    for f in range(n):
        want, _ = oracle_rgb_scale(sname, frames[f], sw, sh, dname, dw, dh, flags)
        for i, (a, d) in enumerate(zip(_crop(want, dname, dw, dh), dst)):
            assert np.array_equal(a, d[f].cpu().numpy()[:, :a.shape[1]]), (f, i)
    ctx.close()


# RGB -> yuv420p / NV12 at the source's size: one kernel (k_sws_rgb420: luma direct, the chroma's vertical bank on a register ring); "two-stage"
# runs the same cases on the converter pass + the walker (FFHIP_SWS_RGB420=0, the measure build)
FUSED = [CASES[0], CASES[6], CASES[8], CASES[13], ("argb", 260, 130, "nv12", 260, 130, ffi.SWS_BICUBIC), ("abgr", 1032, 70, "yuv420p", 1032, 70, ffi.SWS_BICUBIC),
         ("rgba", 64, 12, "nv12", 64, 12, ffi.SWS_BILINEAR), ("rgb24", 1280, 720, "yuv420p", 1280, 720, ffi.SWS_BICUBIC), ("bgr24", 64, 2, "nv12", 64, 2, ffi.SWS_BICUBIC),
         ("bgra", 1920, 1080, "yuv420p", 1920, 1080, ffi.SWS_POINT),
         # every bank the identity: planar 4:2:2 / 4:4:4 targets (one elementwise pass; "two-stage": FFHIP_SWS_RGB_DIRECT_OFF in the measure build)
         CASES[3], ("bgr24", 66, 37, "yuv444p", 66, 37, ffi.SWS_BICUBIC), ("argb", 64, 36, "yuv422p", 64, 36, ffi.SWS_BICUBIC),
         ("rgb24", 1278, 719, "yuv444p", 1278, 719, ffi.SWS_BILINEAR), ("abgr", 130, 35, "yuv422p", 130, 35, ffi.SWS_BICUBIC)]


@pytest.mark.parametrize("variant", ["product", "two-stage"])
@pytest.mark.parametrize("case", FUSED, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_rgb_into_420_at_the_source_size(case, variant, monkeypatch):
    from ffmpeg_amd import swscale as S
    torch = _torch()
    if variant == "two-stage":
        monkeypatch.setenv("FFHIP_SWS_RGB420", "0")
        monkeypatch.setenv("FFHIP_SWS_RGB_DIRECT_OFF", "1")
    sname, sw, sh, dname, dw, dh, flags = case
    n = 4 if sw < 1000 else 2
    rng = np.random.default_rng(abs(hash(case)) & 0xFFF)
    frames = [make_rgb(sname, sw, sh, rng, pad=0) for _ in range(n)]
    ctx = S.SwsContext(sw, sh, PIX[sname], dw, dh, DST[dname][0], flags)
    src = S.alloc_batch(PIX[sname], sw, sh, n, "cuda:0")
    dst = S.alloc_batch(DST[dname][0], dw, dh, n, "cuda:0", fill=7)
    for f in range(n):
        src[0][f, :, :frames[f].shape[1]] = torch.from_numpy(frames[f]).cuda()
    ctx.scale_batch(src, dst)
    torch.cuda.synchronize()
    for f in range(n):
        want, _ = oracle_rgb_scale(sname, frames[f], sw, sh, dname, dw, dh, flags)
        for i, (a, d) in enumerate(zip(_crop(want, dname, dw, dh), dst)):
            got = d[f].cpu().numpy()
            assert np.array_equal(a, got[:, :a.shape[1]]), (f, i)
            assert (got[:, a.shape[1]:] == 7).all(), (f, i)      # nothing written past the rows' ends
    ctx.close()


def test_golden_vectors_on_the_gpu():
    from ffmpeg_amd import swscale as S
    _torch()
    d = np.load(GOLD)
    for i in range(int(d["ncases"][0])):
        sf, sw, sh, df, dw, dh, fl = (int(v) for v in d["c%d_meta" % i])
        ctx = S.SwsContext(sw, sh, sf, dw, dh, df, fl)
        want = [d["c%d_dst%d" % (i, p)] for p in range(3) if "c%d_dst%d" % (i, p) in d.files]
        got = [np.zeros_like(a) for a in want]
        assert ctx.scale([np.ascontiguousarray(d["c%d_src" % i])], got) == dh
        for p, (a, b) in enumerate(zip(want, got)):
            assert np.array_equal(a, b), (i, p)
        ctx.close()


def test_refusals():
    from ffmpeg_amd import swscale as S
    _torch()
    # bgr24 -> yuv420p at equal size is the reference's special converter (ff_rgb24toyv12); alpha on both sides; full-range targets; RGB -> RGB
    for sf, df in (("bgr24", 0), ("rgba", 33), ("rgb24", 12), ("rgb24", PIX["bgr24"])):
        with pytest.raises(Exception):
            S.SwsContext(64, 36, PIX[sf], 64, 36, df, ffi.SWS_BICUBIC)
    # ... while the same bgr24 source into a scaled or non-4:2:0 target is on the path
    S.SwsContext(64, 36, PIX["bgr24"], 128, 72, 0, ffi.SWS_BICUBIC).close()
    S.SwsContext(64, 36, PIX["bgr24"], 64, 36, PIX["nv12"], ffi.SWS_BICUBIC).close()
