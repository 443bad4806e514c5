"""GPU parity of the scaler above 8 bits (k_sws_scale16 through ffhip_sws_scale_batch_dev and the host-pointer ffhip_sws_scale): ==
the oracle (oracle/ffo_sws_hbd.c, pinned to the reference's sws_scale() on the real pixel formats by
tests/test_oracle_vs_ref_sws_hbd.py), bit for bit, on every format pair / scaler / ratio of that test, several frames per launch,
plus a 1080p -> 4K p010 frame."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import u8p
from test_oracle_vs_ref_sws_hbd import CASES, FMT, RANGE_CASES, make_frame, planes_of, oracle_tables

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _oracle_frame(t, sname, dname, src, dw, dh, pad):
    O = ffi.oracle()
    O.ffo_sws_scale_frame_hbd.argtypes = [C.POINTER(ffi.OSwsTables), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(C.c_int),
                                          C.POINTER(u8p), C.POINTER(C.c_int)]
    want = make_frame(dname, dw, dh, None, pad=pad)
    sp, ss = planes_of(src)
    wp, ws = planes_of(want)
    assert O.ffo_sws_scale_frame_hbd(C.byref(t), FMT[sname][1], FMT[sname][2], FMT[dname][1], FMT[dname][2], sp, ss, wp, ws) == 0
    return want


def _run(case, nframes=3, pad=8, ranges=None):
    from ffmpeg_amd import swscale as S
    torch = _torch()
    sname, sw, sh, dname, dw, dh, flags = case
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    if ranges is None:
        ht, t = oracle_tables(sname, sw, sh, dname, dw, dh, flags)
    else:   # sws_setColorspaceDetails()'s srcRange / dstRange: the tables carry them, the context is built from the tables
        ht = S.HostTables(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags, ranges=ranges)
        t = ffi.make_otables(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags, ht.banks(), ht.coeffs(), ranges=ranges, dst_depth=FMT[dname][1])
        assert (ht.t.lumConvertRange_coeff, ht.t.lumConvertRange_offset, ht.t.chrConvertRange_coeff, ht.t.chrConvertRange_offset) == \
               (t.lum_rc_coeff, t.lum_rc_offset, t.chr_rc_coeff, t.chr_rc_offset)
    frames = [make_frame(sname, sw, sh, rng, pad=pad) for _ in range(nframes)]
    wants = [_oracle_frame(t, sname, dname, f, dw, dh, pad) for f in frames]
    ctx = S.SwsContext(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags, tables=ht.t if ranges is not None else None)
    nsp, ndp = len(frames[0]), len(wants[0])
    src = [torch.from_numpy(np.stack([f[i] for f in frames]).view(np.uint8)).cuda() for i in range(nsp)]
    dst = [torch.zeros((nframes,) + tuple(wants[0][i].view(np.uint8).shape), dtype=torch.uint8, device="cuda") for i in range(ndp)]
    ctx.scale_batch(src, dst)
    torch.cuda.synchronize()
    for i in range(ndp):
        got = dst[i].cpu().numpy()
        for f in range(nframes):
            w = wants[f][i].view(np.uint8)
            vis = (wants[f][i].shape[1] - pad) * wants[f][i].itemsize       # the padding columns are not the scaler's to write
            assert np.array_equal(got[f][:, :vis], w[:, :vis]), "frame %d plane %d: %d bytes differ" % (f, i, (got[f][:, :vis] != w[:, :vis]).sum())
            assert not got[f][:, vis:].any()
    # the host-pointer face (sws_scale() shape) on the first frame
    out = make_frame(dname, dw, dh, None, pad=pad)
    ctx.scale([p.view(np.uint8) for p in frames[0]], [p.view(np.uint8) for p in out])
    for i in range(ndp):
        vis = out[i].shape[1] - pad
        assert np.array_equal(out[i][:, :vis], wants[0][i][:, :vis]), i
    ctx.close()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_scale_above_8_bits(case):
    _run(case)


# exact 2x between formats of 9..14 bits laid out alike: the static-schedule kernel's 16-bit twin (k_sws_up2<., ., 1>); "tiled" runs the
# same cases on k_sws_scale16 (FFHIP_SWS_UP2=0, the measure build)
UP2_CASES = [
    ("yuv420p10le", 64, 36, "yuv420p10le", 128, 72, ffi.SWS_BICUBIC),
    ("yuv420p10le", 72, 40, "yuv420p10le", 144, 80, ffi.SWS_BILINEAR),
    ("p010le", 72, 40, "p010le", 144, 80, ffi.SWS_BICUBIC),              # (u, v) columns: the pair path, samples in the high bits
    ("p012le", 64, 36, "p012le", 128, 72, ffi.SWS_POINT),
    ("yuv420p10le", 1000, 62, "yuv420p10le", 2000, 124, ffi.SWS_BICUBIC),  # 250 groups per row: ragged-end blocks shared by frames
    ("p010le", 520, 70, "p010le", 1040, 140, ffi.SWS_BICUBIC),
    # round 6: 8-bit sources on planes widened to 16-bit samples (hScale8To15_c == hScale16To15_c at depth 8): planar into planar, NV12 into P01x
    ("yuv420p", 72, 40, "yuv420p10le", 144, 80, ffi.SWS_BICUBIC),
    ("nv12", 72, 40, "p010le", 144, 80, ffi.SWS_BICUBIC),
    ("yuv420p", 1000, 62, "yuv420p12le", 2000, 124, ffi.SWS_BILINEAR),
    ("nv12", 520, 70, "p012le", 1040, 140, ffi.SWS_BICUBIC),
    ("yuv422p10le", 64, 36, "yuv422p10le", 128, 72, ffi.SWS_BICUBIC),
    ("yuv444p10le", 64, 36, "yuv444p10le", 128, 72, ffi.SWS_BICUBIC),
    ("yuv420p10le", 64, 36, "yuv420p12le", 128, 72, ffi.SWS_BICUBIC),      # depths differ: >> (src depth - 1) across, >> (27 - dst depth) down
    ("yuv420p14le", 64, 36, "yuv420p9le", 128, 72, ffi.SWS_BICUBIC),
    ("yuv420p12le", 64, 36, "yuv420p12le", 128, 72, ffi.SWS_AREA),
]


@pytest.mark.parametrize("variant", ["product", "tiled"])
@pytest.mark.parametrize("case", UP2_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_exact_2x_above_8_bits(case, variant, monkeypatch):
    from ffmpeg_amd import swscale as S
    if variant == "tiled":
        monkeypatch.setenv("FFHIP_SWS_UP2", "0")
    else:
        _torch()
        ctx = S.SwsContext(case[1], case[2], FMT[case[0]][0], case[4], case[5], FMT[case[3]][0], case[6])
        assert ctx.up2_path, "the exact-2x kernel should serve this context"
        ctx.close()
    _run(case, nframes=5)


@pytest.mark.parametrize("case", RANGE_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x_%d%d" % c)
def test_range_conversion_above_8_bits(case):
    """lumRangeToJpeg_c ... / the ...16_c forms (19-bit intermediates, 64-bit products) inside k_sws_scale16"""
    _run(case[:7], ranges=case[7:])


DOWN2_CASES = [
    ("yuv420p10le", 96, 48, "yuv420p10le", 48, 24, ffi.SWS_BICUBIC),
    ("yuv420p10le", 112, 64, "yuv420p10le", 56, 32, ffi.SWS_BILINEAR),
    ("p010le", 96, 48, "p010le", 48, 24, ffi.SWS_BICUBIC),                 # (u, v) columns: the pair path, samples in the high bits
    ("p012le", 96, 48, "p012le", 48, 24, ffi.SWS_AREA),
    ("yuv420p10le", 2000, 124, "yuv420p10le", 1000, 62, ffi.SWS_BICUBIC),  # 250 groups per row: a ragged last block
    ("p010le", 1040, 140, "p010le", 520, 70, ffi.SWS_BICUBIC),
    ("yuv422p10le", 96, 48, "yuv422p10le", 48, 24, ffi.SWS_BICUBIC),
    ("yuv444p10le", 96, 48, "yuv444p10le", 48, 24, ffi.SWS_BICUBIC),
    ("yuv420p10le", 96, 48, "yuv420p12le", 48, 24, ffi.SWS_BICUBIC),
    ("yuv420p14le", 96, 48, "yuv420p9le", 48, 24, ffi.SWS_BICUBIC),
]


@pytest.mark.parametrize("variant", ["product", "tiled"])
@pytest.mark.parametrize("case", DOWN2_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_exact_2to1_above_8_bits(case, variant, monkeypatch):
    """exact 2:1 between formats of 9..14 bits laid out alike: the static-schedule kernel's 16-bit twin (k_sws_down2<1>); "tiled" runs
    the same cases on k_sws_scale16 (FFHIP_SWS_DOWN2=0, the measure build)"""
    from ffmpeg_amd import swscale as S
    if variant == "tiled":
        monkeypatch.setenv("FFHIP_SWS_DOWN2", "0")
    else:
        _torch()
        ctx = S.SwsContext(case[1], case[2], FMT[case[0]][0], case[4], case[5], FMT[case[3]][0], case[6])
        assert ctx.down2_path, "the exact-2:1 kernel should serve this context"
        ctx.close()
    _run(case, nframes=5)


# a 9..14-bit source into an 8-bit planar / NV12 target — a 10-bit decoder's frames for an 8-bit consumer — on the 16-bit column walker
# (round 5: the 16-bit horizontal pass, yuv2planeX_8_c / yuv2nv12cX_c with the ordered dither on the way out; was k_sws_scale16 at 0.05
# of HBM); "tiled" runs the same cases on k_sws_scale16 (FFHIP_SWS_WALK16=0, the measure build)
TO8_CASES = [
    ("p010le", 384, 216, "nv12", 192, 108, ffi.SWS_BICUBIC),          # 2:1, interleaved in and out
    ("p010le", 384, 216, "yuv420p", 256, 144, ffi.SWS_BICUBIC),       # 1.5:1, interleaved in, planar out: the V plane's dither three entries on
    ("yuv420p10le", 384, 216, "yuv420p", 256, 144, ffi.SWS_BICUBIC),
    ("yuv420p10le", 384, 216, "nv12", 192, 108, ffi.SWS_BICUBIC),     # planar in, interleaved out
    ("yuv420p12le", 202, 120, "yuv420p", 302, 180, ffi.SWS_BICUBIC),  # up, ragged last groups
    ("yuv422p10le", 384, 216, "yuv422p", 256, 144, ffi.SWS_BICUBIC),
    ("yuv444p10le", 192, 108, "yuv444p", 288, 162, ffi.SWS_BILINEAR),
    ("p010le", 1048, 600, "nv12", 700, 400, ffi.SWS_BICUBIC),         # several column blocks and strips
    # exact 2:1 into 8 bits, layouts alike: k_sws_down2<1> with the ordered dither on the way out
    ("yuv420p10le", 384, 216, "yuv420p", 192, 108, ffi.SWS_BICUBIC),  # planar: the V plane's dither three entries on
    ("p010le", 2576, 96, "nv12", 1288, 48, ffi.SWS_BICUBIC),          # several lane blocks, ragged last one
    ("yuv420p12le", 2064, 40, "yuv420p", 1032, 20, ffi.SWS_BILINEAR),
    ("yuv444p10le", 384, 216, "yuv444p", 192, 108, ffi.SWS_BICUBIC),
]


def test_exact_half_into_8_bits_takes_the_static_kernel():
    from ffmpeg_amd import swscale as S
    for sname, dname in (("p010le", "nv12"), ("yuv420p10le", "yuv420p")):
        ctx = S.SwsContext(384, 216, FMT[sname][0], 192, 108, FMT[dname][0], ffi.SWS_BICUBIC)
        assert ctx.paths & 16, ctx.paths
        ctx.close()
    ctx = S.SwsContext(384, 216, FMT["yuv420p10le"][0], 192, 108, FMT["nv12"][0], ffi.SWS_BICUBIC)   # planar in, pairs out: the walker
    assert ctx.walk16_path and not ctx.paths & 16
    ctx.close()


@pytest.mark.parametrize("variant", ["product", "tiled"])
@pytest.mark.parametrize("case", TO8_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_deeper_source_into_8_bits(case, variant, monkeypatch):
    from ffmpeg_amd import swscale as S
    if variant == "tiled":
        monkeypatch.setenv("FFHIP_SWS_WALK16", "0")
    else:
        sname, sw, sh, dname, dw, dh, flags = case
        ctx = S.SwsContext(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags)
        # (exact 2:1 between formats laid out alike runs the static-schedule kernel's 16-bit twin with the dithered 8-bit stage: bit 4)
        assert ctx.walk16_path or ctx.paths & 16, "case reaches neither the 16-bit column walker nor the exact-2:1 kernel"
        ctx.close()
    _run(case)


# banks of 9..16 taps on the 16-bit column walker (round 5): ratios between 1/2 and 1/4 — a 4K HDR frame into 720p
# exact 3:2 up between formats of 9..14 bits laid out alike (720p -> 1080p): the static-schedule kernel of sws_up32.hip (fast_path bit 13);
# "walker" runs the same cases on k_sws_walk16 (FFHIP_SWS_UP32=0, the measure build)
UP32_CASES = [
    ("yuv420p10le", 24, 36, "yuv420p10le", 36, 54, ffi.SWS_BICUBIC),        # three groups per chroma row, the least the kernel takes
    ("yuv420p10le", 96, 40, "yuv420p10le", 144, 60, ffi.SWS_BILINEAR),
    ("p010le", 96, 40, "p010le", 144, 60, ffi.SWS_BICUBIC),                  # (u, v) columns: the pair path, samples in the high bits
    ("p012le", 48, 36, "p012le", 72, 54, ffi.SWS_POINT),
    ("yuv420p10le", 520, 124, "yuv420p10le", 780, 186, ffi.SWS_BICUBIC),    # 130 groups per luma row: two full blocks of lanes and a ragged one
    ("p010le", 520, 52, "p010le", 780, 78, ffi.SWS_BICUBIC),
    ("yuv422p10le", 96, 36, "yuv422p10le", 144, 54, ffi.SWS_BICUBIC),
    ("yuv444p10le", 48, 36, "yuv444p10le", 72, 54, ffi.SWS_BICUBIC),
    ("yuv420p10le", 96, 36, "yuv420p12le", 144, 54, ffi.SWS_BICUBIC),        # depths differ
    ("yuv420p14le", 96, 36, "yuv420p9le", 144, 54, ffi.SWS_BICUBIC),
    ("yuv420p12le", 96, 36, "yuv420p12le", 144, 54, ffi.SWS_AREA),
    ("yuv420p10le", 96, 300, "yuv420p10le", 144, 450, ffi.SWS_BICUBIC),      # several strips of rows
    # 8-bit sources on planes widened to words: planar into planar, NV12 into P01x
    ("yuv420p", 96, 40, "yuv420p10le", 144, 60, ffi.SWS_BICUBIC),
    ("nv12", 96, 40, "p010le", 144, 60, ffi.SWS_BICUBIC),
    # exact 4:3 (1080p -> 1440p): period (3 in, 4 out) of the same kernel
    ("yuv420p10le", 36, 54, "yuv420p10le", 48, 72, ffi.SWS_BICUBIC),        # three groups per chroma row
    ("yuv420p10le", 144, 60, "yuv420p10le", 192, 80, ffi.SWS_BILINEAR),
    ("p010le", 144, 60, "p010le", 192, 80, ffi.SWS_BICUBIC),
    ("p012le", 72, 54, "p012le", 96, 72, ffi.SWS_POINT),
    ("yuv420p10le", 792, 114, "yuv420p10le", 1056, 152, ffi.SWS_BICUBIC),    # 132 groups per luma row, 66 per chroma row (ragged)
    ("p010le", 792, 66, "p010le", 1056, 88, ffi.SWS_BICUBIC),
    ("yuv444p10le", 72, 54, "yuv444p10le", 96, 72, ffi.SWS_BICUBIC),
    ("yuv420p10le", 144, 54, "yuv420p12le", 192, 72, ffi.SWS_BICUBIC),
    ("yuv420p10le", 144, 330, "yuv420p10le", 192, 440, ffi.SWS_BICUBIC),     # several strips of rows
    ("nv12", 144, 60, "p010le", 192, 80, ffi.SWS_BICUBIC),
]


@pytest.mark.parametrize("variant", ["product", "walker"])
@pytest.mark.parametrize("case", UP32_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_exact_3to2_up_above_8_bits(case, variant, monkeypatch):
    from ffmpeg_amd import swscale as S
    if variant == "walker":
        monkeypatch.setenv("FFHIP_SWS_UP32", "0")
    else:
        _torch()
        ctx = S.SwsContext(case[1], case[2], FMT[case[0]][0], case[4], case[5], FMT[case[3]][0], case[6])
        assert ctx.paths & 8192, "the exact-3:2 up-scaler should serve this context"
        ctx.close()
    _run(case, nframes=5)


# exact 3:2 down between formats of 9..14 bits laid out alike (1080p -> 720p, 4K -> 1440p): the 16-bit twin of sws_down32.hip (fast_path bit 12)
DOWN32H_CASES = [
    ("yuv420p10le", 72, 54, "yuv420p10le", 48, 36, ffi.SWS_BICUBIC),
    ("yuv420p10le", 144, 60, "yuv420p10le", 96, 40, ffi.SWS_BILINEAR),
    ("p010le", 144, 60, "p010le", 96, 40, ffi.SWS_BICUBIC),                  # (u, v) columns: the pair path, samples in the high bits
    ("p012le", 72, 54, "p012le", 48, 36, ffi.SWS_POINT),
    ("yuv420p10le", 1560, 186, "yuv420p10le", 1040, 124, ffi.SWS_BICUBIC),   # 260 groups per luma row, 130 per chroma row (a ragged block)
    ("p010le", 780, 78, "p010le", 520, 52, ffi.SWS_BICUBIC),
    ("yuv422p10le", 144, 54, "yuv422p10le", 96, 36, ffi.SWS_BICUBIC),
    ("yuv444p10le", 72, 54, "yuv444p10le", 48, 36, ffi.SWS_BICUBIC),
    ("yuv420p10le", 144, 54, "yuv420p12le", 96, 36, ffi.SWS_BICUBIC),        # depths differ
    ("yuv420p14le", 144, 54, "yuv420p9le", 96, 36, ffi.SWS_BICUBIC),
    ("yuv420p12le", 144, 54, "yuv420p12le", 96, 36, ffi.SWS_AREA),
    ("yuv420p10le", 144, 450, "yuv420p10le", 96, 300, ffi.SWS_BICUBIC),      # several strips of rows
    # ... into an 8-bit target laid out alike, with the ordered dither (a 10-bit decoder's 1080p frames for a 720p 8-bit consumer)
    ("p010le", 144, 60, "nv12", 96, 40, ffi.SWS_BICUBIC),
    ("yuv420p10le", 144, 54, "yuv420p", 96, 36, ffi.SWS_BICUBIC),
    ("yuv420p12le", 1560, 186, "yuv420p", 1040, 124, ffi.SWS_BILINEAR),
    ("p010le", 780, 78, "nv12", 520, 52, ffi.SWS_BICUBIC),
    ("yuv444p10le", 72, 54, "yuv444p", 48, 36, ffi.SWS_BICUBIC),
    # exact 4:3 down (1440p -> 1080p): period (4 in, 3 out) of the same kernel
    ("yuv420p10le", 48, 72, "yuv420p10le", 36, 54, ffi.SWS_BICUBIC),        # three groups per chroma row
    ("yuv420p10le", 192, 80, "yuv420p10le", 144, 60, ffi.SWS_BILINEAR),
    ("p010le", 192, 80, "p010le", 144, 60, ffi.SWS_BICUBIC),
    ("p012le", 96, 72, "p012le", 72, 54, ffi.SWS_POINT),
    ("yuv420p10le", 1056, 152, "yuv420p10le", 792, 114, ffi.SWS_BICUBIC),    # 132 / 66 groups: ragged blocks
    ("p010le", 1056, 88, "p010le", 792, 66, ffi.SWS_BICUBIC),
    ("yuv444p10le", 96, 72, "yuv444p10le", 72, 54, ffi.SWS_BICUBIC),
    ("yuv420p10le", 192, 72, "yuv420p12le", 144, 54, ffi.SWS_BICUBIC),
    ("yuv420p10le", 192, 440, "yuv420p10le", 144, 330, ffi.SWS_BICUBIC),     # several strips of rows
]


@pytest.mark.parametrize("variant", ["product", "walker"])
@pytest.mark.parametrize("case", DOWN32H_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_exact_3to2_down_above_8_bits(case, variant, monkeypatch):
    from ffmpeg_amd import swscale as S
    if variant == "walker":
        monkeypatch.setenv("FFHIP_SWS_DOWN32", "0")
    else:
        _torch()
        ctx = S.SwsContext(case[1], case[2], FMT[case[0]][0], case[4], case[5], FMT[case[3]][0], case[6])
        assert ctx.paths & 4096, "the exact-3:2 down-scaler should serve this context"
        ctx.close()
    _run(case, nframes=5)


def test_3to2_up_shapes_the_static_kernel_leaves_to_the_walker():
    """widths that are not whole groups, odd heights, a range change: the context is built, on the walker"""
    from ffmpeg_amd import swscale as S
    _torch()
    for case in (("yuv420p10le", 44, 36, "yuv420p10le", 66, 54, ffi.SWS_BICUBIC), ("p010le", 90, 40, "p010le", 135, 60, ffi.SWS_BICUBIC),
                 ("yuv420p10le", 96, 42, "yuv420p10le", 144, 63, ffi.SWS_BICUBIC)):
        ctx = S.SwsContext(case[1], case[2], FMT[case[0]][0], case[4], case[5], FMT[case[3]][0], case[6])
        assert not ctx.paths & 8192
        ctx.close()
        _run(case, nframes=2)


WIDEN8_CASES = [
    ("yuv420p", 384, 216, "yuv420p10le", 576, 324, ffi.SWS_BICUBIC),   # 1.5x up: the walker on widened planes
    ("nv12", 384, 216, "p010le", 256, 144, ffi.SWS_BICUBIC),           # interleaved in and out
    ("nv12", 384, 216, "yuv420p10le", 288, 162, ffi.SWS_BILINEAR),     # interleaved in, planar out
    ("yuv422p", 202, 120, "yuv422p10le", 302, 180, ffi.SWS_BICUBIC),   # ragged last groups
    ("yuv444p", 200, 120, "yuv420p12le", 150, 90, ffi.SWS_BICUBIC),
    ("yuv420p", 1280, 720, "p010le", 1920, 1080, ffi.SWS_BICUBIC),
]


@pytest.mark.parametrize("variant", ["product", "tiled"])
@pytest.mark.parametrize("case", WIDEN8_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_8_bit_source_into_deeper_target(case, variant, monkeypatch):
    """round 6: 8-bit planar / NV12 sources into 9..14-bit targets on the 16-bit walker (planes widened to words by k_sws_widen8), against the
    oracle — and the tiled kernel (FFHIP_SWS_WALK16=0) that served them before"""
    if variant == "tiled":
        monkeypatch.setenv("FFHIP_SWS_WALK16", "0")
    _run(case, nframes=3 if case[1] < 1000 else 1)


WIDE16_CASES = [
    ("p010le", 576, 324, "p010le", 192, 108, ffi.SWS_BICUBIC),         # 3:1: 12 x 12 taps
    ("yuv420p10le", 640, 360, "yuv420p10le", 160, 90, ffi.SWS_BICUBIC),  # 4:1: 16 x 16 taps
    ("p010le", 576, 324, "nv12", 192, 108, ffi.SWS_BICUBIC),           # ... into 8 bits
    ("yuv420p12le", 640, 180, "yuv420p12le", 160, 90, ffi.SWS_BICUBIC),  # 16 taps across, 8 down
    ("yuv422p10le", 320, 360, "yuv422p10le", 160, 90, ffi.SWS_BICUBIC),  # 8 across, 16 down
    ("p010le", 1152, 648, "yuv420p10le", 400, 226, ffi.SWS_BICUBIC),   # ragged, interleaved in, planar out
]


@pytest.mark.parametrize("case", WIDE16_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_wide_banks_above_8_bits(case):
    from ffmpeg_amd import swscale as S
    sname, sw, sh, dname, dw, dh, flags = case
    ctx = S.SwsContext(sw, sh, FMT[sname][0], dw, dh, FMT[dname][0], flags)
    assert ctx.walk16_path, "case does not reach the 16-bit column walker"
    ctx.close()
    _run(case)


def test_p010_4k_to_nv12_1080p():
    _run(("p010le", 3840, 2160, "nv12", 1920, 1080, ffi.SWS_BICUBIC), nframes=2)


def test_p010_1080p_to_4k():
    _run(("p010le", 1920, 1080, "p010le", 3840, 2160, ffi.SWS_BICUBIC), nframes=1, pad=0)


def test_the_bench_lines_of_the_3to2_and_4to3_kernels_at_full_size():
    """p010 720p -> 1080p (k_sws_up32), yuv420p10 1080p -> 1440p (its 4:3 period), p010 4K -> 1440p (k_sws_down32h): the sizes bench.py times"""
    _run(("p010le", 1280, 720, "p010le", 1920, 1080, ffi.SWS_BICUBIC), nframes=2, pad=0)
    _run(("yuv420p10le", 1920, 1080, "yuv420p10le", 2560, 1440, ffi.SWS_BICUBIC), nframes=1, pad=0)
    _run(("p010le", 3840, 2160, "p010le", 2560, 1440, ffi.SWS_BICUBIC), nframes=1, pad=0)


def test_yuv420p10_1080p_to_4k_and_back():
    _run(("yuv420p10le", 1920, 1080, "yuv420p10le", 3840, 2160, ffi.SWS_BICUBIC), nframes=1, pad=0)
    _run(("yuv420p10le", 3840, 2160, "yuv420p", 1920, 1080, ffi.SWS_BICUBIC), nframes=1, pad=0)


def test_what_is_not_on_the_path_is_refused():
    from ffmpeg_amd import swscale as S
    _torch()
    # equal-size p010 conversion (the reference's special converters); 10-bit into planar RGB; into an odd-width RGB picture and from a
    # 4:4:4 source (the full-chroma writers); one luma tap with a blending chroma pair (yuv2rgb_1's averaged lines)
    for args in ((64, 36, 62, 64, 36, 158), (64, 36, 62, 128, 72, 71), (64, 36, 62, 127, 72, 2), (64, 36, 68, 128, 72, 2),
                 (64, 36, 62, 64, 36, 2, ffi.SWS_BILINEAR)):
        with pytest.raises(ValueError):
            S.SwsContext(*args)


from test_oracle_vs_ref_sws_hbd import RGB_CASES, RGBT, oracle_hbd_rgb  # noqa: E402

HIP_RGB_CASES = [c for c in RGB_CASES if not (c[0] == "yuv420p10le" and c[3] == "rgb24" and c[6] == ffi.SWS_BILINEAR and c[1] == c[4])] + \
                [("p010le", 1920, 1080, "bgra", 1920, 1080, ffi.SWS_BICUBIC), ("yuv420p10le", 3840, 2160, "rgb24", 1920, 1080, ffi.SWS_BICUBIC),
                 ("yuv420p10le", 1280, 720, "rgba", 1920, 1080, ffi.SWS_BILINEAR)]


@pytest.mark.parametrize("case", HIP_RGB_CASES, ids=lambda c: "%s_%dx%d_%s_%dx%d_%x" % c)
def test_deeper_source_into_packed_rgb(case):
    """round 6: a 9..14-bit source into packed 8-bit RGB in two stages — the 16-bit walker into the context's intermediate (an int16 luma
    plane of unclipped sums, 8-bit chroma of half the width, flat rounding seed or none as the reference's writer has it), then k_y16_rgb —
    against the oracle (pinned to the reference's sws_scale() on the CPU tier): the host face and the batched device face"""
    from ffmpeg_amd import swscale as S
    torch = _torch()
    sname, sw, sh, dname, dw, dh, flags = case
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
    dfmt, bpp = RGBT[dname]
    n = 3 if sw < 1000 else 1
    frames = [make_frame(sname, sw, sh, rng, pad=0) for _ in range(n)]
    ctx = S.SwsContext(sw, sh, FMT[sname][0], dw, dh, dfmt, flags)
    want0 = oracle_hbd_rgb(sname, frames[0], sw, sh, dname, dw, dh, flags)[:, :dw * bpp]
    got = np.zeros((dh, dw * bpp + 3), np.uint8)
    assert ctx.scale([a.view(np.uint8) for a in frames[0]], [got]) == dh
    assert np.array_equal(got[:, :dw * bpp], want0), "%d bytes differ" % (got[:, :dw * bpp] != want0).sum()
    src = S.alloc_batch(FMT[sname][0], sw, sh, n, "cuda:0")
    dst = S.alloc_batch(dfmt, dw, dh, n, "cuda:0", fill=9)
    for f in range(n):
        for p, a in enumerate(frames[f]):
            b = a.view(np.uint8)
            src[p][f, :, :b.shape[1]] = torch.from_numpy(b).cuda()
    for rep in range(2):
        ctx.scale_batch(src, dst)
    torch.cuda.synchronize()
    for f in range(n):
        want = want0 if f == 0 else oracle_hbd_rgb(sname, frames[f], sw, sh, dname, dw, dh, flags)[:, :dw * bpp]
        assert np.array_equal(dst[0][f].cpu().numpy()[:, :dw * bpp], want), f
    ctx.close()
