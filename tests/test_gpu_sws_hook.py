"""The frame-level SwsFunc hook of the FFmpeg-side patch (integration/swscale_unscaled_hip.c) as a caller of libswscale's public API
meets it: oracle/_ref/sws_unscaled_hip_test is the reference's libavutil + libswscale (compiled where they lie) with the `hip` arch's
hooks behind their inits, linked with libffhip.so.  It runs sws_getContext() / sws_scale() / sws_setColorspaceDetails() /
sws_freeContext() on host frames with cpu flags 0 and with AV_CPU_FLAG_HIP forced and compares the pictures byte for byte (guard bytes
included): yuv420p / yuv422p / yuva420p to rgb24, bgr24, argb, rgba, abgr, bgra, gbrp; whole frames (BASELINE configs[0]: yuv420p ->
rgb24 1920x1080 first), slices, bottom-up pictures, a colour matrix / range / brightness / contrast / saturation set after the init;
and two pairs the hook must leave with the C converter.  The reference's own harness on the same pointer: test_gpu_checkasm.py
(sw_yuv2rgb)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "sws_unscaled_hip_test")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/sws_unscaled_hip_test not built (needs /root/reference at build time)")
def test_sws_scale_on_host_frames_goes_through_the_hip_swsfunc():
    r = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, cwd="/tmp")
    tail = "\n".join(r.stdout.splitlines()[-40:])
    assert r.returncode == 0, tail
    lines = [l for l in r.stdout.splitlines() if l.startswith(("OK  ", "FAIL"))]
    assert lines[0].startswith("OK  ") and "yuv420p -> rgb24 1920x1080" in lines[0] and "hip SwsFunc == C" in lines[0], lines[0]
    m = re.search(r"(\d+) cases, 0 failed", r.stdout)
    assert m and int(m.group(1)) >= 50, tail
    assert sum("left to C" in l for l in lines) == 2, tail
    assert not [l for l in lines if l.startswith("FAIL")], tail


EXE_SCALED = os.path.join(ROOT, "oracle", "_ref", "sws_scaled_hip_test")


@pytest.mark.skipif(not os.path.exists(EXE_SCALED), reason="oracle/_ref/sws_scaled_hip_test not built (needs /root/reference at build time)")
def test_sws_scale_of_scaled_contexts_goes_through_the_hip_swsfunc():
    """round 5: SCALED contexts (and NV12 -> RGB at the source's size, for which the reference has no special converter) get a frame-level
    SwsFunc at the end of ff_sws_init_scale() (ff_sws_hip_scaled_hook(), integration/swscale_unscaled_hip.c) with the context's own banks:
    the reference's sws_scale() on host frames then runs libffhip's fused kernels — NV12 1080p -> rgb24, 1080p -> 4K (BASELINE configs[1]'s
    conversion), 4K -> 1080p into RGB, ragged widths, thumbnails, 10-bit sources, source slices, bottom-up pictures, colour details set
    after the init — and the pictures are the C scaler's byte for byte"""
    r = subprocess.run([EXE_SCALED], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, cwd="/tmp")
    tail = "\n".join(r.stdout.splitlines()[-45:])
    assert r.returncode == 0, tail
    lines = [l for l in r.stdout.splitlines() if l.startswith(("OK  ", "FAIL"))]
    assert not [l for l in lines if l.startswith("FAIL")], tail
    assert lines[0].startswith("OK  ") and "nv12 1920x1080 -> rgb24 1920x1080" in lines[0] and "hip SwsFunc == ff_swscale" in lines[0], lines[0]
    assert any("nv12 1920x1080 -> nv12 3840x2160" in l and l.startswith("OK  ") for l in lines), tail
    m = re.search(r"(\d+) cases, 0 failed", r.stdout)
    assert m and int(m.group(1)) >= 49, tail
    assert any("yuv444p 1920x1080 -> rgb24 1920x1080" in l and "hip SwsFunc == ff_swscale" in l for l in lines), tail
    # round 6: the frame API asking for TARGET slices of a scaled picture (slice -64 / -1 / -32 in the listing) gives the C
    # scaler's picture; SWS_FAST_BILINEAR contexts (flags 0x1) are left to ff_swscale() like the dithered 16-bit target and the gray source
    assert sum(" slice -" in l and l.startswith("OK  ") for l in lines) == 3, tail
    # (counted over the whole output: the runtime's own stderr lines can land inside a line of the listing)
    assert r.stdout.count("left to C") == 5 and sum("flags 0x1 " in l for l in lines) == 2, tail
    # round 6: packed RGB sources (a capture / an image for an encoder): the input converters run on the device, the context is that of their
    # 14-bit lines (ffhip_sws_from_tables_rgb_source) — seven conversions through the hook, and bgr24 -> yuv420p at the source's size, the
    # reference's own special converter, left alone
    assert sum(l.startswith("OK  ") and re.search(r"^OK   (rgb24|bgr24|rgba|bgra|argb|abgr) ", l) is not None and "hip SwsFunc == ff_swscale" in l
               for l in lines) == 7, tail
    assert any("bgr24 640x360 -> yuv420p 640x360" in l and "left to C" in l for l in lines), tail
    # ... and 10-bit sources into packed RGB (the walker's first stage + k_y16_rgb)
    assert sum(l.startswith("OK  ") and re.search(r"^OK   (p010le|yuv420p10le) .* -> (bgra|rgb24|rgba) ", l) is not None and "hip SwsFunc == ff_swscale" in l
               for l in lines) == 3, tail
