"""The `hip` AVHWDeviceType (integration/avutil_hwcontext_hip.c) through the reference's own generic hwcontext entry points.

oracle/_ref/hwcontext_hip_test (built by oracle/refbuild `make hwcontext` from the reference's libavutil / libswscale objects, its
hwcontext.c compiled unchanged with the hip type in hw_table[]) creates the device, a frame pool in HBM, uploads a host frame, hands
AVFrame.data[] / linesize[] of the device frames to ffhip_sws_scale_batch_dev, downloads, and compares with the reference's sws_scale.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "hwcontext_hip_test")

CASES = [
    ("640", "360", "1280", "720", "yuv420p", "yuv420p"),
    ("1920", "1080", "3840", "2160", "yuv420p", "yuv420p"),      # the headline shape, frames resident between upload and download
    ("1920", "1080", "1280", "720", "nv12", "yuv420p"),
    ("1280", "720", "1280", "720", "yuv420p", "rgb24"),
    ("642", "358", "1000", "562", "yuv420p", "bgr24"),
    ("1920", "1080", "3840", "2160", "p010le", "p010le"),
    ("960", "540", "1920", "1080", "yuv420p10le", "yuv420p"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(c))
def test_hwcontext_hip_frames_scaled_in_hbm(case):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/hwcontext_hip_test not built (needs /root/reference at build time)")
    r = subprocess.run([EXE, *case], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS hwcontext hip" in r.stdout


GRAPH_OK = [("rgb24", "bgr24", "640", "360"), ("rgba", "rgb24", "641", "359"), ("yuv444p", "rgb24", "640", "360"), ("rgb24", "yuv444p", "640", "360"),
            ("yuv444p10le", "yuv444p", "640", "360"), ("yuv444p", "yuv444p16le", "640", "360"), ("bgr0", "rgb24", "1920", "1080"),
            ("yuv444p", "yuv444p", "640", "360", "1280", "360"), ("yuv444p", "yuv444p", "640", "360", "640", "720")]
#: two-dimensional scaling: two passes, the intermediate frame in device memory (integration/swscale_graph_hip.c)
GRAPH_2D = [("yuv444p", "yuv444p", "640", "360", "960", "540"), ("yuv444p", "yuv444p", "1920", "1080", "1280", "720"), ("rgb24", "rgb24", "640", "360", "1280", "720"),
            ("yuv444p", "rgb24", "640", "360", "800", "450"), ("yuv444p10le", "yuv444p10le", "640", "360", "1280", "720"), ("rgba", "bgr24", "641", "359", "320", "200")]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GRAPH_OK, ids=lambda c: "-".join(c))
def test_sws_scale_frame_on_hip_frames(case):
    """sws_scale_frame() of the reference's new API on two frames of the hip device: its graph offers the op list to the backend whose
    hw_format matches (integration/swscale_hw_hip.c), op_pass_run hands it AVFrame.data[] = device pointers, the conversion runs in
    HBM; the downloaded result equals the same call on host frames with backend_c.  Format conversions and one-dimensional scaling
    are single passes."""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/hwcontext_hip_test not built")
    r = subprocess.run([EXE, "graph", *case], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS sws_scale_frame on hip frames" in r.stdout and "bit-exact with backend_c" in r.stdout


#: subsampled formats: no op backend is offered those (format.c:560-600); the graph falls back to the legacy scaler, whose passes on hip
#: frames run libffhip's scaler on the device pointers (integration/swscale_graph_hip.c)
GRAPH_LEGACY = [("yuv420p", "yuv420p", "640", "360", "1280", "720"), ("nv12", "nv12", "1920", "1080", "3840", "2160"), ("nv12", "yuv420p", "640", "360", "320", "180"),
                ("yuv420p", "rgb24", "640", "360", "1280", "720"), ("p010le", "p010le", "640", "360", "1280", "720"), ("yuv420p10le", "yuv420p10le", "1280", "720", "640", "360"),
                ("nv12", "yuv420p", "640", "360"), ("yuv422p", "yuv420p", "640", "360", "800", "600"), ("yuv420p", "bgra", "642", "358", "1000", "500")]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GRAPH_LEGACY, ids=lambda c: "-".join(c))
def test_sws_scale_frame_subsampled_formats_on_hip_frames(case):
    """sws_scale_frame() between two hip frames of subsampled formats — among them BASELINE's headline conversion, nv12 1080p -> 4K
    bicubic: the reference's graph builds a legacy-scaler pass (add_legacy_sws_pass, graph.c:560-660) and its ff_swscale() call arrives
    with device pointers at libffhip's scaler; == the same call on host frames (the reference's own ff_swscale), bit for bit"""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/hwcontext_hip_test not built")
    r = subprocess.run([EXE, "graph", *case], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS sws_scale_frame on hip frames" in r.stdout and "bit-exact with backend_c" in r.stdout and " 1 legacy-scaler passes" in r.stdout


@pytest.mark.gpu
def test_sws_scale_frame_refuses_what_libffhip_does_not_take():
    """a conversion of hip frames the legacy device scaler does not take (gray8 here) is refused — logged, the test program exits 3 —
    not run on device pointers by the reference's C code"""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/hwcontext_hip_test not built")
    r = subprocess.run([EXE, "graph", "yuv420p", "yuv411p", "640", "360", "320", "180"], capture_output=True, text=True, timeout=300)
    assert r.returncode in (1, 3), r.stdout + r.stderr      # 1: the frames context refuses the format, 3: the scaler does


@pytest.mark.gpu
@pytest.mark.parametrize("case", GRAPH_2D, ids=lambda c: "-".join(c))
def test_sws_scale_frame_two_passes_on_hip_frames(case):
    """Two-dimensional scaling is two passes with an intermediate frame pass_alloc_output() allocates (graph.c:130-175) — in host memory
    for every device type but Vulkan; integration/swscale_graph_hip.c gives graphs between two hip frames a device-memory intermediate
    (the branch pass_alloc_output_hw() is for Vulkan), so both passes run in HBM: == the same call on host frames with backend_c"""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/hwcontext_hip_test not built")
    r = subprocess.run([EXE, "graph", *case], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS sws_scale_frame on hip frames" in r.stdout and "bit-exact with backend_c" in r.stdout
    assert " 0 intermediate planes" not in r.stdout, r.stdout


def test_hwcontext_hip_without_a_device_reports_it():
    """No device: the program says so and exits 77 (the type is still compiled into hw_table[])."""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/hwcontext_hip_test not built")
    import ffmpeg_amd._lib as L
    if L.lib().ffhip_device_count() > 0:
        pytest.skip("a device is present")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert r.returncode == 77 and "SKIP" in r.stdout
