"""The reference's OWN harness against the `hip` arch: oracle/_ref/checkasm_hip is tests/checkasm of the reference (harness and tests
compiled where they lie) linked with the reference's C objects, the FFmpeg-side hooks of integration/ and libffhip.so.  Each test
initialises its DSP table through the reference's init (which now ends with the hip hook), and the harness compares every member
the hip arch installs with the C function of the same table on its own randomised inputs — exactly what it does for sse2 / neon.
The binary is built by __graft_entry__.build() where /root/reference exists and travels to the GPU box with the tree."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "checkasm_hip")

# the tables libffhip replaces (include/ffhip.h) and the checkasm test that covers each
TESTS = ["h264dsp", "h264qpel", "h264chroma", "h264pred", "motion", "hevc_add_res", "hevc_idct", "hevc_deblock", "hevc_dequant",
         "hevc_pel", "hevc_sao", "vp9dsp", "float_dsp", "av_tx", "sw_scale", "sw_ops", "sw_yuv2rgb"]


def run(test, seed=1):
    r = subprocess.run([EXE, "--test=" + test, str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500,
                       cwd="/tmp")
    return r.returncode, r.stdout


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/checkasm_hip not built (needs /root/reference at build time)")
@pytest.mark.parametrize("test", TESTS)
def test_reference_checkasm_passes_for_the_hip_flag(test):
    rc, out = run(test)
    tail = "\n".join(out.splitlines()[-25:])
    assert "no HIP device" not in out, tail
    assert rc == 0, tail
    # the harness's own verdict: "checkasm: all N tests passed"; N counts the functions checked for the HIP flag
    m = re.search(r"all (\d+) tests passed", out)
    assert m and int(m.group(1)) > 0, tail
    assert "HIP:" in out or "hip" in out.lower(), tail


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/checkasm_hip not built (needs /root/reference at build time)")
def test_sw_yuv2rgb_rows_of_the_frame_level_hook():
    """tests/checkasm/sw_yuv2rgb.c:179-180 checks c->convert_unscaled, the pointer integration/swscale_unscaled_hip.c installs
    (libswscale/swscale_unscaled.c:2698-2704): the harness must report checked functions for the yuv420p, yuv422p and yuva420p
    sources — 7 targets x 4 widths each — not "no tests to perform" (its comparison allows +-3; byte equality is
    tests/test_gpu_sws_hook.py)"""
    rc, out = run("sw_yuv2rgb")
    assert rc == 0, out[-2000:]
    assert "no tests to perform" not in out, out[-2000:]
    m = re.search(r"all (\d+) tests passed", out)
    assert m and int(m.group(1)) >= 3 * 7 * 4, out[-2000:]
    for name in ("yuv420p", "yuv422p", "yuva420p"):
        assert re.search(r"sw_yuv2rgb\.%s\s.*OK" % name, out) or re.search(r"%s\b.*\[OK\]" % name, out), out[-2000:]
