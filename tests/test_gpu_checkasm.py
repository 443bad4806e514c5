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
         "hevc_pel", "hevc_sao", "vp9dsp", "float_dsp", "av_tx", "sw_scale", "sw_ops"]


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
