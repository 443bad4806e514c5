"""CPU tier: the `hip` device type and pixel format have enum values of their own and rows in the reference's name tables.

oracle/_ref/hwcontext_hip_test checks, before it asks for a device, that av_hwdevice_find_type_by_name("hip"), av_get_pix_fmt("hip"),
av_pix_fmt_desc_get(AV_PIX_FMT_HIP) (AV_PIX_FMT_FLAG_HWACCEL) answer through the reference's own hwcontext.c / pixdesc.c compiled
where they lie (integration/avutil_hwcontext_table_hip.c, avutil_pixdesc_hip.c), and that the CUDA rows are still CUDA's."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "hwcontext_hip_test")


def test_hip_rows_in_the_reference_tables():
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/hwcontext_hip_test not built (needs /root/reference at build time)")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode in (0, 77), r.stdout + r.stderr      # 77: no device here
    assert "hip rows of hw_type_names[] / av_pix_fmt_descriptors[]: OK" in r.stdout, r.stdout + r.stderr
