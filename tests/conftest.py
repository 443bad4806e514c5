import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the in-tree shared objects exist (cheap no-op when they are current)."""
    import __graft_entry__ as g
    g.build(quiet=True)


@pytest.fixture(autouse=True)
def _knob_tests_use_the_measure_build(monkeypatch):
    """The product library reads no environment variable.  A test that sets an FFHIP_* knob (a measured kernel variant, a
    fault-injection hook) therefore runs against ffmpeg_amd/libffhip_measure.so — the same sources with -DFFHIP_MEASURE — from the
    moment it sets the first knob until it ends; every other test runs the product build."""
    from ffmpeg_amd import _lib
    orig = monkeypatch.setenv

    def setenv(name, value, prepend=None):
        if name.startswith("FFHIP_"):
            _lib.select("measure")
        return orig(name, value, prepend)
    monkeypatch.setenv = setenv
    yield
    _lib.select("product")


@pytest.fixture
def measure_build():
    """For tests whose knob must already be live when they bind the library (fault-injection into installed faces)."""
    from ffmpeg_amd import _lib
    _lib.select("measure")
    yield
    _lib.select("product")
