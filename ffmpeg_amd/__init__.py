"""ffmpeg_amd — MI355X (gfx950) "hip" arch for FFmpeg's data-parallel DSP inner loops.

The product is libffhip.so (hand-written HIP kernels behind the C-ABI in include/ffhip.h); this
package is the thin host-side mirror used by the tests and bench: ctypes bindings (`_lib`) and
python classes shaped like the reference's contexts (`swscale`, `h264`, `mecmp`, `tx`).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
