"""ffhip_frames_alloc / ffhip_frames_free (include/ffhip.h; round 6): frame memory built from physical chunks mapped into one virtual range in
shuffled order.  The range is ordinary device memory — kernels, copies and torch views work on it — and the scaler gives the oracle's bytes on it."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import PIX

pytestmark = pytest.mark.gpu


def test_frames_alloc_is_device_memory():
    import torch
    from ffmpeg_amd import _lib
    L = _lib.lib()
    for chunk, nbytes in ((0, 5 << 20), (64 << 20, (64 << 20) + 1), (2 << 20, 3 << 20)):
        mem = _lib.FrameMemory(nbytes, chunk)
        assert mem.ptr % (2 << 20) == 0         # (the runtime does not honour a larger alignment of the reservation; the physical chunks are what counts)
        t = mem.tensor((nbytes,))
        t.copy_(torch.arange(nbytes, device="cuda:0", dtype=torch.int64).to(torch.uint8))
        torch.cuda.synchronize()
        back = t.cpu().numpy()
        assert np.array_equal(back, np.arange(nbytes, dtype=np.int64).astype(np.uint8))
        del t
        mem.close()
    p = C.c_void_p()
    assert L.ffhip_frames_alloc(C.byref(p), 1 << 20, 3 << 20) == -22          # not a power of two
    assert L.ffhip_frames_alloc(None, 1 << 20, 0) == -22
    assert L.ffhip_frames_free(C.c_void_p(0x1000)) == -22                      # not one of ours
    assert L.ffhip_frames_free(None) == 0


def test_scaler_on_frame_memory_matches_the_oracle():
    import torch
    from ffmpeg_amd import _lib, swscale as S
    import test_gpu_sws as T
    w, h, n = 640, 360, 3
    rng = np.random.default_rng(9)
    src = ffi.alloc_frame(PIX["yuv420p"], w, h, rng)
    want = T._oracle_unscaled(src, w, h, ffi.RGB_LAYOUT[PIX["rgb24"]])
    shapes = [(n,) + p.shape for p in src]
    sizes = [int(np.prod(s)) for s in shapes]
    mem_s, mem_d = _lib.FrameMemory(sum(sizes) + (6 << 20), 64 << 20), _lib.FrameMemory(n * h * 3 * w, 64 << 20)
    at, dsrc = 0, []
    for p, sh, sz in zip(src, shapes, sizes):
        t = mem_s.tensor(sh, at)
        t.copy_(torch.from_numpy(np.ascontiguousarray(p)).cuda().unsqueeze(0).expand(n, -1, -1))
        dsrc.append(t)
        at += (sz + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    ddst = [mem_d.tensor((n, h, 3 * w))]
    ctx = S.SwsContext(w, h, PIX["yuv420p"], w, h, PIX["rgb24"], S.SWS_BICUBIC)
    ctx.scale_batch(dsrc, ddst)
    torch.cuda.synchronize()
    got = ddst[0].cpu().numpy()
    for f in range(n):
        assert np.array_equal(got[f], want)
    ctx.close()
