"""GPU edge cases of the batch faces: EMPTY batches are no-ops that leave every buffer alone, and arguments the path does
not support come back as errors (never as a silent fallback)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EINVAL = -22


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def test_empty_batches_are_noops():
    from ffmpeg_amd import _lib, h264, hevc, fdsp, me, tx, swscale as S
    torch = _torch()
    dev = "cuda:0"
    pic = torch.full((64, 64), 0x5A, dtype=torch.uint8, device=dev)
    keep = pic.clone()
    rec = torch.zeros((1, 32), dtype=torch.uint8, device=dev)           # any record array: never read when n == 0
    i16 = torch.full((64,), 7, dtype=torch.int16, device=dev)
    i16k = i16.clone()
    L = _lib.lib()
    assert L.ffhip_h264_qpel_batch_dev(pic.data_ptr(), pic.data_ptr(), 64, rec.data_ptr(), 0, None) == 0
    assert L.ffhip_h264_chroma_mc_batch_dev(pic.data_ptr(), pic.data_ptr(), 64, rec.data_ptr(), 0, None) == 0
    assert L.ffhip_h264_weight_batch_dev(pic.data_ptr(), pic.data_ptr(), 64, rec.data_ptr(), 0, None) == 0
    assert L.ffhip_h264_loop_filter_batch_dev(pic.data_ptr(), 64, rec.data_ptr(), 0, None) == 0
    assert L.ffhip_hevc_idct_batch_dev(hevc.IDCT, 3, i16.data_ptr(), pic.data_ptr(), 64, rec.data_ptr(), 0, None) == 0
    f = torch.ones((4, 16), dtype=torch.float32, device=dev)
    fk = f.clone()
    assert L.ffhip_fdsp_batch_dev(fdsp.FMUL, f.data_ptr(), 64, f.data_ptr(), 64, f.data_ptr(), 64, None, 0, 0.0, 16, 0, None) == 0
    assert L.ffhip_fdsp_batch_dev(fdsp.FMUL, f.data_ptr(), 64, f.data_ptr(), 64, f.data_ptr(), 64, None, 0, 0.0, 0, 4, None) == 0
    ctx = tx.TxContext(tx.FLOAT_MDCT, 0, 64, 1.0)
    tin = torch.ones((1, 128), dtype=torch.float32, device=dev)
    tout = torch.full((1, 64), 3.0, dtype=torch.float32, device=dev)
    assert L.ffhip_tx_batch_dev(ctx._c, tout.data_ptr(), 256, tin.data_ptr(), 512, 4, 0, None) == 0
    ctx.close()
    sws = S.SwsContext(64, 32, 23, 128, 64, 23, S.SWS_BICUBIC)
    src = [torch.zeros((0, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 64, 32)]
    dst = [torch.zeros((0, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 128, 64)]
    sws.scale_batch(src, dst)                                            # zero frames
    sws.close()
    torch.cuda.synchronize()
    assert torch.equal(pic, keep) and torch.equal(i16, i16k) and torch.equal(f, fk) and float(tout[0, 0]) == 3.0


def test_unsupported_arguments_are_errors():
    from ffmpeg_amd import _lib, hevc, fdsp
    torch = _torch()
    dev = "cuda:0"
    L = _lib.lib()
    buf = torch.zeros((256,), dtype=torch.int16, device=dev)
    rec = torch.zeros((1, 12), dtype=torch.uint8, device=dev)
    pic = torch.zeros((64, 64), dtype=torch.uint8, device=dev)
    assert L.ffhip_hevc_idct_batch_dev(hevc.IDCT, 6, buf.data_ptr(), None, 0, rec.data_ptr(), 1, None) == EINVAL      # 64x64: no such transform
    assert L.ffhip_hevc_idct_batch_dev(hevc.DST_4X4, 3, buf.data_ptr(), None, 0, rec.data_ptr(), 1, None) == EINVAL   # DST is 4x4 only
    assert L.ffhip_hevc_idct_batch_dev(hevc.ADD_ONLY, 3, buf.data_ptr(), None, 0, rec.data_ptr(), 1, None) == EINVAL  # nothing to add to
    assert L.ffhip_hevc_idct_batch_dev(9, 3, buf.data_ptr(), pic.data_ptr(), 64, rec.data_ptr(), 1, None) == EINVAL
    f = torch.ones((16,), dtype=torch.float32, device=dev)
    assert L.ffhip_fdsp_batch_dev(fdsp.FMUL, f.data_ptr(), 0, f.data_ptr(), 0, None, 0, None, 0, 0.0, 16, 1, None) == EINVAL   # src1 missing
    assert L.ffhip_fdsp_batch_dev(17, f.data_ptr(), 0, f.data_ptr(), 0, f.data_ptr(), 0, None, 0, 0.0, 16, 1, None) == EINVAL
    assert not L.ffhip_sws_getContext(64, 32, 23, 128, 64, 8, 4)      # AV_PIX_FMT_GRAY8 target: not on the hip path
    assert not L.ffhip_sws_getContext(64, 32, 78, 64, 32, 26, 4)      # yuva422p -> rgba at equal size: the reference's special converters have no 4:2:2-with-alpha form here
    assert not L.ffhip_sws_getContext(64, 32, 0, 128, 64, 71, 4)      # gbrp is a target of the equal-size converter only
    assert b"ffhip" in L.ffhip_last_error()


def test_av_tx_types_off_the_path_are_refused():
    """the RDFT / DCT / DCT-I / DST-I forms of AV_TX_DOUBLE_* / AV_TX_INT32_* (libavutil/tx.h:47-132) are not on the hip
    path: ENOSYS with the device present (the caller keeps its C codelets), as include/ffhip.h says — not a silent fallback, not a wrong
    transform.  (The double / int32 FFT and MDCT, types 2 - 5, and the float DCT-I / DST-I, 12 and 15, are on it since round 6:
    tests/test_gpu_tx_wide.py, test_gpu_tx_dcst1.py)"""
    import ctypes as C
    from ffmpeg_amd import _lib
    _torch()
    L = _lib.lib()
    ENOSYS = -38
    scale = C.c_float(1.0)
    for typ in (7, 8, 10, 11, 13, 14, 16, 17):
        ctx, fn = C.c_void_p(), C.c_void_p()
        assert L.ffhip_tx_init(C.byref(ctx), C.byref(fn), typ, 0, 64, C.byref(scale), 0) == ENOSYS, typ
        assert not ctx.value
