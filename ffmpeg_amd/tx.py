"""Host-side mirror of libavutil's av_tx_init()/av_tx_fn for the float MDCT on the hip path.

  TxContext(type, inv, len, scale)   ~ av_tx_init()              libavutil/tx.c:903
  TxContext.fn(out, in, stride)      ~ av_tx_fn (one transform)  libavutil/tx.h:151   (host numpy)
  TxContext.batch(out, in, ...)      ~ (batched, HBM-resident)   no reference equivalent
"""
import ctypes as C

import numpy as np

from . import _lib

FLOAT_FFT, FLOAT_MDCT, FLOAT_RDFT, FLOAT_DCT = 0, 1, 6, 9
DOUBLE_FFT, DOUBLE_MDCT, INT32_FFT, INT32_MDCT = 2, 3, 4, 5   # libavutil/tx.h:48-69: rows of float64 / int32, *scale a double / a float
FLOAT_DCT_I, FLOAT_DST_I = 12, 15   # libavutil/tx.h:107-128: forward, even lengths 4..1024
FULL_IMDCT, REAL_TO_REAL, REAL_TO_IMAGINARY = 1 << 2, 1 << 3, 1 << 4
BITEXACT = 1 << 32   # FFHIP_TX_BITEXACT: the C reference's operation order (include/ffhip.h)
_TXFN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t)


class TxContext:
    def __init__(self, type_, inv, len_, scale, flags=0):
        L = _lib.lib()
        self.type, self.inv, self.len, self.scale = type_, int(bool(inv)), len_, float(scale)
        self._c = _lib.vp()
        fn = _lib.vp()
        sc = C.c_double(scale) if type_ in (DOUBLE_FFT, DOUBLE_MDCT) else C.c_float(scale)
        self.dtype = np.float64 if type_ in (DOUBLE_FFT, DOUBLE_MDCT) else np.int32 if type_ in (INT32_FFT, INT32_MDCT) else np.float32
        _lib.check(L.ffhip_tx_init(C.byref(self._c), C.byref(fn), type_, self.inv, len_, C.byref(sc), flags), "ffhip_tx_init")
        self._fn = _TXFN(fn.value)

    def close(self):
        if getattr(self, "_c", None) is not None and self._c and _lib is not None:
            _lib.lib().ffhip_tx_uninit(C.byref(self._c))
        self._c = None

    __del__ = close

    def fn(self, out, inp, stride=4):
        """One transform on host arrays of the context's sample type, exactly av_tx_fn's (s, out, in, stride)."""
        assert out.dtype == self.dtype and inp.dtype == self.dtype
        self._fn(self._c, out.ctypes.data, inp.ctypes.data, stride)

    def batch(self, out, inp, stride=None, stream=None):
        """out/inp: 2-D cuda tensors of the context's sample type, one transform per row (row pitch = tensor stride)."""
        import torch
        nt = inp.shape[0]
        es = inp.element_size()
        if stride is None:
            stride = es
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        return _lib.check(_lib.lib().ffhip_tx_batch_dev(self._c, out.data_ptr(), out.stride(0) * es, inp.data_ptr(),
                                                        inp.stride(0) * es, stride, nt, stream), "ffhip_tx_batch_dev")
