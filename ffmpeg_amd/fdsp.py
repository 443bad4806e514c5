"""ctypes mirror of the AVFloatDSPContext faces of libffhip (include/ffhip.h): the vector operations around the MDCT
(libavutil/float_dsp.h:31-175)."""
import ctypes as C

from . import _lib

FMUL, FMAC_SCALAR, FMUL_SCALAR, FMUL_WINDOW, FMUL_ADD, FMUL_REVERSE, BUTTERFLIES = range(7)


def _p(t):
    return None if t is None else t.data_ptr()


def _pitch(t):
    """byte pitch between the vectors of a [nvec, len] tensor; a 1-D tensor is shared by the whole batch"""
    return 0 if t is None or t.dim() == 1 else t.stride(0) * 4


def batch(op, dst, src0, src1=None, src2=None, mul=0.0, length=None, stream=None):
    """dst/src*: float32 device tensors [nvec, n] (or [n]: shared); length = the reference's len argument"""
    nvec = dst.shape[0] if dst.dim() == 2 else 1
    if length is None:
        length = src0.shape[-1]
    return _lib.check(_lib.lib().ffhip_fdsp_batch_dev(op, _p(dst), _pitch(dst), _p(src0), _pitch(src0), _p(src1), _pitch(src1), _p(src2),
                                                      _pitch(src2), mul, length, nvec, None if stream is None else C.c_void_p(stream)),
                      "ffhip_fdsp_batch_dev")


class FloatDSPContext(C.Structure):
    """FFHipFloatDSPContext: host-pointer faces with the reference's signatures"""
    _fields_ = [("vector_fmul", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)),
                ("vector_fmac_scalar", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_float, C.c_int)),
                ("vector_fmul_scalar", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_float, C.c_int)),
                ("vector_fmul_window", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)),
                ("vector_fmul_add", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)),
                ("vector_fmul_reverse", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)),
                ("butterflies_float", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int))]


def dsp_init():
    c = FloatDSPContext()
    _lib.check(_lib.lib().ff_float_dsp_init_hip(C.byref(c)), "ff_float_dsp_init_hip")
    return c
