"""Host-side mirror of AACDecDSP.imdct_and_windowing (float AAC decoder, 1024-sample frames) on the hip path.

  AacImdct(windows, scale_1024, scale_128)        ~ what ff_aac_decode_init() sets up      libavcodec/aac/aacdec.c:1267-1285
  AacImdct.frame(coeffs, seq, kb, saved, out)     ~ dsp.imdct_and_windowing(ac, sce)       aacdec_dsp_template.c:325-387 (host numpy)
  AacImdct.batch(coeffs, out, saved, seq, kb, ..) ~ frames x channels at once, HBM-resident (no reference equivalent)
"""
import ctypes as C

import numpy as np

from . import _lib

ONLY_LONG_SEQUENCE, LONG_START_SEQUENCE, EIGHT_SHORT_SEQUENCE, LONG_STOP_SEQUENCE = range(4)   # libavcodec/aac.h:63-68
#: MDCT_INIT's scale_float for the 1024- and 128-point inverse transforms (aacdec.c:1267-1285)
SCALE_1024, SCALE_128 = 2.0 ** -25, 2.0 ** -22


class AacImdct:
    def __init__(self, windows, scale_1024=SCALE_1024, scale_128=SCALE_128):
        """windows: (sine_1024, sine_128, kbd_long_1024, kbd_short_128) float32 arrays - the decoder's own tables"""
        w = [np.ascontiguousarray(x, np.float32) for x in windows]
        assert [x.size for x in w] == [1024, 128, 1024, 128]
        self._c = _lib.vp()
        _lib.check(_lib.lib().ffhip_aac_imdct_create(C.byref(self._c), w[0].ctypes.data, w[1].ctypes.data, w[2].ctypes.data, w[3].ctypes.data,
                                                     scale_1024, scale_128), "ffhip_aac_imdct_create")

    def close(self):
        if getattr(self, "_c", None) is not None and self._c and _lib is not None:
            _lib.lib().ffhip_aac_imdct_free(C.byref(self._c))
        self._c = None

    __del__ = close

    def frame(self, coeffs, window_sequence, use_kb_window, saved, out):
        """one channel, one frame on host float32 arrays; window_sequence / use_kb_window = (this frame, previous frame)"""
        seq = (C.c_int * 2)(*window_sequence)
        kb = (C.c_int * 2)(*use_kb_window)
        assert coeffs.dtype == saved.dtype == out.dtype == np.float32 and coeffs.size >= 1024 and saved.size >= 512 and out.size >= 1024
        return _lib.check(_lib.lib().ffhip_aac_imdct_and_windowing(self._c, coeffs.ctypes.data, seq, kb, saved.ctypes.data, out.ctypes.data),
                          "ffhip_aac_imdct_and_windowing")

    def batch(self, coeffs, out, saved, window_sequence, use_kb_window, prev_sequence, prev_kb_window, stream=None):
        """coeffs / out: float32 cuda tensors [nframes, nch, 1024]; saved: [nch, 512] in and out; the window arrays are host uint8
        ([nframes, nch] resp. [nch])"""
        import torch
        nframes, nch = coeffs.shape[0], coeffs.shape[1]
        assert coeffs.is_contiguous() and out.is_contiguous() and saved.is_contiguous()
        ws, kb = (np.ascontiguousarray(x, np.uint8).reshape(nframes * nch) for x in (window_sequence, use_kb_window))
        ps, pk = (np.ascontiguousarray(x, np.uint8).reshape(nch) for x in (prev_sequence, prev_kb_window))
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        return _lib.check(_lib.lib().ffhip_aac_imdct_and_windowing_batch_dev(self._c, coeffs.data_ptr(), out.data_ptr(), saved.data_ptr(),
                                                                             ws.ctypes.data, kb.ctypes.data, ps.ctypes.data, pk.ctypes.data,
                                                                             nch, nframes, stream), "ffhip_aac_imdct_and_windowing_batch_dev")


#: FFHipAacTnsFilter (include/ffhip.h)
TNS_FILTER_DTYPE = np.dtype([("frame", np.int32), ("start", np.int16), ("size", np.int16), ("inc", np.int8), ("order", np.uint8),
                             ("pad", np.uint8, 2), ("coef", np.float32, 20)])


def tns_filters(frame, n_filt, length, direction, order, coef, num_windows, num_swb, swb_offset, tns_max_bands, max_sfb):
    """apply_tns's walk over windows and filters for one channel-frame: numpy array of FFHipAacTnsFilter records"""
    rec = np.zeros(32, TNS_FILTER_DTYPE)
    args = [np.ascontiguousarray(x, np.int32) for x in (n_filt, length, direction, order)]
    cf = np.ascontiguousarray(coef, np.float32)
    swb = np.ascontiguousarray(swb_offset, np.uint16)
    n = _lib.check(_lib.lib().ffhip_aac_tns_filters(rec.ctypes.data, frame, args[0].ctypes.data, args[1].ctypes.data, args[2].ctypes.data,
                                                    args[3].ctypes.data, cf.ctypes.data, num_windows, num_swb, swb.ctypes.data, tns_max_bands,
                                                    max_sfb), "ffhip_aac_tns_filters")
    return rec[:n]


def apply_tns_batch(coeffs, filters, nfilters, decode=1, stream=None):
    """coeffs: float32 cuda tensor [nframes, 1024], filtered in place; filters: uint8 cuda tensor [nfilters, 92]"""
    return _lib.check(_lib.lib().ffhip_aac_apply_tns_batch_dev(coeffs.data_ptr(), filters.data_ptr(), nfilters, decode,
                                                               None if stream is None else C.c_void_p(stream)), "ffhip_aac_apply_tns_batch_dev")
