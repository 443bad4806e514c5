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
#: the forward LTP transform's scale_float (aacdec.c:1288-1291)
SCALE_LTP = -32786.0 * 2 + 36


class AacImdct:
    def __init__(self, windows, scale_1024=None, scale_128=None, frame_len=1024):
        """windows: (sine_<L>, sine_<L/8>, kbd_long_<L>, kbd_short_<L/8>) float32 arrays - the decoder's own tables; frame_len L =
        1024, 960 (imdct_and_windowing_960) or 768 (_768); the scales default to MDCT_INIT's (1 / len) / 32768"""
        w = [np.ascontiguousarray(x, np.float32) for x in windows]
        assert [x.size for x in w] == [frame_len, frame_len // 8, frame_len, frame_len // 8]
        scale_1024 = (1.0 / frame_len) / 32768.0 if scale_1024 is None else scale_1024
        scale_128 = (8.0 / frame_len) / 32768.0 if scale_128 is None else scale_128
        self.frame_len = frame_len
        self._c = _lib.vp()
        _lib.check(_lib.lib().ffhip_aac_imdct_create_len(C.byref(self._c), frame_len, w[0].ctypes.data, w[1].ctypes.data, w[2].ctypes.data,
                                                         w[3].ctypes.data, scale_1024, scale_128), "ffhip_aac_imdct_create_len")

    def close(self):
        if getattr(self, "_c", None) is not None and self._c and _lib is not None:
            _lib.lib().ffhip_aac_imdct_free(C.byref(self._c))
        self._c = None

    __del__ = close

    def frame(self, coeffs, window_sequence, use_kb_window, saved, out):
        """one channel, one frame on host float32 arrays; window_sequence / use_kb_window = (this frame, previous frame)"""
        seq = (C.c_int * 2)(*window_sequence)
        kb = (C.c_int * 2)(*use_kb_window)
        assert coeffs.dtype == saved.dtype == out.dtype == np.float32 and coeffs.size >= 1024 and saved.size >= self.frame_len // 2 and out.size >= self.frame_len
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


    # -- long-term prediction (AACDecDSP.apply_ltp / update_ltp) on this context's windows and work space
    def ltp_init(self, scale_ltp=SCALE_LTP):
        return _lib.check(_lib.lib().ffhip_aac_ltp_init(self._c, scale_ltp), "ffhip_aac_ltp_init")

    def ltp_predict(self, ltp_state, pred_freq, recs, n, stream=None):
        """ltp_state: float32 cuda [nch, 3072]; pred_freq: float32 cuda [n, 1024] out; recs: uint8 cuda [n, 16] FFHipAacLtp"""
        return _lib.check(_lib.lib().ffhip_aac_ltp_predict_batch_dev(self._c, ltp_state.data_ptr(), pred_freq.data_ptr(), recs.data_ptr(), n,
                                                                     None if stream is None else C.c_void_p(stream)), "ffhip_aac_ltp_predict_batch_dev")

    def update_ltp(self, ltp_state, out, nch, stream=None):
        """after batch(): ltp_state float32 cuda [nch, 3072] in and out, out = the last frame's samples [nch, 1024]"""
        return _lib.check(_lib.lib().ffhip_aac_update_ltp_batch_dev(self._c, ltp_state.data_ptr(), out.data_ptr(), nch,
                                                                    None if stream is None else C.c_void_p(stream)), "ffhip_aac_update_ltp_batch_dev")


class AacLd:
    """AACDecDSP.imdct_and_windowing_ld (eld=False: windows = (ff_sine_512, ff_sine_128)) / _eld (eld=True: windows =
    (ff_aac_eld_window_512 or _480,), frame_len 512 / 480) on device-resident frames"""

    def __init__(self, windows, eld=False, frame_len=512, scale=None):
        w = [np.ascontiguousarray(x, np.float32) for x in windows]
        scale = (1.0 / frame_len) / 32768.0 if scale is None else scale
        self.frame_len, self.eld = frame_len, eld
        self._c = _lib.vp()
        _lib.check(_lib.lib().ffhip_aac_ld_create(C.byref(self._c), int(eld), frame_len, w[0].ctypes.data, None if eld else w[1].ctypes.data, scale),
                   "ffhip_aac_ld_create")

    def close(self):
        if getattr(self, "_c", None) is not None and self._c and _lib is not None:
            _lib.lib().ffhip_aac_ld_free(C.byref(self._c))
        self._c = None

    __del__ = close

    def batch(self, coeffs, out, saved, kb_prev=None, stream=None):
        """coeffs / out: float32 cuda [nframes, nch, 1024]; saved: [nch, 256] (LD) / [nch, 3 * frame_len] (ELD); kb_prev (LD): host
        uint8 [nframes, nch]"""
        import torch
        nframes, nch = coeffs.shape[0], coeffs.shape[1]
        kp = None
        if not self.eld:
            kb = np.ascontiguousarray(kb_prev, np.uint8).reshape(nframes * nch)
            kp = kb.ctypes.data
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        return _lib.check(_lib.lib().ffhip_aac_ld_batch_dev(self._c, coeffs.data_ptr(), out.data_ptr(), saved.data_ptr(), kp, nch, nframes, stream),
                          "ffhip_aac_ld_batch_dev")


#: FFHipAacBandOp / FFHipAacLtp (include/ffhip.h)
BAND_OP_DTYPE = np.dtype([("frame0", np.int32), ("frame1", np.int32), ("start", np.int16), ("len", np.int16), ("scale", np.float32),
                          ("kind", np.uint8), ("pad", np.uint8, 3)])
LTP_DTYPE = np.dtype([("state", np.int32), ("lag", np.int16), ("seq0", np.uint8), ("kb", np.uint8), ("coef", np.float32), ("pad", np.int32)])
BAND_MS, BAND_INTENSITY, BAND_ADD, BAND_FMAC = 0, 1, 2, 3
PREDICTION_DTYPE = np.dtype([("channel", np.int32), ("frame", np.int32), ("kmax", np.int16), ("flags", np.uint8), ("reset_group", np.uint8),
                             ("enable", np.uint32, 21), ("pad", np.uint32)])


def _p(a, dt):
    a = np.ascontiguousarray(a, dt)
    return a, a.ctypes.data


def ms_bands(frame0, frame1, num_window_groups, group_len, max_sfb_ste, ms_mask, band_type0, band_type1, swb_offset):
    """apply_mid_side_stereo's walk over one channel pair: numpy array of FFHipAacBandOp records"""
    rec = np.zeros(64, BAND_OP_DTYPE)
    keep = [_p(group_len, np.uint8), _p(ms_mask, np.uint8), _p(band_type0, np.int32), _p(band_type1, np.int32), _p(swb_offset, np.uint16)]
    n = _lib.check(_lib.lib().ffhip_aac_ms_bands(rec.ctypes.data, frame0, frame1, num_window_groups, keep[0][1], max_sfb_ste, keep[1][1],
                                                 keep[2][1], keep[3][1], keep[4][1]), "ffhip_aac_ms_bands")
    return rec[:n]


def is_bands(frame0, frame1, num_window_groups, group_len, max_sfb, ms_present, ms_mask, band_type1, sf1, swb_offset):
    """apply_intensity_stereo's walk over one channel pair"""
    rec = np.zeros(128, BAND_OP_DTYPE)
    keep = [_p(group_len, np.uint8), _p(ms_mask, np.uint8), _p(band_type1, np.int32), _p(sf1, np.float32), _p(swb_offset, np.uint16)]
    n = _lib.check(_lib.lib().ffhip_aac_is_bands(rec.ctypes.data, frame0, frame1, num_window_groups, keep[0][1], max_sfb, ms_present, keep[1][1],
                                                 keep[2][1], keep[3][1], keep[4][1]), "ffhip_aac_is_bands")
    return rec[:n]


def ltp_bands(frame, pred_frame, max_sfb, used, swb_offset):
    """the band-wise add that ends apply_ltp"""
    rec = np.zeros(20, BAND_OP_DTYPE)
    keep = [_p(used, np.int8), _p(swb_offset, np.uint16)]
    n = _lib.check(_lib.lib().ffhip_aac_ltp_bands(rec.ctypes.data, frame, pred_frame, max_sfb, keep[0][1], keep[1][1]), "ffhip_aac_ltp_bands")
    return rec[:n]


def coupling_bands(dest_frame, src_frame, num_window_groups, group_len, max_sfb, band_type, gain, swb_offset):
    """apply_dependent_coupling's walk (FMAC records)"""
    rec = np.zeros(128, BAND_OP_DTYPE)
    keep = [_p(group_len, np.uint8), _p(band_type, np.int32), _p(gain, np.float32), _p(swb_offset, np.uint16)]
    n = _lib.check(_lib.lib().ffhip_aac_coupling_bands(rec.ctypes.data, dest_frame, src_frame, num_window_groups, keep[0][1], max_sfb, keep[1][1],
                                                       keep[2][1], keep[3][1]), "ffhip_aac_coupling_bands")
    return rec[:n]


def prediction_record(channel, frame, is_long, initialized, predictor_present, prediction_used, pred_sfb_max, swb_offset, reset_group):
    rec = np.zeros(1, PREDICTION_DTYPE)
    keep = [_p(prediction_used, np.uint8), _p(swb_offset, np.uint16)]
    _lib.check(_lib.lib().ffhip_aac_prediction_record(rec.ctypes.data, channel, frame, int(is_long), int(initialized), int(predictor_present),
                                                      keep[0][1], pred_sfb_max, keep[1][1], reset_group), "ffhip_aac_prediction_record")
    return rec


def apply_prediction_batch(predictor_state, coeffs, recs, n, stream=None):
    """predictor_state: float32 cuda [channels, 672, 8]; coeffs: float32 cuda [frames, 1024]; recs: uint8 cuda [n, 100]"""
    return _lib.check(_lib.lib().ffhip_aac_apply_prediction_batch_dev(predictor_state.data_ptr(), coeffs.data_ptr(), recs.data_ptr(), n,
                                                                      None if stream is None else C.c_void_p(stream)),
                      "ffhip_aac_apply_prediction_batch_dev")


def band_ops_batch(a, b, ops, n, stream=None):
    """a / b: float32 cuda tensors of channel-frames [., 1024]; ops: uint8 cuda tensor [n, 20]"""
    return _lib.check(_lib.lib().ffhip_aac_band_ops_batch_dev(a.data_ptr(), b.data_ptr(), ops.data_ptr(), n,
                                                              None if stream is None else C.c_void_p(stream)), "ffhip_aac_band_ops_batch_dev")


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
