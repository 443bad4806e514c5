"""ctypes bindings used by the test-suite (and tools/, bench.py's cpu_baseline leg).

  oracle()  -> oracle/liboracle.so      our CPU restatement            (test infrastructure)
  ref()     -> oracle/_ref/libffref.so  the real reference, if built   (test infrastructure)
  The product library is bound in ffmpeg_amd/_lib.py, never here.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libffref.so")

PIX = {"gbrp": 71, "yuv420p": 0, "yuv422p": 4, "yuv444p": 5, "yuva420p": 33, "yuva422p": 78, "yuva444p": 79, "yuvj420p": 12, "yuvj422p": 13, "yuvj444p": 14, "rgb24": 2, "bgr24": 3, "nv12": 23, "nv21": 24, "argb": 25, "rgba": 26, "abgr": 27, "bgra": 28}
RGB_LAYOUT = {2: 0, 3: 1, 25: 2, 26: 3, 27: 4, 28: 5, 71: 6}   # AVPixelFormat -> the oracle's packed layout number
SWS_BICUBIC, SWS_BILINEAR, SWS_POINT, SWS_AREA, SWS_BICUBLIN = 4, 2, 0x10, 0x20, 0x40
SWS_ACCURATE_RND, SWS_BITEXACT = 0x40000, 0x80000

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
i16p = C.POINTER(C.c_int16)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)


def ptr(a, t=u8p):
    return a.ctypes.data_as(t)


class OSwsFilter(C.Structure):
    _fields_ = [("filter", i16p), ("pos", i32p), ("size", C.c_int), ("n", C.c_int)]


class OYuv2RgbCoeffs(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("cy", "oy", "crv", "cbu", "cgu", "cgv")] + [("yoffs", C.c_int)]


class OSwsTables(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("srcW", "srcH", "srcFormat", "dstW", "dstH", "dstFormat", "flags")] + \
               [(k, OSwsFilter) for k in ("hLum", "hChr", "vLum", "vChr")] + [("k", OYuv2RgbCoeffs)] + \
               [("src_range", C.c_int), ("dst_range", C.c_int), ("lum_rc_coeff", C.c_uint32), ("chr_rc_coeff", C.c_uint32),
                ("lum_rc_offset", C.c_int64), ("chr_rc_offset", C.c_int64), ("full_chr", C.c_int), ("full_coef", C.c_int * 6)]


class OLuts(C.Structure):
    _fields_ = [("ramp", C.c_uint8 * 2048), ("rV", C.c_int * 1280), ("gU", C.c_int * 1280),
                ("gV", C.c_int * 1280), ("bU", C.c_int * 1280)]


class OEdge(C.Structure):
    _fields_ = [("offset", C.c_int32), ("kind", C.c_uint8), ("alpha", C.c_uint8), ("beta", C.c_uint8),
                ("pad", C.c_uint8), ("tc0", C.c_int8 * 4)]


EDGE_DTYPE = np.dtype([("offset", "<i4"), ("kind", "u1"), ("alpha", "u1"), ("beta", "u1"), ("pad", "u1"),
                       ("tc0", "i1", (4,))])
QPEL_DTYPE = np.dtype([("dst_offset", "<i4"), ("src_offset", "<i4"), ("mcxy", "u1"), ("size_idx", "u1"),
                       ("avg", "u1"), ("flags", "u1"), ("src_x", "<i2"), ("src_y", "<i2")])     # FFHipQpelBlock, 16 bytes
CHROMA_DTYPE = np.dtype([("dst_offset", "<i4"), ("src_offset", "<i4"), ("w_idx", "u1"), ("h", "u1"), ("x", "u1"), ("y", "u1"), ("avg", "u1"),
                         ("flags", "u1"), ("src_x", "<i2"), ("src_y", "<i2"), ("pad", "<i2")])  # FFHipChromaBlock, 20 bytes

_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("oracle/liboracle.so missing - run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(ORACLE_SO)
        L.ffo_yuv2rgb_luts_init.argtypes = [C.POINTER(OLuts), C.POINTER(OYuv2RgbCoeffs)]
        L.ffo_yuv420p_to_rgb24.argtypes = [C.POINTER(OLuts), C.c_int, C.POINTER(u8p), C.POINTER(C.c_int), C.c_int,
                                           C.c_int, u8p, C.c_int, C.c_int]
        L.ffo_yuv2rgb_unscaled.argtypes = [C.POINTER(OLuts), C.c_int, C.POINTER(u8p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                           C.POINTER(u8p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
        L.ffo_hscale8to15.argtypes = [i16p, C.c_int, u8p, i16p, i32p, C.c_int]
        L.ffo_yuv2planeX8.argtypes = [i16p, C.c_int, C.POINTER(i16p), u8p, C.c_int, u8p, C.c_int]
        L.ffo_yuv2plane1_8.argtypes = [i16p, u8p, C.c_int, u8p, C.c_int]
        L.ffo_yuv2nv12cX.argtypes = [C.c_int, u8p, i16p, C.c_int, C.POINTER(i16p), C.POINTER(i16p), u8p, C.c_int]
        pp = C.POINTER(i16p)
        L.ffo_yuv2rgb_X.argtypes = [C.POINTER(OLuts), i16p, pp, C.c_int, i16p, pp, pp, C.c_int, u8p, C.c_int, C.c_int]
        L.ffo_yuv2rgb_2.argtypes = [C.POINTER(OLuts), pp, pp, pp, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_yuv2rgb_1.argtypes = [C.POINTER(OLuts), i16p, pp, pp, u8p, C.c_int, C.c_int, C.c_int]
        for f in (L.ffo_yuv2rgb_X, L.ffo_yuv2rgb_2, L.ffo_yuv2rgb_1):
            f.restype = None
        L.ffo_sws_range_constants.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]
        L.ffo_sws_range_constants.restype = None
        L.ffo_sws_rgba_alpha.argtypes = [C.POINTER(OSwsTables), u8p, C.c_int, u8p, C.c_int]
        L.ffo_sws_scale_frame.argtypes = [C.POINTER(OSwsTables), C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(u8p),
                                          C.POINTER(C.c_int)]
        for n in ("ffo_h264_idct_add", "ffo_h264_idct8_add", "ffo_h264_idct_dc_add", "ffo_h264_idct8_dc_add"):
            getattr(L, n).argtypes = [u8p, i16p, C.c_ssize_t]
            getattr(L, n).restype = None
        for n in ("ffo_h264_idct_add16", "ffo_h264_idct8_add4", "ffo_h264_idct_add16intra"):
            getattr(L, n).argtypes = [u8p, i32p, i16p, C.c_ssize_t, u8p]
            getattr(L, n).restype = None
        L.ffo_h264_idct_add8.argtypes = [C.POINTER(u8p), i32p, i16p, C.c_ssize_t, u8p]
        L.ffo_h264_idct_add8.restype = None
        L.ffo_h264_luma_dc_dequant_idct.argtypes = [i16p, i16p, C.c_int]
        L.ffo_h264_luma_dc_dequant_idct.restype = None
        L.ffo_h264_chroma_dc_dequant_idct.argtypes = [i16p, C.c_int]
        L.ffo_h264_chroma_dc_dequant_idct.restype = None
        L.ffo_h264_add_pixels_clear.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
        L.ffo_h264_add_pixels_clear.restype = None
        L.ffo_h264_loop_filter.argtypes = [C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, i8p]
        L.ffo_h264_loop_filter.restype = None
        L.ffo_h264_qpel.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t]
        L.ffo_h264_qpel.restype = None
        L.ffo_h264_chroma_mc.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.ffo_h264_chroma_mc.restype = None
        L.ffo_h264_weight.argtypes = [C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_h264_weight.restype = None
        L.ffo_fft_run.argtypes = [C.c_int, C.c_int, f32p, f32p]
        L.ffo_fft_run.restype = None
        L.ffo_rdft_run.argtypes = [C.c_int, C.c_int, C.c_float, f32p, f32p]
        L.ffo_rdft_run.restype = None
        L.ffo_txw_fft_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ffo_txw_fft_run.restype = None
        L.ffo_txw_mdct_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.ffo_txw_mdct_run.restype = None
        L.ffo_dct_run.argtypes = [C.c_int, C.c_int, C.c_float, f32p, f32p]
        L.ffo_dct_run.restype = None
        L.ffo_dcst1_run.argtypes = [C.c_int, C.c_int, C.c_float, f32p, f32p, C.c_ssize_t]
        L.ffo_dcst1_run.restype = None
        L.ffo_rdft_half_run.argtypes = [C.c_int, C.c_int, C.c_float, f32p, f32p]
        L.ffo_rdft_half_run.restype = None
        L.ffo_aac_sine_window.argtypes = [f32p, C.c_int]
        L.ffo_aac_sine_window.restype = None
        L.ffo_aac_kbd_window.argtypes = [f32p, C.c_float, C.c_int]
        L.ffo_aac_kbd_window.restype = None
        L.ffo_aac_imdct_and_windowing.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(f32p), f32p, i32p, i32p, f32p, f32p]
        L.ffo_aac_imdct_and_windowing.restype = None
        L.ffo_aac_imdct_and_windowing_len.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(f32p), f32p, i32p, i32p, f32p, f32p]
        L.ffo_aac_imdct_and_windowing_len.restype = None
        L.ffo_aac_apply_prediction.argtypes = [f32p, f32p, C.c_int, i32p, C.c_int, u8p, C.c_int, C.POINTER(C.c_uint16), C.c_int]
        L.ffo_aac_apply_prediction.restype = None
        L.ffo_aac_apply_dependent_coupling.argtypes = [f32p, f32p, C.c_int, u8p, C.c_int, i32p, f32p, C.POINTER(C.c_uint16)]
        L.ffo_aac_apply_dependent_coupling.restype = None
        L.ffo_aac_apply_independent_coupling.argtypes = [f32p, f32p, C.c_float, C.c_int]
        L.ffo_aac_apply_independent_coupling.restype = None
        L.ffo_aac_imdct_and_windowing_ld.argtypes = [C.c_void_p, f32p, f32p, f32p, C.c_int, f32p, f32p]
        L.ffo_aac_imdct_and_windowing_ld.restype = None
        L.ffo_aac_imdct_and_windowing_eld.argtypes = [C.c_int, C.c_void_p, f32p, f32p, f32p, f32p]
        L.ffo_aac_imdct_and_windowing_eld.restype = None
        L.ffo_aac_tns_filters.argtypes = [C.c_void_p, i32p, i32p, i32p, i32p, f32p, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.c_int, C.c_int]
        L.ffo_aac_tns_filters.restype = C.c_int
        L.ffo_aac_tns_run.argtypes = [f32p, C.c_void_p, C.c_int]
        L.ffo_aac_tns_run.restype = None
        u16p_ = C.POINTER(C.c_uint16)
        L.ffo_aac_apply_mid_side_stereo.argtypes = [f32p, f32p, C.c_int, u8p, C.c_int, u8p, i32p, i32p, u16p_]
        L.ffo_aac_apply_mid_side_stereo.restype = None
        L.ffo_aac_apply_intensity_stereo.argtypes = [f32p, f32p, C.c_int, u8p, C.c_int, C.c_int, u8p, i32p, f32p, u16p_]
        L.ffo_aac_apply_intensity_stereo.restype = None
        L.ffo_aac_apply_ltp.argtypes = [C.c_void_p, C.POINTER(f32p), f32p, f32p, C.c_int, C.c_float, C.POINTER(C.c_int8), i32p, i32p, C.c_int, u16p_,
                                        C.c_void_p, C.c_int, f32p]
        L.ffo_aac_apply_ltp.restype = None
        L.ffo_aac_update_ltp.argtypes = [C.POINTER(f32p), f32p, f32p, f32p, f32p, C.c_int, C.c_int]
        L.ffo_aac_update_ltp.restype = None
        L.ffo_fdsp.argtypes = [C.c_int, f32p, f32p, f32p, f32p, C.c_float, C.c_int]
        L.ffo_fdsp.restype = None
        L.ffo_h264_hl_decode_intra_mb.argtypes = [u8p, u8p, u8p, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, C.c_int, u8p, C.c_uint, C.c_uint,
                                                  u8p, C.c_int, i16p, i16p, i32p, u8p]
        L.ffo_h264_hl_decode_intra_mb.restype = None
        L.ffo_hevc_coef.argtypes = [C.c_int, C.c_int]
        L.ffo_hevc_coef.restype = C.c_int
        L.ffo_hevc_idct.argtypes = [C.c_int, i16p, C.c_int]
        L.ffo_hevc_idct.restype = None
        L.ffo_hevc_idct_dc.argtypes = [C.c_int, i16p]
        L.ffo_hevc_idct_dc.restype = None
        L.ffo_hevc_transform_4x4_luma.argtypes = [i16p]
        L.ffo_hevc_transform_4x4_luma.restype = None
        L.ffo_hevc_add_residual.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
        L.ffo_hevc_add_residual.restype = None
        L.ffo_hevc_mc.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_hevc_mc.restype = None
        L.ffo_hevc_mc_w.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, C.c_ssize_t, i16p] + [C.c_int] * 8
        L.ffo_hevc_mc_w.restype = None
        L.ffo_vp9_itxfm_add.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, i16p, C.c_int]
        L.ffo_vp9_itxfm_add.restype = None
        L.ffo_vp9_mc.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_vp9_mc.restype = None
        L.ffo_vp9_smc.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, C.c_ssize_t] + [C.c_int] * 6
        L.ffo_vp9_smc.restype = None
        L.ffo_vp9_intra_pred.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, u8p]
        L.ffo_vp9_intra_pred.restype = None
        L.ffo_h264_pred4x4.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t]
        L.ffo_h264_pred8x8l.argtypes = [C.c_int, u8p, C.c_int, C.c_int, C.c_ssize_t]
        L.ffo_h264_pred8x8.argtypes = [C.c_int, u8p, C.c_ssize_t]
        L.ffo_h264_pred16x16.argtypes = [C.c_int, u8p, C.c_ssize_t]
        L.ffo_h264_pred4x4_add.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
        L.ffo_h264_pred8x8l_add.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
        L.ffo_h264_pred8x8l_filter_add.argtypes = [C.c_int, u8p, i16p, C.c_int, C.c_int, C.c_ssize_t]
        L.ffo_h264_pred8x8_add.argtypes = [C.c_int, u8p, i32p, i16p, C.c_ssize_t]
        L.ffo_h264_pred16x16_add.argtypes = [C.c_int, u8p, i32p, i16p, C.c_ssize_t]
        for _f in ("pred4x4", "pred8x8l", "pred8x8", "pred16x16", "pred4x4_add", "pred8x8l_add", "pred8x8l_filter_add", "pred8x8_add",
                   "pred16x16_add"):
            getattr(L, "ffo_h264_" + _f).restype = None
        L.ffo_vp9_loop_filter.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.ffo_vp9_loop_filter.restype = None
        for name in ("mc", "smc", "intra_pred", "loop_filter"):
            f8, fb = getattr(L, "ffo_vp9_" + name), getattr(L, "ffo_vp9_" + name + "_bd")
            fb.argtypes = [C.c_int] + list(f8.argtypes)
            fb.restype = None
        L.ffo_vp9_loopfilter_sb.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_int, C.c_int, u8p, u8p, u8p, C.c_ssize_t, C.c_ssize_t, u8p, u8p]
        L.ffo_vp9_loopfilter_sb.restype = None
        L.ffo_vp9_itxfm_add_bd.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_ssize_t, i32p, C.c_int]
        L.ffo_vp9_itxfm_add_bd.restype = None
        L.ffo_hevc_dequant.argtypes = [i16p, C.c_int]
        L.ffo_hevc_dequant.restype = None
        L.ffo_hevc_transform_rdpcm.argtypes = [i16p, C.c_int, C.c_int]
        L.ffo_hevc_transform_rdpcm.restype = None
        L.ffo_hevc_sao_edge_restore.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, i32p, C.c_int, C.c_int, u8p, u8p, u8p]
        L.ffo_hevc_sao_edge_restore.restype = None
        L.ffo_hevc_sao_band.argtypes = [u8p, u8p, C.c_ssize_t, C.c_ssize_t, i16p, C.c_int, C.c_int, C.c_int]
        L.ffo_hevc_sao_band.restype = None
        L.ffo_hevc_sao_edge.argtypes = [u8p, u8p, C.c_ssize_t, C.c_ssize_t, i16p, C.c_int, C.c_int, C.c_int]
        L.ffo_hevc_sao_edge.restype = None
        L.ffo_hevc_loop_filter.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, i32p, u8p, u8p]
        L.ffo_hevc_loop_filter.restype = None
        # the same members at a given bit depth: a leading `bd`, pixels uint16 above 8 bits, strides in bytes
        for name in ("idct", "idct_dc", "transform_4x4_luma", "add_residual", "mc", "mc_w", "sao_edge_restore", "sao_band",
                     "sao_edge", "loop_filter"):
            f8, fb = getattr(L, "ffo_hevc_" + name), getattr(L, "ffo_hevc_" + name + "_bd")
            fb.argtypes = [C.c_int] + list(f8.argtypes)
            fb.restype = None
        L.ffo_hevc_dequant_bd.argtypes = [C.c_int, i16p, C.c_int]
        L.ffo_hevc_dequant_bd.restype = None
        L.ffo_h264_biweight.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_h264_biweight.restype = None
        L.ffo_h264_deblock_frame.argtypes = [u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p]
        L.ffo_h264_deblock_frame.restype = None
        L.ffo_h264_deblock_frame_chroma.argtypes = [u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p]
        L.ffo_h264_deblock_frame_chroma.restype = None
        L.ffo_sad.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t, C.c_int]
        L.ffo_hadamard8_diff8x8.argtypes = [u8p, u8p, C.c_ssize_t]
        L.ffo_hadamard8_diff16.argtypes = [u8p, u8p, C.c_ssize_t, C.c_int]
        L.ffo_me_search_esa.argtypes = [u8p, u8p] + [C.c_int] * 8 + [i32p]
        L.ffo_me_search_esa.restype = C.c_uint64
        L.ffo_me_esa_frame.argtypes = [u8p, u8p] + [C.c_int] * 6 + [i16p, C.POINTER(C.c_uint32)]
        L.ffo_me_esa_frame.restype = None
        L.ffo_mdct_create.argtypes = [C.c_int, C.c_int, C.c_float]
        L.ffo_mdct_create.restype = C.c_void_p
        L.ffo_fft_pfa_factor.argtypes = [C.c_int]
        L.ffo_fft_pfa_factor.restype = C.c_int
        L.ffo_mdct_pfa_factor.argtypes = [C.c_int]
        L.ffo_mdct_pfa_factor.restype = C.c_int
        L.ffo_mdct_run.argtypes = [C.c_void_p, f32p, f32p, C.c_ssize_t]
        L.ffo_mdct_run.restype = None
        L.ffo_imdct_full_run.argtypes = [C.c_void_p, f32p, f32p]
        L.ffo_imdct_full_run.restype = None
        L.ffo_mdct_free.argtypes = [C.c_void_p]
        L.ffo_mdct_free.restype = None
        L.ffo_mdct_naive_fwd.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double), f32p]
        L.ffo_mdct_naive_inv.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double), f32p]
        L.ffo_mdct_naive_fwd.restype = L.ffo_mdct_naive_inv.restype = None
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ffref_sws_create.argtypes = [C.c_int] * 8
        L.ffref_sws_create.restype = C.c_void_p
        L.ffref_sws_create_ranges.argtypes = [C.c_int] * 10
        L.ffref_sws_create_ranges.restype = C.c_void_p
        L.ffref_sws_free.argtypes = [C.c_void_p]
        L.ffref_sws_free.restype = None
        L.ffref_sws_set_colorspace.argtypes = [C.c_void_p] + [C.c_int] * 5
        L.ffref_sws_coefficients.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.ffref_sws_coefficients.restype = None
        L.ffref_sws_scale.argtypes = [C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                      C.POINTER(u8p), C.POINTER(C.c_int)]
        if hasattr(L, "ffref_h264_idct_batch"):
            L.ffref_h264_idct_batch.argtypes = [C.c_int, u8p, C.c_ssize_t, i32p, i16p, C.c_int, C.c_int]
        if hasattr(L, "ffref_h264_idct_batch_timed"):
            L.ffref_h264_idct_batch_timed.argtypes = [C.c_int, u8p, C.c_ssize_t, i32p, i16p, C.c_int, C.c_int, C.c_double,
                                                      C.POINTER(C.c_double), C.POINTER(C.c_int)]
            L.ffref_sws_scale_frames_mt.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                                                    C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
        if hasattr(L, "ffref_sws_yuv2packedX"):
            pp = C.POINTER(i16p)
            L.ffref_sws_yuv2packedX.argtypes = [C.c_void_p, i16p, pp, C.c_int, i16p, pp, pp, C.c_int, u8p, C.c_int, C.c_int]
            L.ffref_sws_yuv2packed2.argtypes = [C.c_void_p, pp, pp, pp, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
            L.ffref_sws_yuv2packed1.argtypes = [C.c_void_p, i16p, pp, pp, u8p, C.c_int, C.c_int, C.c_int]
            for f in (L.ffref_sws_yuv2packedX, L.ffref_sws_yuv2packed2, L.ffref_sws_yuv2packed1):
                f.restype = None
        L.ffref_sws_filter.argtypes = [C.c_void_p, C.c_int, C.POINTER(i16p), C.POINTER(i32p), C.POINTER(C.c_int)]
        L.ffref_sws_is_unscaled.argtypes = [C.c_void_p]
        L.ffref_sws_flags.argtypes = [C.c_void_p]
        L.ffref_sws_full_coeffs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ffref_sws_hyscale.argtypes = [C.c_void_p, i16p, C.c_int, u8p, i16p, i32p, C.c_int]
        L.ffref_sws_hyscale.restype = None
        L.ffref_sws_yuv2planeX.argtypes = [C.c_void_p, i16p, C.c_int, C.POINTER(i16p), u8p, C.c_int, u8p, C.c_int]
        L.ffref_sws_yuv2planeX.restype = None
        L.ffref_sws_yuv2plane1.argtypes = [C.c_void_p, i16p, u8p, C.c_int, u8p, C.c_int]
        L.ffref_sws_yuv2plane1.restype = None
        L.ffref_sws_yuv2nv12cX.argtypes = [C.c_void_p, C.c_int, u8p, i16p, C.c_int, C.POINTER(i16p), C.POINTER(i16p),
                                           u8p, C.c_int]
        L.ffref_sws_yuv2nv12cX.restype = None
        L.ffref_h264_idct.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
        L.ffref_h264_idct.restype = None
        L.ffref_h264_idct_multi.argtypes = [C.c_int, u8p, i32p, i16p, C.c_ssize_t, u8p]
        L.ffref_h264_idct_multi.restype = None
        L.ffref_h264_idct_add8.argtypes = [C.POINTER(u8p), i32p, i16p, C.c_ssize_t, u8p, C.c_int]
        L.ffref_h264_idct_add8.restype = None
        if hasattr(L, "ffref_h264_luma_dc_dequant_idct"):
            L.ffref_h264_luma_dc_dequant_idct.argtypes = [i16p, i16p, C.c_int]
            L.ffref_h264_luma_dc_dequant_idct.restype = None
            L.ffref_h264_chroma_dc_dequant_idct.argtypes = [i16p, C.c_int]
            L.ffref_h264_chroma_dc_dequant_idct.restype = None
            L.ffref_h264_add_pixels_clear.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
            L.ffref_h264_add_pixels_clear.restype = None
        L.ffref_h264_loop_filter.argtypes = [C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, i8p]
        L.ffref_h264_loop_filter.restype = None
        L.ffref_h264_qpel.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_ssize_t]
        L.ffref_h264_qpel.restype = None
        L.ffref_h264_chroma.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.ffref_h264_chroma.restype = None
        L.ffref_h264_weight.argtypes = [C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffref_h264_weight.restype = None
        L.ffref_fdsp.argtypes = [C.c_int, f32p, f32p, f32p, f32p, C.c_float, C.c_int]
        L.ffref_fdsp.restype = None
        L.ffref_hevc_idct.argtypes = [C.c_int, i16p, C.c_int]
        L.ffref_hevc_idct.restype = None
        L.ffref_hevc_idct_dc.argtypes = [C.c_int, i16p]
        L.ffref_hevc_idct_dc.restype = None
        L.ffref_hevc_transform_4x4_luma.argtypes = [i16p]
        L.ffref_hevc_transform_4x4_luma.restype = None
        L.ffref_hevc_add_residual.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
        L.ffref_hevc_add_residual.restype = None
        L.ffref_hevc_mc.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffref_hevc_mc.restype = None
        L.ffref_hevc_mc_w.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, C.c_ssize_t, i16p] + [C.c_int] * 8
        L.ffref_hevc_mc_w.restype = None
        L.ffref_vp9_itxfm_add.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, i16p, C.c_int]
        L.ffref_vp9_itxfm_add.restype = None
        L.ffref_vp9_mc.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffref_vp9_mc.restype = None
        L.ffref_vp9_smc.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, C.c_ssize_t] + [C.c_int] * 6
        L.ffref_vp9_smc.restype = None
        L.ffref_vp9_intra_pred.argtypes = [C.c_int, C.c_int, u8p, C.c_ssize_t, u8p, u8p]
        L.ffref_vp9_intra_pred.restype = None
        L.ffref_aac_window.argtypes = [C.c_int]
        L.ffref_aac_window.restype = f32p
        L.ffref_aac_imdct_and_windowing.argtypes = [f32p, i32p, i32p, f32p, f32p]
        L.ffref_aac_imdct_and_windowing.restype = C.c_int
        L.ffref_aac_apply_tns.argtypes = [f32p, i32p, i32p, i32p, i32p, f32p, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_int]
        L.ffref_aac_apply_tns.restype = C.c_int
        if hasattr(L, "ffref_aac_apply_prediction"):
            L.ffref_aac_apply_prediction.argtypes = [f32p, f32p, C.c_int, i32p, C.c_int, u8p, C.c_int, C.POINTER(C.c_uint16), C.c_int]
            L.ffref_aac_apply_prediction.restype = C.c_int
            L.ffref_aac_apply_dependent_coupling.argtypes = [f32p, f32p, C.c_int, u8p, C.c_int, i32p, f32p, C.POINTER(C.c_uint16)]
            L.ffref_aac_apply_dependent_coupling.restype = C.c_int
            L.ffref_aac_apply_independent_coupling.argtypes = [f32p, f32p, C.c_float, C.c_int]
            L.ffref_aac_apply_independent_coupling.restype = C.c_int
        if hasattr(L, "ffref_aac_imdct_and_windowing_ld"):
            L.ffref_aac_ld_table.argtypes = [C.c_int]
            L.ffref_aac_ld_table.restype = f32p
            L.ffref_aac_imdct_and_windowing_ld.argtypes = [f32p, C.c_int, f32p, f32p]
            L.ffref_aac_imdct_and_windowing_ld.restype = C.c_int
            L.ffref_aac_imdct_and_windowing_eld.argtypes = [C.c_int, f32p, f32p, f32p]
            L.ffref_aac_imdct_and_windowing_eld.restype = C.c_int
        if hasattr(L, "ffref_aac_imdct_and_windowing_len"):
            L.ffref_aac_window_len.argtypes = [C.c_int, C.c_int]
            L.ffref_aac_window_len.restype = f32p
            L.ffref_aac_imdct_and_windowing_len.argtypes = [C.c_int, f32p, i32p, i32p, f32p, f32p]
            L.ffref_aac_imdct_and_windowing_len.restype = C.c_int
        if hasattr(L, "ffref_aac_apply_ltp"):
            u16p_ = C.POINTER(C.c_uint16)
            L.ffref_aac_apply_mid_side_stereo.argtypes = [f32p, f32p, C.c_int, u8p, C.c_int, u8p, i32p, i32p, u16p_]
            L.ffref_aac_apply_mid_side_stereo.restype = C.c_int
            L.ffref_aac_apply_intensity_stereo.argtypes = [f32p, f32p, C.c_int, u8p, C.c_int, C.c_int, u8p, i32p, f32p, u16p_]
            L.ffref_aac_apply_intensity_stereo.restype = C.c_int
            L.ffref_aac_apply_ltp.argtypes = [f32p, f32p, C.c_int, C.c_float, C.POINTER(C.c_int8), i32p, i32p, C.c_int, C.c_int, C.c_int, u16p_, C.c_int,
                                              i32p, i32p, i32p, i32p, f32p, f32p]
            L.ffref_aac_apply_ltp.restype = C.c_int
            L.ffref_aac_update_ltp.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int]
            L.ffref_aac_update_ltp.restype = C.c_int
        L.ffref_h264_pred_set_codec.argtypes = [C.c_int]
        L.ffref_h264_pred_has.argtypes = [C.c_int, C.c_int]
        L.ffref_h264_pred4x4.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t]
        L.ffref_h264_pred8x8l.argtypes = [C.c_int, u8p, C.c_int, C.c_int, C.c_ssize_t]
        L.ffref_h264_pred8x8.argtypes = [C.c_int, u8p, C.c_ssize_t]
        L.ffref_h264_pred16x16.argtypes = [C.c_int, u8p, C.c_ssize_t]
        L.ffref_h264_pred4x4_add.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
        L.ffref_h264_pred8x8l_add.argtypes = [C.c_int, u8p, i16p, C.c_ssize_t]
        L.ffref_h264_pred8x8l_filter_add.argtypes = [C.c_int, u8p, i16p, C.c_int, C.c_int, C.c_ssize_t]
        L.ffref_h264_pred8x8_add.argtypes = [C.c_int, u8p, i32p, i16p, C.c_ssize_t]
        L.ffref_h264_pred16x16_add.argtypes = [C.c_int, u8p, i32p, i16p, C.c_ssize_t]
        for _f in ("pred4x4", "pred8x8l", "pred8x8", "pred16x16", "pred4x4_add", "pred8x8l_add", "pred8x8l_filter_add", "pred8x8_add",
                   "pred16x16_add"):
            getattr(L, "ffref_h264_" + _f).restype = None
        L.ffref_vp9_loop_filter.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.ffref_vp9_loop_filter.restype = None
        L.ffref_hevc_dequant.argtypes = [i16p, C.c_int]
        L.ffref_hevc_dequant.restype = None
        L.ffref_hevc_transform_rdpcm.argtypes = [i16p, C.c_int, C.c_int]
        L.ffref_hevc_transform_rdpcm.restype = None
        L.ffref_hevc_sao_edge_restore.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, i32p, C.c_int, C.c_int, u8p, u8p, u8p]
        L.ffref_hevc_sao_edge_restore.restype = None
        L.ffref_hevc_sao_band.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t, C.c_ssize_t, i16p, C.c_int, C.c_int, C.c_int]
        L.ffref_hevc_sao_band.restype = None
        L.ffref_hevc_sao_edge.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t, i16p, C.c_int, C.c_int, C.c_int]
        L.ffref_hevc_sao_edge.restype = None
        L.ffref_hevc_loop_filter.argtypes = [C.c_int, u8p, C.c_ssize_t, C.c_int, i32p, u8p, u8p]
        L.ffref_hevc_loop_filter.restype = None
        if hasattr(L, "ffref_vp9_loopfilter_sb"):
            L.ffref_vp9_loopfilter_sb.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_int, C.c_int, u8p, u8p, u8p, C.c_int, C.c_int, u8p, u8p]
            L.ffref_vp9_loopfilter_sb.restype = C.c_int
        if hasattr(L, "ffref_vp9_set_bit_depth"):
            L.ffref_vp9_set_bit_depth.argtypes = [C.c_int]
            L.ffref_vp9_set_bit_depth.restype = None
        if hasattr(L, "ffref_hevc_set_bit_depth"):
            L.ffref_hevc_set_bit_depth.argtypes = [C.c_int]
            L.ffref_hevc_set_bit_depth.restype = None
        if hasattr(L, "ffref_h264_hl_decode_intra_mb"):
            L.ffref_h264_hl_decode_intra_mb.argtypes = [u8p, u8p, u8p] + [C.c_int] * 8 + [u8p, C.c_uint, C.c_uint, u8p, C.c_int, i16p, i16p, i32p, u8p]
            L.ffref_h264_hl_decode_intra_mb.restype = C.c_int
        L.ffref_h264_biweight.argtypes = [C.c_int, u8p, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffref_h264_biweight.restype = None
        L.ffref_me_cmp.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_ssize_t, C.c_int]
        L.ffref_me_search_esa.argtypes = [u8p, u8p] + [C.c_int] * 7 + [i32p]
        L.ffref_me_search_esa.restype = C.c_uint64
        L.ffref_tx_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64]
        L.ffref_tx_create.restype = C.c_void_p
        if hasattr(L, "ffref_tx_create_d"):
            L.ffref_tx_create_d.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64]
            L.ffref_tx_create_d.restype = C.c_void_p
        L.ffref_tx_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t]
        L.ffref_tx_run.restype = None
        L.ffref_tx_free.argtypes = [C.c_void_p]
        L.ffref_tx_free.restype = None
        _ref = L
    return _ref


# ---------------------------------------------------------------------------------------------
# helpers shared by tests
# ---------------------------------------------------------------------------------------------
def planes(arrs):
    """list of 2-D uint8 arrays -> (uint8*[4], int[4])"""
    p = (u8p * 4)()
    s = (C.c_int * 4)()
    for i, a in enumerate(arrs):
        p[i] = ptr(a)
        s[i] = a.strides[0]
    return p, s


def ref_tables(ctx):
    """Pull the four filter banks out of a reference SwsContext as numpy copies."""
    L = ref()
    out = {}
    for w, name in enumerate(("hLum", "hChr", "vLum", "vChr")):
        f, p, n = i16p(), i32p(), C.c_int()
        fs = L.ffref_sws_filter(ctx, w, C.byref(f), C.byref(p), C.byref(n))
        out[name] = (np.ctypeslib.as_array(f, (n.value * fs,)).copy(), np.ctypeslib.as_array(p, (n.value,)).copy(), fs,
                     n.value)
    return out


# BT.601 limited-range defaults the reference derives (yuv2rgb.c:760-800)
DEFAULT_COEFFS = dict(cy=76309, oy=16 << 16, crv=89830, cbu=113537, cgu=-22049, cgv=-45756, yoffs=838)


def make_otables(srcW, srcH, srcFmt, dstW, dstH, dstFmt, flags, banks, coeffs=DEFAULT_COEFFS, ranges=None, dst_depth=8, full=None):
    """banks: dict name -> (filter int16[], pos int32[], size, n).  Keeps numpy refs alive on the struct.
    ranges = (src_range, dst_range): the oracle derives the conversion constants itself (ffo_sws_range_constants)."""
    t = OSwsTables(srcW, srcH, srcFmt, dstW, dstH, dstFmt, flags)
    if full is not None:     # the six coefficients of the full-chroma RGB writers
        t.full_chr = 1
        t.full_coef = (C.c_int * 6)(*[int(v) for v in full])
    if ranges is not None and ranges[0] != ranges[1]:
        t.src_range, t.dst_range = ranges
        lc, cc, lo, co = C.c_uint32(), C.c_uint32(), C.c_int64(), C.c_int64()
        oracle().ffo_sws_range_constants(ranges[0], dst_depth, C.byref(lc), C.byref(lo), C.byref(cc), C.byref(co))
        t.lum_rc_coeff, t.chr_rc_coeff, t.lum_rc_offset, t.chr_rc_offset = lc.value, cc.value, lo.value, co.value
    keep = []
    for name in ("hLum", "hChr", "vLum", "vChr"):
        f, p, fs, n = banks[name]
        f = np.ascontiguousarray(f, np.int16)
        p = np.ascontiguousarray(p, np.int32)
        keep += [f, p]
        setattr(t, name, OSwsFilter(ptr(f, i16p), ptr(p, i32p), fs, n))
    t.k = OYuv2RgbCoeffs(coeffs["cy"], coeffs["oy"], coeffs["crv"], coeffs["cbu"], coeffs["cgu"], coeffs["cgv"],
                         coeffs["yoffs"])
    t._keep = keep
    return t

YUVA_BASE = {33: 0, 78: 4, 79: 5}   # yuva420p / 422p / 444p -> the base formats (libavutil/pixfmt.h)


def alloc_frame(fmt, w, h, rng=None, pad=0):
    """Allocate the planes of one frame (strides = width + pad). Random content when rng is given."""
    def mk(r, c):
        a = np.zeros((r, c + pad), np.uint8)
        if rng is not None:
            a[:] = rng.integers(0, 256, a.shape, dtype=np.uint8)
        return a
    if fmt in YUVA_BASE:
        return alloc_frame(YUVA_BASE[fmt], w, h, rng, pad) + [mk(h, w)]
    hs, vs = (0, 0) if fmt == PIX["yuv444p"] else (1, 0) if fmt == PIX["yuv422p"] else (1, 1)
    cw, ch = -((-w) >> hs), -((-h) >> vs)
    if fmt in (PIX["yuv420p"], PIX["yuv422p"], PIX["yuv444p"]):
        return [mk(h, w), mk(ch, cw), mk(ch, cw)]
    if fmt in (PIX["nv12"], PIX["nv21"]):
        return [mk(h, w), mk(ch, 2 * cw)]
    if fmt == PIX["gbrp"]:
        return [mk(h, w), mk(h, w), mk(h, w)]
    return [mk(h, (3 if fmt in (PIX["rgb24"], PIX["bgr24"]) else 4) * w)]
