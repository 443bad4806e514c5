"""ctypes mirror of the hevcdsp inverse-transform faces of libffhip (include/ffhip.h): HEVCDSPContext.idct / idct_dc /
transform_4x4_luma / add_residual (libavcodec/hevc/dsp.h:46-61).  bit_depth = 8 (uint8 planes) or 10 / 12 (uint16 planes; strides and
record offsets stay in bytes): the *_hbd entry points."""
import ctypes as C

import numpy as np

from . import _lib

IDCT, IDCT_DC, DST_4X4, ADD_ONLY, DEQUANT, RDPCM_H, RDPCM_V = 0, 1, 2, 3, 4, 5, 6

#: FFHipHevcTU (include/ffhip.h)
TU_DTYPE = np.dtype([("coeff_offset", np.int32), ("dst_offset", np.int32), ("col_limit", np.int32)])


def _stream(stream):
    return None if stream is None else C.c_void_p(stream)


def idct_batch(kind, log2_size, coeffs, dst, stride, tus, n, stream=None, bit_depth=8):
    """coeffs: int16 device tensor (transformed in place); dst: uint8 / uint16 device tensor or None; tus: uint8 [n, 12] FFHipHevcTU"""
    return _lib.check(_lib.lib().ffhip_hevc_idct_batch_dev_hbd(bit_depth, kind, log2_size, coeffs.data_ptr(),
                                                               dst.data_ptr() if dst is not None else None, stride, tus.data_ptr(), n,
                                                               _stream(stream)), "ffhip_hevc_idct_batch_dev_hbd")


LF_H_LUMA, LF_V_LUMA, LF_H_CHROMA, LF_V_CHROMA = 0, 1, 2, 3

#: FFHipHevcEdge (include/ffhip.h)
EDGE_DTYPE = np.dtype([("offset", np.int32), ("kind", np.uint8), ("beta", np.uint8), ("no_p", np.uint8, 2), ("no_q", np.uint8, 2),
                       ("tc", np.int16, 2), ("pad", np.uint8, 2)])


def loop_filter_batch(base, stride, edges, n, stream=None, bit_depth=8):
    """edges: uint8 [n, 16] FFHipHevcEdge records whose pixels are disjoint (one direction of a picture per call)"""
    if bit_depth == 8:
        return _lib.check(_lib.lib().ffhip_hevc_loop_filter_batch_dev(base.data_ptr(), stride, edges.data_ptr(), n, _stream(stream)),
                          "ffhip_hevc_loop_filter_batch_dev")
    return _lib.check(_lib.lib().ffhip_hevc_loop_filter_batch_dev_hbd(bit_depth, base.data_ptr(), stride, edges.data_ptr(), n, _stream(stream)),
                      "ffhip_hevc_loop_filter_batch_dev_hbd")


#: FFHipHevcSao (include/ffhip.h)
SAO_DTYPE = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("offset_val", np.int16, 5), ("edge", np.uint8), ("cls", np.uint8),
                      ("width", np.uint8), ("height", np.uint8), ("pad", np.uint8, 2)])


def sao_batch(dst, stride_dst, src, stride_src, blocks, n, stream=None, bit_depth=8):
    """blocks: uint8 [n, 24] FFHipHevcSao records"""
    if bit_depth == 8:
        return _lib.check(_lib.lib().ffhip_hevc_sao_batch_dev(dst.data_ptr(), stride_dst, src.data_ptr(), stride_src, blocks.data_ptr(), n,
                                                              _stream(stream)), "ffhip_hevc_sao_batch_dev")
    return _lib.check(_lib.lib().ffhip_hevc_sao_batch_dev_hbd(bit_depth, dst.data_ptr(), stride_dst, src.data_ptr(), stride_src, blocks.data_ptr(),
                                                              n, _stream(stream)), "ffhip_hevc_sao_batch_dev_hbd")


#: FFHipHevcMcBlock (include/ffhip.h)
MC_DTYPE = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("width", np.uint8), ("height", np.uint8), ("mx", np.uint8),
                     ("my", np.uint8)])


def mc_batch(chroma, uni, dst, dststride, src, srcstride, blocks, n, stream=None, bit_depth=8):
    """blocks: uint8 [n, 12] FFHipHevcMcBlock records; dst: pixels (uni) or int16 (plain, rows 64 elements apart) device tensor"""
    if bit_depth == 8:
        return _lib.check(_lib.lib().ffhip_hevc_mc_batch_dev(chroma, uni, dst.data_ptr(), dststride, src.data_ptr(), srcstride, blocks.data_ptr(),
                                                             n, _stream(stream)), "ffhip_hevc_mc_batch_dev")
    return _lib.check(_lib.lib().ffhip_hevc_mc_batch_dev_hbd(bit_depth, chroma, uni, dst.data_ptr(), dststride, src.data_ptr(), srcstride,
                                                             blocks.data_ptr(), n, _stream(stream)), "ffhip_hevc_mc_batch_dev_hbd")


class SAOParams(C.Structure):
    """FFHipSAOParams == SAOParams (libavcodec/hevc/dsp.h:34-46)"""
    _fields_ = [("offset_abs", C.c_int * 4 * 3), ("offset_sign", C.c_int * 4 * 3), ("band_position", C.c_uint8 * 3), ("eo_class", C.c_int * 3),
                ("offset_val", C.c_int16 * 5 * 3), ("type_idx", C.c_uint8 * 3)]


#: FFHipHevcSaoRestore (include/ffhip.h)
RESTORE_DTYPE = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("offset0", np.int16), ("width", np.uint8), ("height", np.uint8),
                          ("eo", np.uint8), ("variant", np.uint8), ("borders", np.uint8), ("vert_edge", np.uint8), ("horiz_edge", np.uint8),
                          ("diag_edge", np.uint8), ("pad", np.uint8, 2)])


def sao_restore_batch(dst, stride_dst, src, stride_src, blocks, n, stream=None, bit_depth=8):
    """blocks: uint8 [n, 20] FFHipHevcSaoRestore records"""
    if bit_depth == 8:
        return _lib.check(_lib.lib().ffhip_hevc_sao_restore_batch_dev(dst.data_ptr(), stride_dst, src.data_ptr(), stride_src, blocks.data_ptr(),
                                                                      n, _stream(stream)), "ffhip_hevc_sao_restore_batch_dev")
    return _lib.check(_lib.lib().ffhip_hevc_sao_restore_batch_dev_hbd(bit_depth, dst.data_ptr(), stride_dst, src.data_ptr(), stride_src,
                                                                      blocks.data_ptr(), n, _stream(stream)), "ffhip_hevc_sao_restore_batch_dev_hbd")


MC_UNI_W, MC_BI, MC_BI_W = 2, 3, 4

#: FFHipHevcMcWBlock (include/ffhip.h)
MCW_DTYPE = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("src2_offset", np.int32), ("width", np.uint8), ("height", np.uint8),
                      ("mx", np.uint8), ("my", np.uint8), ("wx0", np.int16), ("wx1", np.int16), ("ox", np.int16), ("denom", np.uint8),
                      ("pad", np.uint8)])


def mc_w_batch(chroma, mode, dst, dststride, src, srcstride, src2, blocks, n, stream=None, bit_depth=8):
    """blocks: uint8 [n, 24] FFHipHevcMcWBlock records; src2: int16 device tensor (the other list's put_hevc_* output) or None for uni_w"""
    s2 = src2.data_ptr() if src2 is not None else None
    if bit_depth == 8:
        return _lib.check(_lib.lib().ffhip_hevc_mc_w_batch_dev(chroma, mode, dst.data_ptr(), dststride, src.data_ptr(), srcstride, s2,
                                                               blocks.data_ptr(), n, _stream(stream)), "ffhip_hevc_mc_w_batch_dev")
    return _lib.check(_lib.lib().ffhip_hevc_mc_w_batch_dev_hbd(bit_depth, chroma, mode, dst.data_ptr(), dststride, src.data_ptr(), srcstride, s2,
                                                               blocks.data_ptr(), n, _stream(stream)), "ffhip_hevc_mc_w_batch_dev_hbd")


_UNI_W = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int)
_BI = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int)
_BI_W = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ssize_t,
                    C.c_ssize_t, C.c_int)


class HEVCDSPContext(C.Structure):
    """FFHipHEVCDSPContext: host-pointer faces with the reference's signatures"""
    _fields_ = [("add_residual", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t) * 4),
                ("transform_4x4_luma", C.CFUNCTYPE(None, C.c_void_p)),
                ("idct", C.CFUNCTYPE(None, C.c_void_p, C.c_int) * 4),
                ("idct_dc", C.CFUNCTYPE(None, C.c_void_p) * 4)] + \
               [(nm, C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
                 if "luma" in nm else C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p, C.c_void_p))
                for nm in ("hevc_h_loop_filter_luma", "hevc_v_loop_filter_luma", "hevc_h_loop_filter_chroma", "hevc_v_loop_filter_chroma",
                           "hevc_h_loop_filter_luma_c", "hevc_v_loop_filter_luma_c", "hevc_h_loop_filter_chroma_c",
                           "hevc_v_loop_filter_chroma_c")] + \
               [("sao_band_filter", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.c_void_p, C.c_int, C.c_int, C.c_int) * 5),
                ("sao_edge_filter", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_int, C.c_int, C.c_int) * 5),
                ("put_hevc_qpel", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int) * 2 * 2 * 10),
                ("put_hevc_qpel_uni", C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int) * 2 * 2 * 10),
                ("put_hevc_epel", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int) * 2 * 2 * 10),
                ("put_hevc_epel_uni", C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int) * 2 * 2 * 10),
                ("dequant", C.CFUNCTYPE(None, C.c_void_p, C.c_int16)),
                ("transform_rdpcm", C.CFUNCTYPE(None, C.c_void_p, C.c_int16, C.c_int)),
                ("sao_edge_restore", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                 C.c_int, C.c_void_p, C.c_void_p, C.c_void_p) * 2),
                ("put_hevc_qpel_uni_w", _UNI_W * 2 * 2 * 10), ("put_hevc_qpel_bi", _BI * 2 * 2 * 10), ("put_hevc_qpel_bi_w", _BI_W * 2 * 2 * 10),
                ("put_hevc_epel_uni_w", _UNI_W * 2 * 2 * 10), ("put_hevc_epel_bi", _BI * 2 * 2 * 10), ("put_hevc_epel_bi_w", _BI_W * 2 * 2 * 10)]


def dsp_init(bit_depth=8):
    c = HEVCDSPContext()
    _lib.check(_lib.lib().ff_hevc_dsp_init_hip(C.byref(c), bit_depth), "ff_hevc_dsp_init_hip")
    return c
