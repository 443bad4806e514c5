"""ctypes mirror of the vp9dsp inverse-transform faces of libffhip (include/ffhip.h): VP9DSPContext.itxfm_add[tx][txtp]
(libavcodec/vp9dsp.h:71-75), 8 bits."""
import ctypes as C

import numpy as np

from . import _lib

#: FFHipVp9TU (include/ffhip.h)
TU_DTYPE = np.dtype([("coeff_offset", np.int32), ("dst_offset", np.int32), ("txtp", np.uint8), ("dc_only", np.uint8), ("pad", np.uint8, 2)])
DCT_DCT, DCT_ADST, ADST_DCT, ADST_ADST = 0, 1, 2, 3
TX_4X4, TX_8X8, TX_16X16, TX_32X32, TX_WHT = 0, 1, 2, 3, 4


def _st(stream):
    return None if stream is None else C.c_void_p(stream)


def itxfm_add_batch(tx, coeffs, dst, stride, tus, n, stream=None, bit_depth=8):
    """coeffs: int16 (bit_depth 8) / int32 (10, 12) device tensor (consumed); dst: device tensor of samples; tus: uint8 [n, 12] FFHipVp9TU"""
    if bit_depth != 8:
        return _lib.check(_lib.lib().ffhip_vp9_itxfm_add_batch_dev_hbd(bit_depth, tx, coeffs.data_ptr(), dst.data_ptr(), stride, tus.data_ptr(),
                                                                       n, _st(stream)), "ffhip_vp9_itxfm_add_batch_dev_hbd")
    return _lib.check(_lib.lib().ffhip_vp9_itxfm_add_batch_dev(tx, coeffs.data_ptr(), dst.data_ptr(), stride, tus.data_ptr(), n,
                                                               None if stream is None else C.c_void_p(stream)),
                      "ffhip_vp9_itxfm_add_batch_dev")


class VP9ItxfmContext(C.Structure):
    """FFHipVP9ItxfmContext: host-pointer faces with the reference's signature"""
    _fields_ = [("itxfm_add", C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_int) * 4 * 5)]


def dsp_init(bpp=8):
    c = VP9ItxfmContext()
    _lib.check(_lib.lib().ff_vp9dsp_itxfm_init_hip(C.byref(c), bpp), "ff_vp9dsp_itxfm_init_hip")
    return c


#: FFHipVp9McBlock (include/ffhip.h)
MC_DTYPE = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("width", np.uint8), ("height", np.uint8), ("filter", np.uint8),
                     ("mx", np.uint8), ("my", np.uint8), ("avg", np.uint8), ("pad", np.uint8, 2)])
FILTER_SMOOTH, FILTER_REGULAR, FILTER_SHARP, FILTER_BILINEAR = 0, 1, 2, 3


def mc_batch(dst, dststride, src, srcstride, blocks, n, stream=None, bit_depth=8):
    """blocks: uint8 [n, 16] FFHipVp9McBlock records"""
    if bit_depth != 8:
        return _lib.check(_lib.lib().ffhip_vp9_mc_batch_dev_hbd(bit_depth, dst.data_ptr(), dststride, src.data_ptr(), srcstride,
                                                                blocks.data_ptr(), n, _st(stream)), "ffhip_vp9_mc_batch_dev_hbd")
    return _lib.check(_lib.lib().ffhip_vp9_mc_batch_dev(dst.data_ptr(), dststride, src.data_ptr(), srcstride, blocks.data_ptr(), n,
                                                        None if stream is None else C.c_void_p(stream)), "ffhip_vp9_mc_batch_dev")


class VP9McContext(C.Structure):
    """FFHipVP9McContext: mc[size][filter][avg][!!mx][!!my]"""
    _fields_ = [("mc", C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int) * 2 * 2 * 2 * 4 * 5)]


def mc_init(bpp=8):
    c = VP9McContext()
    _lib.check(_lib.lib().ff_vp9dsp_mc_init_hip(C.byref(c), bpp), "ff_vp9dsp_mc_init_hip")
    return c


#: FFHipVp9Edge (include/ffhip.h)
EDGE_DTYPE = np.dtype([("offset", np.int32), ("wd_idx", np.uint8), ("dir", np.uint8), ("E", np.uint8), ("I", np.uint8), ("H", np.uint8),
                       ("pad", np.uint8, 3)])


def loop_filter_batch(base, stride, edges, n, stream=None, bit_depth=8):
    """edges: uint8 [n, 12] FFHipVp9Edge records (8-sample segments that share no sample)"""
    if bit_depth != 8:
        return _lib.check(_lib.lib().ffhip_vp9_loop_filter_batch_dev_hbd(bit_depth, base.data_ptr(), stride, edges.data_ptr(), n, _st(stream)),
                          "ffhip_vp9_loop_filter_batch_dev_hbd")
    return _lib.check(_lib.lib().ffhip_vp9_loop_filter_batch_dev(base.data_ptr(), stride, edges.data_ptr(), n,
                                                                 None if stream is None else C.c_void_p(stream)),
                      "ffhip_vp9_loop_filter_batch_dev")


def lf_sb_tables(filters, sb_cols, sb_rows, lim_lut, mblim_lut):
    """host side of the decoder-order loop filter: filters = uint8 numpy [sb_rows * sb_cols, 192] VP9Filter records in raster
    order, the frame's filter_lut (uint8 [64] each) -> uint32 numpy [sb_rows * sb_cols, 320] FFHipVp9LfSb tables (4:2:0)"""
    import numpy as np
    filters = np.ascontiguousarray(filters, np.uint8).reshape(sb_rows * sb_cols, 192)
    lim_lut, mblim_lut = np.ascontiguousarray(lim_lut, np.uint8), np.ascontiguousarray(mblim_lut, np.uint8)
    out = np.zeros((sb_rows * sb_cols, 320), np.uint32)
    L = _lib.lib()
    for r in range(sb_rows):
        for c in range(sb_cols):
            k = r * sb_cols + c
            _lib.check(L.ffhip_vp9_lf_sb_tables(out[k].ctypes.data, filters[k].ctypes.data, 8 * r, 8 * c, 1, 1, lim_lut.ctypes.data,
                                                mblim_lut.ctypes.data), "ffhip_vp9_lf_sb_tables")
    return out


def loopfilter_frame(y, u, v, stride_y, stride_uv, cols, rows, tables, stream=None, bit_depth=8, ss=(1, 1)):
    """ff_vp9_loopfilter_sb over a picture of cols x rows 8x8 blocks in the decoder's order, one launch: y / u / v device tensors,
    strides in bytes, tables = device uint32 [sb_rows * sb_cols, 320] from lf_sb_tables (sb_* = (* + 7) >> 3); ss = (ss_h, ss_v):
    (1, 1) 4:2:0, (0, 0) 4:4:4 (all planes by the luma tables)"""
    if tuple(ss) != (1, 1):
        return _lib.check(_lib.lib().ffhip_vp9_loopfilter_frame_ss_dev(bit_depth, ss[0], ss[1], y.data_ptr(), u.data_ptr(), v.data_ptr(), stride_y,
                                                                       stride_uv, cols, rows, tables.data_ptr(), _st(stream)),
                          "ffhip_vp9_loopfilter_frame_ss_dev")
    return _lib.check(_lib.lib().ffhip_vp9_loopfilter_frame_dev(bit_depth, y.data_ptr(), u.data_ptr(), v.data_ptr(), stride_y, stride_uv, cols,
                                                                rows, tables.data_ptr(), _st(stream)), "ffhip_vp9_loopfilter_frame_dev")


class LfPic(C.Structure):   # FFHipVp9LfPic (include/ffhip.h)
    _fields_ = [("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("tables", C.c_void_p)]


def loopfilter_frames(pics, stride_y, stride_uv, cols, rows, stream=None, bit_depth=8, ss=(1, 1)):
    """ffhip_vp9_loopfilter_frames_dev: pics = [(y, u, v, tables)] device tensors of pictures that share the geometry; one launch"""
    arr = (LfPic * len(pics))(*[LfPic(y.data_ptr(), u.data_ptr(), v.data_ptr(), t.data_ptr()) for y, u, v, t in pics])
    return _lib.check(_lib.lib().ffhip_vp9_loopfilter_frames_dev(bit_depth, ss[0], ss[1], len(pics), C.cast(arr, C.c_void_p), stride_y, stride_uv, cols,
                                                                 rows, _st(stream)), "ffhip_vp9_loopfilter_frames_dev")


class LfPicC(C.Structure):   # == FFHipVp9LfPicC
    _fields_ = [("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("tables", C.c_void_p), ("ctables", C.c_void_p)]


def loopfilter_frames_ssc(pics, stride_y, stride_uv, cols, rows, ss, stream=None, bit_depth=8):
    """ffhip_vp9_loopfilter_frames_ssc_dev: pics = [(y, u, v, tables, ctables)] of 4:2:2 / 4:4:0 pictures that share the geometry; one launch"""
    arr = (LfPicC * len(pics))(*[LfPicC(y.data_ptr(), u.data_ptr(), v.data_ptr(), t.data_ptr(), c.data_ptr()) for y, u, v, t, c in pics])
    return _lib.check(_lib.lib().ffhip_vp9_loopfilter_frames_ssc_dev(bit_depth, ss[0], ss[1], len(pics), C.cast(arr, C.c_void_p), stride_y, stride_uv,
                                                                     cols, rows, _st(stream)), "ffhip_vp9_loopfilter_frames_ssc_dev")


_LF = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int)


class VP9LoopFilterContext(C.Structure):
    _fields_ = [("loop_filter_8", _LF * 2 * 3), ("loop_filter_16", _LF * 2), ("loop_filter_mix2", _LF * 2 * 2 * 2)]


def lf_init(bpp=8):
    c = VP9LoopFilterContext()
    _lib.check(_lib.lib().ff_vp9dsp_loopfilter_init_hip(C.byref(c), bpp), "ff_vp9dsp_loopfilter_init_hip")
    return c


#: FFHipVp9Intra (include/ffhip.h)
INTRA_DTYPE = np.dtype([("dst_offset", np.int32), ("edge_offset", np.int32), ("mode", np.uint8), ("pad", np.uint8, 3)])


def intra_pred_batch(tx, dst, stride, edges, blocks, n, stream=None, bit_depth=8):
    """edges: device tensor of edge lines (left[0..N-1], corner, top[0..max(N,8)-1] per block; samples of the depth); blocks: uint8 [n, 12]"""
    if bit_depth != 8:
        return _lib.check(_lib.lib().ffhip_vp9_intra_pred_batch_dev_hbd(bit_depth, tx, dst.data_ptr(), stride, edges.data_ptr(),
                                                                        blocks.data_ptr(), n, _st(stream)), "ffhip_vp9_intra_pred_batch_dev_hbd")
    return _lib.check(_lib.lib().ffhip_vp9_intra_pred_batch_dev(tx, dst.data_ptr(), stride, edges.data_ptr(), blocks.data_ptr(), n,
                                                                None if stream is None else C.c_void_p(stream)),
                      "ffhip_vp9_intra_pred_batch_dev")


class VP9IntraContext(C.Structure):
    _fields_ = [("intra_pred", C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p) * 15 * 4)]


def intra_init(bpp=8):
    c = VP9IntraContext()
    _lib.check(_lib.lib().ff_vp9dsp_intrapred_init_hip(C.byref(c), bpp), "ff_vp9dsp_intrapred_init_hip")
    return c


#: FFHipVp9ScaledBlock (include/ffhip.h)
SMC_DTYPE = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("width", np.uint8), ("height", np.uint8), ("filter", np.uint8),
                      ("mx", np.uint8), ("my", np.uint8), ("avg", np.uint8), ("dx", np.uint8), ("dy", np.uint8)])


def scaled_mc_batch(dst, dststride, src, srcstride, blocks, n, stream=None, bit_depth=8):
    """blocks: uint8 [n, 16] FFHipVp9ScaledBlock records"""
    if bit_depth != 8:
        return _lib.check(_lib.lib().ffhip_vp9_scaled_mc_batch_dev_hbd(bit_depth, dst.data_ptr(), dststride, src.data_ptr(), srcstride,
                                                                       blocks.data_ptr(), n, _st(stream)), "ffhip_vp9_scaled_mc_batch_dev_hbd")
    return _lib.check(_lib.lib().ffhip_vp9_scaled_mc_batch_dev(dst.data_ptr(), dststride, src.data_ptr(), srcstride, blocks.data_ptr(), n,
                                                               None if stream is None else C.c_void_p(stream)),
                      "ffhip_vp9_scaled_mc_batch_dev")


class VP9ScaledMcContext(C.Structure):
    _fields_ = [("smc", C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int) * 2 * 4 * 5)]


def smc_init(bpp=8):
    c = VP9ScaledMcContext()
    _lib.check(_lib.lib().ff_vp9dsp_scaled_mc_init_hip(C.byref(c), bpp), "ff_vp9dsp_scaled_mc_init_hip")
    return c


def lf_sb_tables_ss(filters, sb_cols, sb_rows, lim_lut, mblim_lut, ss):
    """4:2:2 (ss = (1, 0)) / 4:4:0 ((0, 1)): (tables uint32 [n, 320] with the luma part filled, ctables uint32 [n, 128]) of the picture's
    superblocks (ffhip_vp9_lf_sb_tables + ffhip_vp9_lf_sb_ctables)"""
    L = _lib.lib()
    n = sb_cols * sb_rows
    out, cout = np.zeros((n, 320), np.uint32), np.zeros((n, 128), np.uint32)
    for r in range(sb_rows):
        for c in range(sb_cols):
            k = r * sb_cols + c
            _lib.check(L.ffhip_vp9_lf_sb_tables(out[k].ctypes.data, filters[k].ctypes.data, 8 * r, 8 * c, ss[0], ss[1], lim_lut.ctypes.data,
                                                mblim_lut.ctypes.data), "ffhip_vp9_lf_sb_tables")
            _lib.check(L.ffhip_vp9_lf_sb_ctables(cout[k].ctypes.data, filters[k].ctypes.data, 8 * r, 8 * c, ss[0], ss[1], lim_lut.ctypes.data,
                                                 mblim_lut.ctypes.data), "ffhip_vp9_lf_sb_ctables")
    return out, cout


def loopfilter_frame_ssc(y, u, v, stride_y, stride_uv, cols, rows, tables, ctables, ss, stream=None, bit_depth=8):
    """ffhip_vp9_loopfilter_frame_ssc_dev: a 4:2:2 / 4:4:0 picture (rectangular chroma superblocks)"""
    return _lib.check(_lib.lib().ffhip_vp9_loopfilter_frame_ssc_dev(bit_depth, ss[0], ss[1], y.data_ptr(), u.data_ptr(), v.data_ptr(), stride_y,
                                                                    stride_uv, cols, rows, tables.data_ptr(), ctables.data_ptr(), _st(stream)),
                      "ffhip_vp9_loopfilter_frame_ssc_dev")
