"""Host-side mirror of the reference's H264DSPContext / H264QpelContext batch faces (device tensors)."""
import ctypes as C

import numpy as np

from . import _lib

IDCT4, IDCT8, IDCT4_DC, IDCT8_DC, ADD_PIXELS4_CLEAR, ADD_PIXELS8_CLEAR = 0, 1, 2, 3, 4, 5


def _stream(stream):
    import torch
    return torch.cuda.current_stream().cuda_stream if stream is None else stream


def idct_add_batch(kind, dst, stride, dst_offset, blocks, stream=None):
    """dst: uint8 cuda tensor (plane), dst_offset: int32 [n], blocks: int16 [n, 16|64] (zeroed on return)."""
    n = dst_offset.numel()
    return _lib.check(_lib.lib().ffhip_h264_idct_add_batch_dev(kind, dst.data_ptr(), stride, dst_offset.data_ptr(),
                                                               blocks.data_ptr(), n, _stream(stream)),
                      "ffhip_h264_idct_add_batch_dev")


def idct_add8_batch(cb, cr, stride, mb_offset, blockoffset48, blocks, nnzc, stream=None):
    """idct_add8 (4:2:0) over a batch: blocks int16 [nmb, 768] (sl->mb), nnzc uint8 [nmb, 120], blockoffset48 int32 [48]."""
    nmb = mb_offset.numel()
    return _lib.check(_lib.lib().ffhip_h264_idct_add8_batch_dev(cb.data_ptr(), cr.data_ptr(), stride, mb_offset.data_ptr(),
                                                                blockoffset48.data_ptr(), blocks.data_ptr(), nnzc.data_ptr(), nmb,
                                                                _stream(stream)), "ffhip_h264_idct_add8_batch_dev")


def luma_dc_dequant_batch(output, inp, qmul, stream=None):
    """output int16 [n, 256], inp int16 [n, 16], qmul int32 [n]"""
    n = qmul.numel()
    return _lib.check(_lib.lib().ffhip_h264_luma_dc_dequant_idct_batch_dev(output.data_ptr(), output.stride(0), inp.data_ptr(), inp.stride(0),
                                                                           qmul.data_ptr(), n, _stream(stream)),
                      "ffhip_h264_luma_dc_dequant_idct_batch_dev")


def chroma_dc_dequant_batch(blocks, block_offset, qmul, stream=None):
    """in place on blocks[block_offset[m] + {0, 16, 32, 48}]"""
    n = qmul.numel()
    return _lib.check(_lib.lib().ffhip_h264_chroma_dc_dequant_idct_batch_dev(blocks.data_ptr(), block_offset.data_ptr(), qmul.data_ptr(), n,
                                                                             _stream(stream)), "ffhip_h264_chroma_dc_dequant_idct_batch_dev")


def idct_add_mb_batch(which, dst, stride, mb_offset, blockoffset16, blocks, nnzc, stream=None):
    nmb = mb_offset.numel()
    return _lib.check(_lib.lib().ffhip_h264_idct_add_mb_batch_dev(which, dst.data_ptr(), stride, mb_offset.data_ptr(),
                                                                  blockoffset16.data_ptr(), blocks.data_ptr(),
                                                                  nnzc.data_ptr(), nmb, _stream(stream)),
                      "ffhip_h264_idct_add_mb_batch_dev")


def loop_filter_batch(base, stride, edges, n, stream=None):
    return _lib.check(_lib.lib().ffhip_h264_loop_filter_batch_dev(base.data_ptr(), stride, edges.data_ptr(), n,
                                                                  _stream(stream)), "ffhip_h264_loop_filter_batch_dev")


def deblock_frame(luma, stride, mb_w, mb_h, edges, stream=None):
    return _lib.check(_lib.lib().ffhip_h264_deblock_frame_dev(luma.data_ptr(), stride, mb_w, mb_h, edges.data_ptr(),
                                                              _stream(stream)), "ffhip_h264_deblock_frame_dev")


def deblock_frames_chroma(plane, frame_pitch, nframes, stride, mb_w, mb_h, edges, stream=None):
    """one 4:2:0 chroma plane per frame (8x8 samples per MB), decoder order; edges: uint8 [nframes * mb_w * mb_h * 4, 12]"""
    return _lib.check(_lib.lib().ffhip_h264_deblock_frames_chroma_dev(plane.data_ptr(), frame_pitch, nframes, stride, mb_w, mb_h,
                                                                      edges.data_ptr(), _stream(stream)),
                      "ffhip_h264_deblock_frames_chroma_dev")


def deblock_frames(luma, frame_pitch, nframes, stride, mb_w, mb_h, edges, stream=None):
    """nframes independent pictures in one launch (edges: nframes * mb_w*mb_h*8 records)."""
    return _lib.check(_lib.lib().ffhip_h264_deblock_frames_dev(luma.data_ptr(), frame_pitch, nframes, stride, mb_w, mb_h,
                                                               edges.data_ptr(), _stream(stream)),
                      "ffhip_h264_deblock_frames_dev")


def deblock_frames_hbd(bit_depth, plane, frame_pitch, nframes, stride, mb_w, mb_h, edges, chroma=False, stream=None):
    """frame-order deblocking at 8 / 9 / 10 / 12 / 14 bits (uint16 samples above 8; stride and frame pitch in bytes); chroma: one
    4:2:0 chroma plane per frame"""
    return _lib.check(_lib.lib().ffhip_h264_deblock_frames_dev_hbd(bit_depth, 1 if chroma else 0, plane.data_ptr(), frame_pitch, nframes, stride,
                                                                   mb_w, mb_h, edges.data_ptr(), _stream(stream)),
                      "ffhip_h264_deblock_frames_dev_hbd")


def qpel_batch(dst, src, stride, blocks, n, stream=None, pic=None):
    """blocks: uint8 [n, 16] FFHipQpelBlock records; pic = (pic_w, pic_h): the _pic form, which honours FFHIP_MC_EMU records"""
    if pic is not None:
        return _lib.check(_lib.lib().ffhip_h264_qpel_batch_dev_pic(dst.data_ptr(), src.data_ptr(), stride, pic[0], pic[1], blocks.data_ptr(),
                                                                   n, _stream(stream)), "ffhip_h264_qpel_batch_dev_pic")
    return _lib.check(_lib.lib().ffhip_h264_qpel_batch_dev(dst.data_ptr(), src.data_ptr(), stride, blocks.data_ptr(),
                                                           n, _stream(stream)), "ffhip_h264_qpel_batch_dev")


def chroma_mc_batch(dst, src, stride, blocks, n, stream=None, pic=None):
    """blocks: uint8 [n, 20] FFHipChromaBlock records; pic = (pic_w, pic_h) of the chroma plane: the _pic form (FFHIP_MC_EMU)"""
    if pic is not None:
        return _lib.check(_lib.lib().ffhip_h264_chroma_mc_batch_dev_pic(dst.data_ptr(), src.data_ptr(), stride, pic[0], pic[1], blocks.data_ptr(),
                                                                        n, _stream(stream)), "ffhip_h264_chroma_mc_batch_dev_pic")
    return _lib.check(_lib.lib().ffhip_h264_chroma_mc_batch_dev(dst.data_ptr(), src.data_ptr(), stride, blocks.data_ptr(), n,
                                                                _stream(stream)), "ffhip_h264_chroma_mc_batch_dev")


def weight_batch(dst, src, stride, blocks, n, stream=None):
    """blocks: uint8 [n, 20] FFHipWeightBlock records; src may be None when no record is a biweight"""
    return _lib.check(_lib.lib().ffhip_h264_weight_batch_dev(dst.data_ptr(), src.data_ptr() if src is not None else None, stride,
                                                             blocks.data_ptr(), n, _stream(stream)), "ffhip_h264_weight_batch_dev")


MC_PUT, MC_TMP, MC_AVG = 0, 1, 2
MC_EMU = 1   # FFHIP_MC_EMU: the record's footprint is read at clamped coordinates of the reference picture (include/ffhip.h)
#: FFHipQpelBlock / FFHipChromaBlock / FFHipWeightBlock / FFHipH264Edge (include/ffhip.h)
QPEL_DTYPE = np.dtype([("dst_offset", "<i4"), ("src_offset", "<i4"), ("mcxy", "u1"), ("size_idx", "u1"), ("avg", "u1"), ("flags", "u1"),
                       ("src_x", "<i2"), ("src_y", "<i2")])
CHROMA_DTYPE = np.dtype([("dst_offset", "<i4"), ("src_offset", "<i4"), ("w_idx", "u1"), ("h", "u1"), ("x", "u1"), ("y", "u1"), ("avg", "u1"),
                         ("flags", "u1"), ("src_x", "<i2"), ("src_y", "<i2"), ("pad", "<i2")])
WEIGHT_DTYPE = np.dtype([("dst_offset", "<i4"), ("src_offset", "<i4"), ("w_idx", "u1"), ("height", "u1"), ("log2_denom", "u1"), ("bi", "u1"),
                         ("weightd", "<i2"), ("weights", "<i2"), ("offset", "<i2"), ("pad", "<i2")])
EDGE_DTYPE = np.dtype([("offset", "<i4"), ("kind", "u1"), ("alpha", "u1"), ("beta", "u1"), ("pad", "u1"), ("tc0", "i1", (4,))])
assert QPEL_DTYPE.itemsize == 16 and CHROMA_DTYPE.itemsize == 20 and WEIGHT_DTYPE.itemsize == 20 and EDGE_DTYPE.itemsize == 12


class Picture:
    """ctypes mirror of FFHipH264Picture: record a picture's per-block dsp calls on the host, flush them as a handful of
    launches (include/ffhip.h).  Records are numpy structured scalars / arrays of the batch faces' dtypes."""

    def __init__(self, mb_w, mb_h, bit_depth=8, chroma_format=1):
        """chroma_format: sps->chroma_format_idc, 1 (4:2:0) or 3 (4:4:4: Cb / Cr through the luma members)"""
        self._p = _lib.vp()
        _lib.check(_lib.lib().ffhip_h264_picture_create_fmt(C.byref(self._p), mb_w, mb_h, bit_depth, chroma_format), "ffhip_h264_picture_create_fmt")

    def close(self):
        if getattr(self, "_p", None) is not None and self._p and _lib is not None:   # _lib is gone during interpreter shutdown
            _lib.lib().ffhip_h264_picture_free(C.byref(self._p))
        self._p = None

    __del__ = close

    def begin(self):
        _lib.lib().ffhip_h264_picture_begin(self._p)

    def mc_luma(self, stage, rec):
        return _lib.check(_lib.lib().ffhip_h264_picture_mc_luma(self._p, stage, rec.ctypes.data), "ffhip_h264_picture_mc_luma")

    def mc_luma_plane(self, plane, stage, rec):
        return _lib.check(_lib.lib().ffhip_h264_picture_mc_luma_plane(self._p, plane, stage, rec.ctypes.data), "ffhip_h264_picture_mc_luma_plane")

    def mc_chroma(self, plane, stage, rec):
        return _lib.check(_lib.lib().ffhip_h264_picture_mc_chroma(self._p, plane, stage, rec.ctypes.data), "ffhip_h264_picture_mc_chroma")

    def weight(self, plane, rec):
        return _lib.check(_lib.lib().ffhip_h264_picture_weight(self._p, plane, rec.ctypes.data), "ffhip_h264_picture_weight")

    def idct_add(self, plane, kind, dst_offset, block):
        return _lib.check(_lib.lib().ffhip_h264_picture_idct_add(self._p, plane, kind, dst_offset, block.ctypes.data),
                          "ffhip_h264_picture_idct_add")

    def intra_mb(self, rec, nnzc, mb, luma_dc=None, pcm=None):
        """rec: one FFHipH264IntraMB (numpy structured, the decoder-side fields set); nnzc: uint8[120]; mb: int16[768], consumed"""
        return _lib.check(_lib.lib().ffhip_h264_picture_intra_mb(self._p, rec.ctypes.data, None if nnzc is None else nnzc.ctypes.data,
                                                                 None if mb is None else mb.ctypes.data,
                                                                 None if luma_dc is None else luma_dc.ctypes.data,
                                                                 None if pcm is None else pcm.ctypes.data), "ffhip_h264_picture_intra_mb")

    def deblock_mb(self, plane, mb_x, mb_y, edges):
        return _lib.check(_lib.lib().ffhip_h264_picture_deblock_mb(self._p, plane, mb_x, mb_y, edges.ctypes.data),
                          "ffhip_h264_picture_deblock_mb")

    def flush(self, dst, strides, ref, stream=None):
        """dst / ref: three uint8 cuda tensors each (Y, Cb, Cr) — or device addresses as ints (a field: the frame plane's address + one
        line for the bottom field, with twice the line size as stride); strides: the row pitches in bytes"""
        ptr = lambda t: t if isinstance(t, int) else t.data_ptr()
        dp = (C.c_void_p * 3)(*[ptr(t) for t in dst])
        rp = (C.c_void_p * 3)(*[ptr(t) for t in ref])
        st = (C.c_int * 3)(*strides)
        return _lib.check(_lib.lib().ffhip_h264_picture_flush(self._p, dp, st, rp, _stream(stream)), "ffhip_h264_picture_flush")


def pictures_flush(pics, dsts, strides, refs, stream=None):
    """ffhip_h264_pictures_flush: Picture objects flushed together; dsts / refs: per picture three uint8 cuda tensors"""
    n = len(pics)
    pp = (C.c_void_p * n)(*[p._p for p in pics])
    dp = (C.c_void_p * (3 * n))(*[t.data_ptr() for d in dsts for t in d])
    rp = (C.c_void_p * (3 * n))(*[t.data_ptr() for r in refs for t in r])
    st = (C.c_int * 3)(*strides)
    return _lib.check(_lib.lib().ffhip_h264_pictures_flush(pp, n, dp, st, rp, _stream(stream)), "ffhip_h264_pictures_flush")


# ---- H264PredContext (include/ffhip.h; libavcodec/h264pred.h:92-116) ----
PRED4x4, PRED8x8L, PRED8x8, PRED16x16, PRED4x4_ADD, PRED8x8L_ADD, PRED8x8L_FILTER_ADD, PRED8x16 = range(8)
PRED_TOPLEFT, PRED_TOPRIGHT, PRED_TR_SPLAT = 1, 2, 4
CODEC_ID_H264 = 27
#: FFHipH264Pred
PRED_DTYPE = np.dtype([("offset", np.int32), ("aux", np.int32), ("mode", np.uint8), ("flags", np.uint8), ("pad", np.uint8, 2)])


def pred_batch(kind, plane, stride, blocks, n, coeffs=None, stream=None):
    """n independent blocks of one kind predicted in place; blocks: uint8 [n, 12] FFHipH264Pred records (device)"""
    return _lib.check(_lib.lib().ffhip_h264_pred_batch_dev(kind, plane.data_ptr(), stride, None if coeffs is None else coeffs.data_ptr(),
                                                           blocks.data_ptr(), n, None if stream is None else C.c_void_p(stream)),
                      "ffhip_h264_pred_batch_dev")


class H264PredContext(C.Structure):
    _fields_ = [("pred4x4", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t) * 15),
                ("pred8x8l", C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t) * 12),
                ("pred8x8", C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t) * 11),
                ("pred16x16", C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t) * 9),
                ("pred4x4_add", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t) * 2),
                ("pred8x8l_add", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_ssize_t) * 2),
                ("pred8x8l_filter_add", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t) * 2),
                ("pred8x8_add", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t) * 3),
                ("pred16x16_add", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t) * 3)]


def pred_init(codec_id=CODEC_ID_H264, bit_depth=8, chroma_format_idc=1):
    h = H264PredContext()
    _lib.check(_lib.lib().ff_h264_pred_init_hip(C.byref(h), codec_id, bit_depth, chroma_format_idc), "ff_h264_pred_init_hip")
    return h
