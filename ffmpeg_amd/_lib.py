"""ctypes binding of libffhip.so (the product).  Fails loudly when the library is missing: there is
no CPU fallback anywhere in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libffhip.so")
# the same sources with -DFFHIP_MEASURE: FFHIP_* environment knobs (measured kernel variants, fault injection) are live only
# there.  Test / measurement infrastructure: select("measure") is called by the tests that set a knob, never by the package.
SO_MEASURE = os.path.join(HERE, "libffhip_measure.so")

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
i16p = C.POINTER(C.c_int16)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p
EINVAL, ENOMEM, ENOSYS, EIO = -22, -12, -38, -5   # FFHIP_E* (include/ffhip.h)


class SwsFilter(C.Structure):
    _fields_ = [("filter", i16p), ("pos", i32p), ("size", C.c_int), ("n", C.c_int)]


class SwsTables(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("srcW", "srcH", "srcFormat", "dstW", "dstH", "dstFormat", "flags")] + \
               [(k, SwsFilter) for k in ("hLum", "hChr", "vLum", "vChr")] + \
               [(k, C.c_int64) for k in ("yuv2rgb_cy", "yuv2rgb_oy", "yuv2rgb_crv", "yuv2rgb_cbu", "yuv2rgb_cgu",
                                         "yuv2rgb_cgv")] + [("yuv2rgb_yoffs", C.c_int)] + \
               [("src_range", C.c_int), ("dst_range", C.c_int), ("lumConvertRange_coeff", C.c_uint32), ("chrConvertRange_coeff", C.c_uint32),
                ("lumConvertRange_offset", C.c_int64), ("chrConvertRange_offset", C.c_int64),
                ("full_chr_h_int", C.c_int), ("yuv2rgb_full", C.c_int * 6), ("dst_alpha_fill", C.c_int)]


_lib = None
_libs = {}
_which = "product"


def select(which):
    """"product" (default) or "measure": which build lib() hands out from now on.  Objects made by one build must not be
    passed to the other (each has its own per-device tables)."""
    global _lib, _which
    if which not in ("product", "measure"):
        raise ValueError(which)
    _which = which
    _lib = _libs.get(which)


def lib():
    """Load libffhip.so once and declare every entry point of include/ffhip.h."""
    global _lib
    if _lib is not None:
        return _lib
    so = SO if _which == "product" else SO_MEASURE
    if not os.path.exists(so):
        raise ImportError("ffmpeg_amd/%s is missing - build it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)" % os.path.basename(so))
    # In a process that also uses torch, torch's bundled HIP runtime must be the one the process initialises:
    # libffhip.so's DT_NEEDED libamdhip64.so.7 then binds to the copy torch already loaded (same SONAME).  The
    # other order loads /opt/rocm's runtime first, torch then brings its own, and two runtimes fight over the
    # device (torch.cuda.is_available() turns False).  A host without torch (FFmpeg itself) is unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(so)
    i64p = C.POINTER(C.c_int64)
    sig = {
        "ffhip_device_count": (C.c_int, []),
        "ffhip_set_device": (C.c_int, [C.c_int]),
        "ffhip_get_device": (C.c_int, []),
        "ffhip_stream_create": (C.c_int, [C.POINTER(vp)]),
        "ffhip_stream_order": (C.c_int, [vp, vp]),
        "ffhip_device_push": (C.c_int, [C.c_int, vp]),
        "ffhip_device_pop": (None, [C.c_int]),
        "ffhip_stream_destroy": (C.c_int, [vp]),
        "ffhip_shard_range": (None, [C.c_int64, C.c_int, C.c_int, i64p, i64p]),
        "ffhip_shard_frame_pairs": (None, [C.c_int64, C.c_int, C.c_int, i64p, i64p, i64p, i64p]),
        "ffhip_device_set_create": (C.c_int, [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]),
        "ffhip_device_set_free": (None, [C.POINTER(vp)]),
        "ffhip_device_set_size": (C.c_int, [vp]),
        "ffhip_device_set_device": (C.c_int, [vp, C.c_int]),
        "ffhip_device_set_stream": (vp, [vp, C.c_int]),
        "ffhip_device_set_bind": (C.c_int, [vp, C.c_int]),
        "ffhip_device_set_synchronize": (C.c_int, [vp]),
        "ffhip_batch_scatter": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.c_int64, C.POINTER(vp)]),
        "ffhip_batch_scatter_ranges": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.c_int64, i64p, i64p, C.POINTER(vp)]),
        "ffhip_batch_scatter_frames_for_pairs": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.c_int64, C.POINTER(vp)]),
        "ffhip_batch_gather": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.c_int64, C.POINTER(vp)]),
        "ffhip_batch_gather_ranges": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.c_int64, i64p, i64p, C.POINTER(vp)]),
        "ffhip_last_error": (C.c_char_p, []),
        "ffhip_version": (C.c_char_p, []),
        "ffhip_malloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
        "ffhip_frames_alloc": (C.c_int, [C.POINTER(vp), C.c_size_t, C.c_size_t]),
        "ffhip_frames_free": (C.c_int, [vp]),
        "ffhip_free": (C.c_int, [vp]),
        "ffhip_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t]),
        "ffhip_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t]),
        "ffhip_stream_synchronize": (C.c_int, [vp]),
        "ffhip_pointer_device": (C.c_int, [vp]),
        "ffhip_memcpy2d_h2d_async": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]),
        "ffhip_memcpy2d_d2h_async": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]),
        "ffhip_sws_getContext": (vp, [C.c_int] * 7),
        "ffhip_sws_from_tables": (vp, [C.POINTER(SwsTables)]),
        "ffhip_sws_from_tables_rgb_source": (vp, [C.POINTER(SwsTables), C.c_int, C.POINTER(C.c_int32)]),
        "ffhip_sws_set_rgb2yuv": (C.c_int, [vp, C.POINTER(C.c_int32)]),
        "ffhip_sws_yuv2rgb_coeffs": (C.c_int, [C.POINTER(SwsTables), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]),
        "ffhip_sws_set_yuv2rgb": (C.c_int, [vp, C.POINTER(SwsTables)]),
        "ffhip_sws_freeContext": (None, [vp]),
        "ffhip_sws_fast_path": (C.c_int, [vp]),
        "ffhip_sws_tuned_numbering": (C.c_int, [vp]),
        "ff_sws_init_swscale_hip": (C.c_int, [vp, C.c_int, C.c_int]),
        "ffhip_sws_yuv2packed1": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]),
        "ffhip_sws_yuv2packed2": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "ffhip_sws_yuv2packedX": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int]),
        "ffhip_sws_uops_compile": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
        "ffhip_sws_uops_free": (None, [C.POINTER(vp)]),
        "ffhip_sws_uops_block_size": (C.c_int, [vp]),
        "ffhip_sws_uops_source": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_size_t]),
        "ffhip_sws_uops_check": (C.c_int, [vp, C.c_int]),
        "ffhip_sws_uops_cache_stats": (None, [C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "ffhip_sws_uops_set_cache_dir": (C.c_int, [C.c_char_p]),
        "ffhip_sws_uops_set_fallback": (None, [vp, vp, vp]),
        "ffhip_sws_uops_func": (None, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "ffhip_sws_uops_run_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
        "ffhip_membw_probe": (C.c_int, [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
        "ffhip_sws_up2_virtual_bank_host": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
        "ffhip_sws_upn_virtual_bank_host": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "ffhip_sws_up2rgb_hco_host": (C.c_int, [vp, C.c_int, vp, C.c_int, vp]),
        "ffhip_sws_down2_virtual_bank_host": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "ffhip_sws_d32_virtual_bank_host": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "ffhip_sws_mfma_tiles_host": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]),
        "ffhip_sws_tables_create": (vp, [C.c_int] * 7),
        "ffhip_sws_tables_get": (C.c_int, [vp, C.POINTER(SwsTables)]),
        "ffhip_sws_tables_is_unscaled_yuv2rgb": (C.c_int, [vp]),
        "ffhip_sws_tables_set_ranges": (C.c_int, [vp, C.c_int, C.c_int]),
        "ffhip_sws_tables_free": (None, [vp]),
        "ffhip_sws_scale": (C.c_int, [vp, C.POINTER(u8p), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(u8p),
                                      C.POINTER(C.c_int)]),
        "ffhip_sws_scale_batch_dev": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int),
                                                C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_int),
                                                C.POINTER(C.c_size_t), vp]),
        "ffhip_sws_hscale8to15_dev": (C.c_int, [vp, C.c_int, C.c_ssize_t, vp, C.c_ssize_t, C.c_int, vp, vp, C.c_int,
                                                vp]),
        "ffhip_sws_yuv2planeX8_dev": (C.c_int, [vp, C.c_int, vp, C.c_ssize_t, vp, C.c_int, vp, C.c_int, vp]),
        "ffhip_h264_idct_add_batch_dev": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, vp, C.c_int, vp]),
        "ffhip_shim_fallbacks": (C.c_long, []),
        "ffhip_h264_idct_add8_batch_dev": (C.c_int, [vp, vp, C.c_ssize_t, vp, vp, vp, vp, C.c_int, vp]),
        "ffhip_h264_luma_dc_dequant_idct_batch_dev": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, vp, C.c_int, vp]),
        "ffhip_h264_chroma_dc_dequant_idct_batch_dev": (C.c_int, [vp, vp, vp, C.c_int, vp]),
        "ffhip_h264_idct_add_mb_batch_dev": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, vp, vp, vp, C.c_int, vp]),
        "ffhip_h264_loop_filter_batch_dev": (C.c_int, [vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_h264_deblock_frame_dev": (C.c_int, [vp, C.c_ssize_t, C.c_int, C.c_int, vp, vp]),
        "ffhip_h264_picture_create": (C.c_int, [vp, C.c_int, C.c_int]),
        "ffhip_h264_picture_create_hbd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "ffhip_h264_picture_create_fmt": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "ffhip_h264_picture_mc_luma_plane": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "ffhip_h264_intra_pack_plane": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int32]),
        "ffhip_h264_intra_planes_dev": (C.c_int, [C.c_int, C.c_int, vp, C.c_ssize_t, C.c_int, C.c_int, vp]),
        "ffhip_h264_picture_lists": (C.c_int, [vp, vp]),
        "ffhip_h264_picture_status": (C.c_int, [vp]),
        "ffhip_h264_picture_free": (None, [vp]),
        "ffhip_h264_picture_begin": (None, [vp]),
        "ffhip_h264_picture_mc_luma": (C.c_int, [vp, C.c_int, vp]),
        "ffhip_h264_picture_mc_chroma": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "ffhip_h264_picture_weight": (C.c_int, [vp, C.c_int, vp]),
        "ffhip_h264_picture_idct_add": (C.c_int, [vp, C.c_int, C.c_int, C.c_int32, vp]),
        "ffhip_h264_picture_deblock_mb": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
        "ffhip_h264_picture_flush": (C.c_int, [vp, vp, vp, vp, vp]),
        "ffhip_h264_pictures_flush": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
        "ffhip_h264_mbaff_create": (C.c_int, [vp, C.c_int, C.c_int]),
        "ffhip_h264_mbaff_create_fmt": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "ffhip_h264_mbaff_free": (None, [vp]),
        "ffhip_h264_mbaff_begin": (None, [vp]),
        "ffhip_h264_mbaff_intra_mb": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp]),
        "ffhip_h264_mbaff_filter_call": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
        "ffhip_h264_mbaff_lists": (C.c_int, [vp, vp]),
        "ffhip_h264_mbaff_flush": (C.c_int, [vp, vp, vp, vp]),
        "ffhip_h264_picture_intra_mb": (C.c_int, [vp, vp, vp, vp, vp, vp]),
        "ffhip_h264_intra_pack": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int32]),
        "ffhip_h264_intra_pack_hbd": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int32]),
        "ffhip_h264_intra_frame_dev": (C.c_int, [vp, vp, vp, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, vp, vp, vp, vp]),
        "ffhip_h264_intra_frame_dev_hbd": (C.c_int, [C.c_int, vp, vp, vp, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, vp, vp, vp, vp]),
        "ffhip_h264_intra_frames_dev": (C.c_int, [C.c_int, C.c_int, vp, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, vp]),
        "ffhip_h264_deblock_frames_chroma_dev": (C.c_int, [vp, C.c_size_t, C.c_int, C.c_ssize_t, C.c_int, C.c_int, vp, vp]),
        "ffhip_h264_deblock_frames_dev": (C.c_int, [vp, C.c_size_t, C.c_int, C.c_ssize_t, C.c_int, C.c_int, vp, vp]),
        "ffhip_h264_deblock_frames_dev_hbd": (C.c_int, [C.c_int, C.c_int, vp, C.c_size_t, C.c_int, C.c_ssize_t, C.c_int, C.c_int, vp, vp]),
        "ffhip_h264_qpel_batch_dev": (C.c_int, [vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_h264_qpel_batch_dev_pic": (C.c_int, [vp, vp, C.c_ssize_t, C.c_int, C.c_int, vp, C.c_int, vp]),
        "ffhip_h264_chroma_mc_batch_dev_pic": (C.c_int, [vp, vp, C.c_ssize_t, C.c_int, C.c_int, vp, C.c_int, vp]),
        "ffhip_h264_qpel_batch_dev_hbd_pic": (C.c_int, [C.c_int, vp, vp, C.c_ssize_t, C.c_int, C.c_int, vp, C.c_int, vp]),
        "ffhip_h264_chroma_mc_batch_dev_hbd_pic": (C.c_int, [C.c_int, vp, vp, C.c_ssize_t, C.c_int, C.c_int, vp, C.c_int, vp]),
        "ffhip_h264_picture_idct_mb": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp]),
        "ffhip_me_cmp_batch_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_me_esa_batch_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_size_t, C.c_int, C.c_int,
                                             C.c_int, C.c_int, vp, vp, vp]),
        "ffhip_tx_init": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_uint64]),
        "ffhip_tx_uninit": (None, [C.POINTER(vp)]),
        "ffhip_tx_batch_dev": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_ssize_t, C.c_int, vp]),
        "ffhip_h264_chroma_mc_batch_dev": (C.c_int, [vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_h264_weight_batch_dev": (C.c_int, [vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ff_h264chroma_init_hip": (C.c_int, [vp, C.c_int]),
        "ff_h264dsp_weight_init_hip": (C.c_int, [vp, C.c_int]),
        "ffhip_hevc_idct_batch_dev": (C.c_int, [C.c_int, C.c_int, vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ff_hevc_dsp_init_hip": (C.c_int, [vp, C.c_int]),
        "ffhip_hevc_loop_filter_batch_dev": (C.c_int, [vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_hevc_mc_batch_dev": (C.c_int, [C.c_int, C.c_int, vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_vp9_scaled_mc_batch_dev": (C.c_int, [vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ff_vp9dsp_scaled_mc_init_hip": (C.c_int, [vp, C.c_int]),
        "ffhip_vp9_intra_pred_batch_dev": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, vp, C.c_int, vp]),
        "ff_vp9dsp_intrapred_init_hip": (C.c_int, [vp, C.c_int]),
        "ffhip_aac_imdct_create": (C.c_int, [vp, vp, vp, vp, vp, C.c_float, C.c_float]),
        "ffhip_aac_imdct_free": (None, [vp]),
        "ffhip_aac_tns_filters": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
        "ffhip_aac_apply_tns_batch_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
        "ffhip_aac_imdct_and_windowing": (C.c_int, [vp, vp, vp, vp, vp, vp]),
        "ffhip_aac_imdct_and_windowing_batch_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
        "ffhip_h264_pred_batch_dev": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, vp, C.c_int, vp]),
        "ff_h264_pred_init_hip": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "ffhip_vp9_loop_filter_batch_dev": (C.c_int, [vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ff_vp9dsp_loopfilter_init_hip": (C.c_int, [vp, C.c_int]),
        "ffhip_vp9_mc_batch_dev": (C.c_int, [vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ff_vp9dsp_mc_init_hip": (C.c_int, [vp, C.c_int]),
        "ffhip_vp9_itxfm_add_batch_dev": (C.c_int, [C.c_int, vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ff_vp9dsp_itxfm_init_hip": (C.c_int, [vp, C.c_int]),
        "ffhip_hevc_sao_restore_batch_dev": (C.c_int, [vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_hevc_mc_w_batch_dev": (C.c_int, [C.c_int, C.c_int, vp, C.c_ssize_t, vp, C.c_ssize_t, vp, vp, C.c_int, vp]),
        "ffhip_hevc_sao_batch_dev": (C.c_int, [vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_hevc_idct_batch_dev_hbd": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_hevc_loop_filter_batch_dev_hbd": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_hevc_mc_batch_dev_hbd": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_hevc_sao_restore_batch_dev_hbd": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_hevc_mc_w_batch_dev_hbd": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, C.c_ssize_t, vp, C.c_ssize_t, vp, vp, C.c_int, vp]),
        "ffhip_aac_imdct_create_len": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float]),
        "ffhip_aac_coupling_bands": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp]),
        "ffhip_aac_prediction_record": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]),
        "ffhip_aac_apply_prediction_batch_dev": (C.c_int, [vp, vp, vp, C.c_int, vp]),
        "ffhip_aac_ld_create": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_float]),
        "ffhip_aac_ld_free": (None, [vp]),
        "ffhip_aac_ld_batch_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
        "ffhip_aac_ms_bands": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp]),
        "ffhip_aac_is_bands": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
        "ffhip_aac_ltp_bands": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
        "ffhip_aac_band_ops_batch_dev": (C.c_int, [vp, vp, vp, C.c_int, vp]),
        "ffhip_aac_ltp_init": (C.c_int, [vp, C.c_float]),
        "ffhip_aac_ltp_predict_batch_dev": (C.c_int, [vp, vp, vp, vp, C.c_int, vp]),
        "ffhip_aac_update_ltp_batch_dev": (C.c_int, [vp, vp, vp, C.c_int, vp]),
        "ffhip_vp9_lf_sb_tables": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
        "ffhip_vp9_lf_sb_ctables": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
        "ffhip_vp9_loopfilter_frames_ssc_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, vp]),
        "ffhip_vp9_loopfilter_frame_ssc_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, vp, vp, vp]),
        "ffhip_vp9_loopfilter_frame_dev": (C.c_int, [C.c_int, vp, vp, vp, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, vp, vp]),
        "ffhip_vp9_loopfilter_frame_ss_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, vp, vp]),
        "ffhip_vp9_loopfilter_frames_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, vp]),
        "ffhip_vp9_itxfm_add_batch_dev_hbd": (C.c_int, [C.c_int, C.c_int, vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_vp9_mc_batch_dev_hbd": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_vp9_scaled_mc_batch_dev_hbd": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_vp9_loop_filter_batch_dev_hbd": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_vp9_intra_pred_batch_dev_hbd": (C.c_int, [C.c_int, C.c_int, vp, C.c_ssize_t, vp, vp, C.c_int, vp]),
        "ffhip_hevc_sao_batch_dev_hbd": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_fdsp_batch_dev": (C.c_int, [C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.c_float, C.c_int,
                                           C.c_int, vp]),
        "ff_float_dsp_init_hip": (C.c_int, [vp]),
        "ff_h264dsp_init_hip": (C.c_int, [vp, C.c_int, C.c_int]),
        "ff_h264qpel_init_hip": (C.c_int, [vp, C.c_int]),
        "ffhip_h264_idct_add_batch_dev_hbd": (C.c_int, [C.c_int, C.c_int, vp, C.c_ssize_t, vp, vp, C.c_int, vp]),
        "ffhip_h264_idct_mb_batch_dev_hbd": (C.c_int, [C.c_int, C.c_int, vp, vp, C.c_ssize_t, vp, vp, vp, vp, C.c_int, vp]),
        "ffhip_h264_dc_dequant_batch_dev_hbd": (C.c_int, [C.c_int, C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, vp, C.c_int, vp]),
        "ffhip_h264_loop_filter_batch_dev_hbd": (C.c_int, [C.c_int, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_h264_qpel_batch_dev_hbd": (C.c_int, [C.c_int, vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_h264_chroma_mc_batch_dev_hbd": (C.c_int, [C.c_int, vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ffhip_h264_weight_batch_dev_hbd": (C.c_int, [C.c_int, vp, vp, C.c_ssize_t, vp, C.c_int, vp]),
        "ff_me_cmp_init_hip": (C.c_int, [vp]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    L._missing = missing
    _libs[_which] = L
    _lib = L
    return L


def check(rc, what="ffhip call"):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise RuntimeError("%s failed (%s): %s" % (what, rc, lib().ffhip_last_error().decode()))
    return rc


class FrameMemory:
    """A range of ffhip_frames_alloc() (device memory whose virtual and physical addresses share a large alignment) that torch can view:
    `.tensor(shape, offset)` gives uint8 tensors inside it through __cuda_array_interface__; the range lives as long as this object."""

    def __init__(self, nbytes, chunk=0):
        p = C.c_void_p()
        check(lib().ffhip_frames_alloc(C.byref(p), nbytes, chunk), "ffhip_frames_alloc")
        self.ptr, self.nbytes = p.value, nbytes

    def tensor(self, shape, offset=0):
        import torch
        n = 1
        for s_ in shape:
            n *= s_
        assert offset + n <= self.nbytes

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (self.ptr + offset, False), "version": 3, "strides": None}
        v._keep = self
        t = torch.as_tensor(v, device="cuda")
        t._ffhip_frames = self          # the tensor keeps the range alive
        return t

    def close(self):
        if self.ptr:
            lib().ffhip_frames_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
