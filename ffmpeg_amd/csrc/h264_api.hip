/*
 * h264_api.hip — C-ABI entry points of the h264dsp / h264qpel part of libffhip (include/ffhip.h).
 * Batched device faces first; the signature-exact host shims (ff_h264dsp_init_hip & co) stage one
 * call's operands through device scratch and are meant for parity harnesses such as checkasm.
 */
#include <mutex>
#include <string.h>

#include "kernels/common.h"
#include "kernels/h264_kernels.h"

extern "C" int ffhip_h264_idct_add_batch_dev(int kind, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset,
                                             int16_t *blocks, int n, void *stream)
{
    if (!dst_base || !dst_offset || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_idct_add(kind, dst_base, stride, dst_offset, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_h264_idct_add_mb_batch_dev(int which, uint8_t *dst_base, ptrdiff_t stride, const int32_t *mb_offset,
                                                const int32_t *blockoffset16, int16_t *blocks, const uint8_t *nnzc,
                                                int nmb, void *stream)
{
    if (!dst_base || !mb_offset || !blockoffset16 || !blocks || !nnzc || nmb < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_idct_add_mb(which, dst_base, stride, mb_offset, blockoffset16, blocks, nnzc, nmb,
                                         (hipStream_t)stream);
}

extern "C" int ffhip_h264_idct_add8_batch_dev(uint8_t *cb_base, uint8_t *cr_base, ptrdiff_t stride, const int32_t *mb_offset,
                                              const int32_t *blockoffset48, int16_t *blocks, const uint8_t *nnzc, int nmb, void *stream)
{
    if (!cb_base || !cr_base || !mb_offset || !blockoffset48 || !blocks || !nnzc || nmb < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_idct_add8(cb_base, cr_base, stride, mb_offset, blockoffset48, blocks, nnzc, nmb, (hipStream_t)stream);
}

extern "C" int ffhip_h264_luma_dc_dequant_idct_batch_dev(int16_t *output, size_t out_pitch, const int16_t *input, size_t in_pitch,
                                                         const int32_t *qmul, int n, void *stream)
{
    if (!output || !input || !qmul || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_luma_dc_dequant(output, out_pitch, input, in_pitch, qmul, n, (hipStream_t)stream);
}

extern "C" int ffhip_h264_chroma_dc_dequant_idct_batch_dev(int16_t *blocks, const int32_t *block_offset, const int32_t *qmul, int n,
                                                           void *stream)
{
    if (!blocks || !block_offset || !qmul || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_chroma_dc_dequant(blocks, block_offset, qmul, n, (hipStream_t)stream);
}

extern "C" int ffhip_h264_loop_filter_batch_dev(uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n,
                                                void *stream)
{
    if (!base || !edges || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_loop_filter(base, stride, edges, n, (hipStream_t)stream);
}

extern "C" int ffhip_h264_deblock_frame_dev(uint8_t *luma, ptrdiff_t stride, int mb_w, int mb_h,
                                            const FFHipH264Edge *edges, void *stream)
{
    if (!luma || !edges || mb_w < 0 || mb_h < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_deblock_frame(luma, stride, mb_w, mb_h, edges, (hipStream_t)stream);
}

extern "C" int ffhip_h264_deblock_frames_chroma_dev(uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                                    const FFHipH264Edge *edges, void *stream)
{
    if (!plane || !edges || mb_w <= 0 || mb_h <= 0 || nframes < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_deblock_frames_chroma(plane, frame_pitch, nframes, stride, mb_w, mb_h, edges, (hipStream_t)stream);
}

extern "C" int ffhip_h264_deblock_frames_dev_hbd(int bit_depth, int chroma, uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w,
                                                 int mb_h, const FFHipH264Edge *edges, void *stream)
{
    if (!plane || !edges || mb_w <= 0 || mb_h <= 0 || nframes < 0 || (bit_depth != 8 && bit_depth != 9 && bit_depth != 10 && bit_depth != 12 && bit_depth != 14))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    if (bit_depth == 8)
        return chroma ? ffhip_launch_h264_deblock_frames_chroma(plane, frame_pitch, nframes, stride, mb_w, mb_h, edges, (hipStream_t)stream)
                      : ffhip_launch_h264_deblock_frames(plane, frame_pitch, nframes, stride, mb_w, mb_h, edges, (hipStream_t)stream);
    return ffhip_launch_h264_deblock_frames_bd(bit_depth, chroma, plane, frame_pitch, nframes, stride, mb_w, mb_h, edges, (hipStream_t)stream);
}

extern "C" int ffhip_h264_deblock_frames_dev(uint8_t *luma, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                             const FFHipH264Edge *edges, void *stream)
{
    if (!luma || !edges || mb_w < 0 || mb_h < 0 || nframes < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_deblock_frames(luma, frame_pitch, nframes, stride, mb_w, mb_h, edges, (hipStream_t)stream);
}

extern "C" int ffhip_h264_qpel_batch_dev(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks,
                                         int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_qpel(dst, src, stride, blocks, n, (hipStream_t)stream);
}

/* the FFHIP_MC_EMU forms: the reference pictures' dimensions in samples of the plane (h264_mb.c:229-247, 297-317) */
static int pic_dims_ok(const char *who, int pic_w, int pic_h)
{
    if (pic_w <= 0 || pic_h <= 0 || pic_w > 32767 || pic_h > 32767) {
        ffhip_set_error("%s: picture dimensions %d x %d", who, pic_w, pic_h);
        return 0;
    }
    return 1;
}

extern "C" int ffhip_h264_qpel_batch_dev_pic(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int pic_w, int pic_h,
                                             const FFHipQpelBlock *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0 || !pic_dims_ok("ffhip_h264_qpel_batch_dev_pic", pic_w, pic_h))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_qpel(dst, src, stride, blocks, n, (hipStream_t)stream, pic_w, pic_h);
}

extern "C" int ffhip_h264_chroma_mc_batch_dev_pic(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int pic_w, int pic_h,
                                                  const FFHipChromaBlock *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0 || !pic_dims_ok("ffhip_h264_chroma_mc_batch_dev_pic", pic_w, pic_h))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_chroma_mc(dst, src, stride, blocks, n, (hipStream_t)stream, pic_w, pic_h);
}

extern "C" int ffhip_h264_qpel_batch_dev_hbd_pic(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int pic_w, int pic_h,
                                                 const FFHipQpelBlock *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0 || !pic_dims_ok("ffhip_h264_qpel_batch_dev_hbd_pic", pic_w, pic_h))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_qpel_bd(bit_depth, dst, src, stride, blocks, n, (hipStream_t)stream, pic_w, pic_h);
}

extern "C" int ffhip_h264_chroma_mc_batch_dev_hbd_pic(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int pic_w, int pic_h,
                                                      const FFHipChromaBlock *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0 || !pic_dims_ok("ffhip_h264_chroma_mc_batch_dev_hbd_pic", pic_w, pic_h))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_chroma_mc_bd(bit_depth, dst, src, stride, blocks, n, (hipStream_t)stream, pic_w, pic_h);
}

extern "C" int ffhip_h264_chroma_mc_batch_dev(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks,
                                              int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_chroma_mc(dst, src, stride, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_h264_weight_batch_dev(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks,
                                           int n, void *stream)
{
    if (!dst || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_weight(dst, src ? src : dst, stride, blocks, n, (hipStream_t)stream);
}

/* ---- hevcdsp inverse transforms ------------------------------------------------ */
extern "C" int ffhip_hevc_idct_batch_dev(int kind, int log2_size, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride,
                                         const FFHipHevcTU *tus, int n, void *stream)
{
    if (!coeffs || !tus || n < 0 || kind < FFHIP_HEVC_IDCT || kind > FFHIP_HEVC_RDPCM_V || log2_size < 2 || log2_size > 5 ||
        (kind == FFHIP_HEVC_DST_4X4 && log2_size != 2) || (kind == FFHIP_HEVC_ADD_ONLY && !dst))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_idct(kind, log2_size, coeffs, dst, stride, tus, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_loop_filter_batch_dev(uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, void *stream)
{
    if (!base || !edges || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_loop_filter(base, stride, edges, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_sao_batch_dev(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *src, ptrdiff_t stride_src,
                                        const FFHipHevcSao *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_sao(dst, stride_dst, src, stride_src, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_scaled_mc_batch_dev(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                             const FFHipVp9ScaledBlock *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_smc(dst, dststride, src, srcstride, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_intra_pred_batch_dev(int tx, uint8_t *dst, ptrdiff_t stride, const uint8_t *edges, const FFHipVp9Intra *blocks, int n,
                                              void *stream)
{
    if (!dst || !edges || !blocks || n < 0 || tx < 0 || tx > 3)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_intra(tx, dst, stride, edges, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_h264_pred_batch_dev(int kind, uint8_t *plane, ptrdiff_t stride, int16_t *coeffs, const FFHipH264Pred *blocks, int n,
                                         void *stream)
{
    if (!plane || !blocks || n < 0 || kind < 0 || kind > FFHIP_H264_PRED_CODEC ||
        (kind >= FFHIP_H264_PRED4x4_ADD && kind <= FFHIP_H264_PRED8x8L_FILTER_ADD && !coeffs))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_pred(kind, plane, stride, coeffs, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_loop_filter_batch_dev(uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, void *stream)
{
    if (!base || !edges || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_loop_filter(base, stride, edges, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_mc_batch_dev(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                      const FFHipVp9McBlock *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_mc(dst, dststride, src, srcstride, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_itxfm_add_batch_dev(int tx, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n,
                                             void *stream)
{
    if (!coeffs || !dst || !tus || n < 0 || tx < 0 || tx > 4)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_itxfm(tx, coeffs, dst, stride, tus, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_sao_restore_batch_dev(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *src, ptrdiff_t stride_src,
                                                const FFHipHevcSaoRestore *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_sao_restore(dst, stride_dst, src, stride_src, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_mc_batch_dev(int chroma, int uni, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                       const FFHipHevcMcBlock *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_mc(chroma, uni ? 1 : 0, dst, dststride, src, srcstride, nullptr, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_mc_w_batch_dev(int chroma, int mode, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                         const int16_t *src2, const FFHipHevcMcWBlock *blocks, int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0 || mode < FFHIP_HEVC_MC_UNI_W || mode > FFHIP_HEVC_MC_BI_W)
        return FFHIP_EINVAL;
    if (mode != FFHIP_HEVC_MC_UNI_W && !src2)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_mc(chroma, mode, dst, dststride, src, srcstride, src2, blocks, n, (hipStream_t)stream);
}

/* ---- hevcdsp above 8 bits: the same batch faces with the depth the reference instantiates its templates for ---------------- */
static bool hevc_bd_ok(int bd) { return bd == 8 || bd == 10 || bd == 12; }

extern "C" int ffhip_hevc_idct_batch_dev_hbd(int bit_depth, int kind, int log2_size, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride,
                                             const FFHipHevcTU *tus, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !coeffs || !tus || n < 0 || kind < FFHIP_HEVC_IDCT || kind > FFHIP_HEVC_RDPCM_V || log2_size < 2 ||
        log2_size > 5 || (kind == FFHIP_HEVC_DST_4X4 && log2_size != 2) || (kind == FFHIP_HEVC_ADD_ONLY && !dst))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_idct_bd(bit_depth, kind, log2_size, coeffs, dst, stride, tus, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_loop_filter_batch_dev_hbd(int bit_depth, uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !base || !edges || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_loop_filter_bd(bit_depth, base, stride, edges, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_sao_batch_dev_hbd(int bit_depth, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *src, ptrdiff_t stride_src,
                                            const FFHipHevcSao *blocks, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_sao_bd(bit_depth, dst, stride_dst, src, stride_src, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_sao_restore_batch_dev_hbd(int bit_depth, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *src, ptrdiff_t stride_src,
                                                    const FFHipHevcSaoRestore *blocks, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_sao_restore_bd(bit_depth, dst, stride_dst, src, stride_src, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_mc_batch_dev_hbd(int bit_depth, int chroma, int uni, void *dst, ptrdiff_t dststride, const uint8_t *src,
                                           ptrdiff_t srcstride, const FFHipHevcMcBlock *blocks, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_mc_bd(bit_depth, chroma, uni ? 1 : 0, dst, dststride, src, srcstride, nullptr, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_hevc_mc_w_batch_dev_hbd(int bit_depth, int chroma, int mode, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src,
                                             ptrdiff_t srcstride, const int16_t *src2, const FFHipHevcMcWBlock *blocks, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !dst || !src || !blocks || n < 0 || mode < FFHIP_HEVC_MC_UNI_W || mode > FFHIP_HEVC_MC_BI_W)
        return FFHIP_EINVAL;
    if (mode != FFHIP_HEVC_MC_UNI_W && !src2)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_hevc_mc_bd(bit_depth, chroma, mode, dst, dststride, src, srcstride, src2, blocks, n, (hipStream_t)stream);
}

/* ---- vp9dsp above 8 bits: the same batch faces at the depth ff_vp9dsp_init(dsp, bpp, ...) instantiates its template for ------- */
extern "C" int ffhip_vp9_itxfm_add_batch_dev_hbd(int bit_depth, int tx, void *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n,
                                                 void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !coeffs || !dst || !tus || n < 0 || tx < 0 || tx > 4)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_itxfm_bd(bit_depth, tx, coeffs, dst, stride, tus, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_mc_batch_dev_hbd(int bit_depth, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                          const FFHipVp9McBlock *blocks, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_mc_bd(bit_depth, dst, dststride, src, srcstride, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_scaled_mc_batch_dev_hbd(int bit_depth, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                                 const FFHipVp9ScaledBlock *blocks, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_smc_bd(bit_depth, dst, dststride, src, srcstride, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_loop_filter_batch_dev_hbd(int bit_depth, uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !base || !edges || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_loop_filter_bd(bit_depth, base, stride, edges, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_intra_pred_batch_dev_hbd(int bit_depth, int tx, uint8_t *dst, ptrdiff_t stride, const uint8_t *edges,
                                                  const FFHipVp9Intra *blocks, int n, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !dst || !edges || !blocks || n < 0 || tx < 0 || tx > 3)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_intra_bd(bit_depth, tx, dst, stride, edges, blocks, n, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_loopfilter_frame_dev(int bit_depth, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t stride_y, ptrdiff_t stride_uv, int cols,
                                              int rows, const FFHipVp9LfSb *tables, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !y || !u || !v || !tables || cols < 0 || rows < 0 || rows > 8 * 2047)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_lf_frame(bit_depth, y, u, v, stride_y, stride_uv, cols, rows, tables, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_loopfilter_frame_ss_dev(int bit_depth, int ss_h, int ss_v, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t stride_y,
                                                 ptrdiff_t stride_uv, int cols, int rows, const FFHipVp9LfSb *tables, void *stream)
{
    if (ss_h == 1 && ss_v == 1)
        return ffhip_vp9_loopfilter_frame_dev(bit_depth, y, u, v, stride_y, stride_uv, cols, rows, tables, stream);
    if (!hevc_bd_ok(bit_depth) || !y || !u || !v || !tables || cols < 0 || rows < 0 || rows > 8 * 1364)
        return FFHIP_EINVAL;
    if (ss_h || ss_v) { /* 4:4:0 / 4:2:2: rectangular chroma superblocks take tables of their own */
        ffhip_set_error("ffhip_vp9_loopfilter_frame_ss_dev: chroma sub-sampling %d x %d needs the chroma tables (ffhip_vp9_loopfilter_frame_ssc_dev)", ss_h, ss_v);
        return FFHIP_EINVAL;
    }
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_lf_frame(bit_depth, y, u, v, stride_y, stride_uv, cols, rows, tables, (hipStream_t)stream, 1);
}

extern "C" int ffhip_vp9_loopfilter_frame_ssc_dev(int bit_depth, int ss_h, int ss_v, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t stride_y,
                                                  ptrdiff_t stride_uv, int cols, int rows, const FFHipVp9LfSb *tables, const FFHipVp9LfSbC *ctables,
                                                  void *stream)
{
    if (!hevc_bd_ok(bit_depth) || !y || !u || !v || !tables || !ctables || cols < 0 || rows < 0 || rows > 8 * 1364 || ss_h == ss_v ||
        ((ss_h | ss_v) & ~1))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_lf_frame_ssc(bit_depth, ss_h, ss_v, y, u, v, stride_y, stride_uv, cols, rows, tables, ctables, (hipStream_t)stream);
}

extern "C" int ffhip_vp9_loopfilter_frames_dev(int bit_depth, int ss_h, int ss_v, int npics, const FFHipVp9LfPic *pics, ptrdiff_t stride_y,
                                               ptrdiff_t stride_uv, int cols, int rows, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || npics < 0 || (npics && !pics) || cols < 0 || rows < 0 || rows > 8 * 1364)
        return FFHIP_EINVAL;
    if (ss_h != ss_v) { /* 4:4:0 / 4:2:2 need the chroma tables of their rectangular superblocks: ffhip_vp9_loopfilter_frames_ssc_dev */
        ffhip_set_error("ffhip_vp9_loopfilter_frames_dev: chroma sub-sampling %d x %d takes chroma tables: ffhip_vp9_loopfilter_frames_ssc_dev", ss_h, ss_v);
        return FFHIP_EINVAL;
    }
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_lf_frames(bit_depth, npics, pics, stride_y, stride_uv, cols, rows, (hipStream_t)stream, ss_h ? 0 : 1);
}

extern "C" int ffhip_vp9_loopfilter_frames_ssc_dev(int bit_depth, int ss_h, int ss_v, int npics, const FFHipVp9LfPicC *pics, ptrdiff_t stride_y,
                                                   ptrdiff_t stride_uv, int cols, int rows, void *stream)
{
    if (!hevc_bd_ok(bit_depth) || npics < 0 || (npics && !pics) || cols < 0 || rows < 0 || rows > 8 * 1364 || ss_h == ss_v || ((ss_h | ss_v) & ~1))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_vp9_lf_frames_ssc(bit_depth, ss_h, ss_v, npics, pics, stride_y, stride_uv, cols, rows, (hipStream_t)stream);
}

/* ---- AVFloatDSPContext vector operations ---------------------------------------- */
extern "C" int ffhip_fdsp_batch_dev(int op, float *dst, size_t dst_pitch, const float *src0, size_t pitch0, const float *src1,
                                    size_t pitch1, const float *src2, size_t pitch2, float mul, int len, int nvec, void *stream)
{
    static const int need1[7] = { 1, 0, 0, 1, 1, 1, 0 }, need2[7] = { 0, 0, 0, 1, 1, 0, 0 };
    if (op < 0 || op > FFHIP_FDSP_BUTTERFLIES || !dst || !src0 || len < 0 || nvec < 0 || (need1[op] && !src1) || (need2[op] && !src2))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_fdsp(op, dst, dst_pitch, src0, pitch0, src1, pitch1, src2, pitch2, mul, len, nvec, (hipStream_t)stream);
}

/* ---- any bit depth, MBAFF and 4:2:2 members (kernels/h264_hbd.hip) ---------------------------------------------------------- */
extern "C" int ffhip_h264_idct_add_batch_dev_hbd(int bit_depth, int kind, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset,
                                                 int16_t *blocks, int n, void *stream)
{
    if (!dst_base || !dst_offset || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_idct_add_bd(bit_depth, kind, dst_base, stride, dst_offset, blocks, n, (hipStream_t)stream);
}
extern "C" int ffhip_h264_idct_mb_batch_dev_hbd(int bit_depth, int which, uint8_t *dst_base, uint8_t *dst2, ptrdiff_t stride,
                                                const int32_t *mb_offset, const int32_t *blockoffset, int16_t *blocks, const uint8_t *nnzc,
                                                int nmb, void *stream)
{
    if (!dst_base || (which >= 3 && !dst2) || !mb_offset || !blockoffset || !blocks || !nnzc || nmb < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_idct_mb_bd(bit_depth, which, dst_base, dst2, stride, mb_offset, blockoffset, blocks, nnzc, nmb, (hipStream_t)stream);
}
extern "C" int ffhip_h264_dc_dequant_batch_dev_hbd(int bit_depth, int which, int16_t *output, size_t out_pitch, const int16_t *input,
                                                   size_t in_pitch, const int32_t *block_offset, const int32_t *qmul, int n, void *stream)
{
    if (!output || !qmul || n < 0 || (which == 0 && !input) || (which != 0 && !block_offset))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_dc_dequant_bd(bit_depth, which, output, out_pitch, input, in_pitch, block_offset, qmul, n, (hipStream_t)stream);
}
extern "C" int ffhip_h264_loop_filter_batch_dev_hbd(int bit_depth, uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n, void *stream)
{
    if (!base || !edges || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_loop_filter_bd(bit_depth, base, stride, edges, n, (hipStream_t)stream);
}
extern "C" int ffhip_h264_qpel_batch_dev_hbd(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                                             void *stream)
{
    if (!dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_qpel_bd(bit_depth, dst, src, stride, blocks, n, (hipStream_t)stream);
}
extern "C" int ffhip_h264_chroma_mc_batch_dev_hbd(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks,
                                                  int n, void *stream)
{
    if (!dst || !src || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_chroma_mc_bd(bit_depth, dst, src, stride, blocks, n, (hipStream_t)stream);
}
extern "C" int ffhip_h264_weight_batch_dev_hbd(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks,
                                               int n, void *stream)
{
    if (!dst || !blocks || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_weight_bd(bit_depth, dst, src ? src : dst, stride, blocks, n, (hipStream_t)stream);
}
