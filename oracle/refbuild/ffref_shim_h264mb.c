/*
 * ffref_shim_h264mb.c — ours, TEST INFRASTRUCTURE ONLY.  Drives the reference's own ff_h264_hl_decode_mb()
 * (libavcodec/h264_mb.c:802, compiled where it lies) on ONE intra macroblock: a zeroed H264Context / H264SliceContext with just
 * the fields hl_decode_mb() reads for IS_INTRA(mb_type) filled in from the arguments.  It pins oracle/ffo_h264.c's
 * ffo_h264_hl_decode_intra_mb() — the order of the dsp calls, not only the members — to the reference.
 * 8 bits, 4:2:0, frame macroblock, deblocking_filter = 0 (no xchg_mb_border: the caller's planes hold unfiltered samples).
 */
#include <stdlib.h>
#include <string.h>

#include "libavcodec/h264dec.h"
#include "libavcodec/h264_ps.h"
#include "libavcodec/h264pred.h"
#include "libavcodec/h264dsp.h"
#include "libavcodec/videodsp.h"
#include "libavcodec/mpegutils.h"
#include "libavutil/frame.h"
#include "libavutil/mem.h"

static int decode_intra_mb(int bit_depth, uint8_t *y, uint8_t *cb, uint8_t *cr, int linesize, int uvlinesize, int mb_x, int mb_y, int mb_w,
                           int type, int intra16x16_pred_mode, int chroma_pred_mode, const uint8_t *intra4x4_pred_mode,
                           unsigned topleft_samples_available, unsigned topright_samples_available, const uint8_t *nnzc,
                           int cbp, int16_t *mb, int16_t *mb_luma_dc, const int *qmul, const uint8_t *intra_pcm_ptr);

/* type: 0 Intra16x16, 1 Intra4x4, 2 Intra4x4 + 8x8 transform, 3 I_PCM.  nnzc: 15 x 8.  mb: 3 x 256.  Returns 0. */
int ffref_h264_hl_decode_intra_mb(uint8_t *y, uint8_t *cb, uint8_t *cr, int linesize, int uvlinesize, int mb_x, int mb_y, int mb_w,
                                  int type, int intra16x16_pred_mode, int chroma_pred_mode, const uint8_t *intra4x4_pred_mode,
                                  unsigned topleft_samples_available, unsigned topright_samples_available, const uint8_t *nnzc,
                                  int cbp, int16_t *mb, int16_t *mb_luma_dc, const int *qmul, const uint8_t *intra_pcm_ptr)
{
    return decode_intra_mb(8, y, cb, cr, linesize, uvlinesize, mb_x, mb_y, mb_w, type, intra16x16_pred_mode, chroma_pred_mode, intra4x4_pred_mode,
                           topleft_samples_available, topright_samples_available, nnzc, cbp, mb, mb_luma_dc, qmul, intra_pcm_ptr);
}

/* The same at bit_depth 9 / 10 / 12 / 14 (h->pixel_shift = 1: hl_decode_mb_simple_16 / hl_decode_mb_complex): uint16_t samples,
 * linesizes in bytes, mb = 3 x 256 int32 (dctcoef), mb_luma_dc = 16 int32, intra_pcm_ptr = the 384 bit_depth-bit fields as they
 * stand in the bitstream. */
int ffref_h264_hl_decode_intra_mb_bd(int bit_depth, uint8_t *y, uint8_t *cb, uint8_t *cr, int linesize, int uvlinesize, int mb_x, int mb_y,
                                     int mb_w, int type, int intra16x16_pred_mode, int chroma_pred_mode, const uint8_t *intra4x4_pred_mode,
                                     unsigned topleft_samples_available, unsigned topright_samples_available, const uint8_t *nnzc,
                                     int cbp, int16_t *mb, int16_t *mb_luma_dc, const int *qmul, const uint8_t *intra_pcm_ptr)
{
    return decode_intra_mb(bit_depth, y, cb, cr, linesize, uvlinesize, mb_x, mb_y, mb_w, type, intra16x16_pred_mode, chroma_pred_mode,
                           intra4x4_pred_mode, topleft_samples_available, topright_samples_available, nnzc, cbp, mb, mb_luma_dc, qmul,
                           intra_pcm_ptr);
}

static int decode_intra_mb(int bit_depth, uint8_t *y, uint8_t *cb, uint8_t *cr, int linesize, int uvlinesize, int mb_x, int mb_y, int mb_w,
                           int type, int intra16x16_pred_mode, int chroma_pred_mode, const uint8_t *intra4x4_pred_mode,
                           unsigned topleft_samples_available, unsigned topright_samples_available, const uint8_t *nnzc,
                           int cbp, int16_t *mb, int16_t *mb_luma_dc, const int *qmul, const uint8_t *intra_pcm_ptr)
{
    const int ps = bit_depth > 8; /* pixel_shift */
    H264Context *h = av_mallocz(sizeof(*h));
    H264SliceContext *sl = av_mallocz(sizeof(*sl));
    SPS *sps = av_mallocz(sizeof(*sps));
    PPS *pps = av_mallocz(sizeof(*pps));
    AVFrame *f = av_frame_alloc();
    const int mb_xy = mb_x + mb_y * mb_w;
    uint32_t *mb_type = av_mallocz(sizeof(uint32_t) * (mb_xy + 1));
    uint8_t *list_counts = av_mallocz(mb_xy + 1);
    if (!h || !sl || !sps || !pps || !f || !mb_type || !list_counts)
        abort();
    sps->chroma_format_idc = 1;
    sps->bit_depth_luma = bit_depth;
    sps->bit_depth_chroma = bit_depth;
    sps->transform_bypass = 0;
    sps->profile_idc = 100;
    for (int k = 0; k < 6; k++)
        pps->dequant4_coeff[k] = pps->dequant4_buffer[k];
    sl->qscale = 26;
    sl->chroma_qp[0] = 27;
    sl->chroma_qp[1] = 28;
    pps->dequant4_buffer[0][sl->qscale][0] = qmul[0];
    pps->dequant4_buffer[1][sl->chroma_qp[0]][0] = qmul[1];
    pps->dequant4_buffer[2][sl->chroma_qp[1]][0] = qmul[2];
    h->ps.sps = sps;
    h->ps.pps = pps;
    h->pixel_shift = ps;
    h->chroma_x_shift = h->chroma_y_shift = 1;
    ff_h264dsp_init(&h->h264dsp, bit_depth, 1);
    ff_h264_pred_init(&h->hpc, AV_CODEC_ID_H264, bit_depth, 1);
    ff_videodsp_init(&h->vdsp, bit_depth);
    /* ff_h264_init_... block_offset (h264_slice.c:init_dimensions / h264dec.c): scan8-ordered 4x4 block positions */
    for (int i = 0; i < 16; i++) {
        const int x = 4 * ((i & 1) + ((i >> 2) & 1) * 2), yy = 4 * (((i >> 1) & 1) + ((i >> 3) & 1) * 2);
        h->block_offset[i] = (x << ps) + yy * linesize;
        h->block_offset[16 + i] = h->block_offset[32 + i] = (x << ps) + yy * uvlinesize;
    }
    f->data[0] = y;
    f->data[1] = cb;
    f->data[2] = cr;
    h->cur_pic.f = f;
    h->cur_pic.mb_type = mb_type;
    h->list_counts = list_counts;
    mb_type[mb_xy] = type == 0 ? MB_TYPE_INTRA16x16 : type == 1 ? MB_TYPE_INTRA4x4 : type == 2 ? (MB_TYPE_INTRA4x4 | MB_TYPE_8x8DCT)
                                                                                               : MB_TYPE_INTRA_PCM;
    sl->mb_x = mb_x;
    sl->mb_y = mb_y;
    sl->mb_xy = mb_xy;
    sl->linesize = linesize;
    sl->uvlinesize = uvlinesize;
    sl->deblocking_filter = 0;
    sl->is_complex = 0;
    sl->intra16x16_pred_mode = intra16x16_pred_mode;
    sl->chroma_pred_mode = chroma_pred_mode;
    sl->topleft_samples_available = topleft_samples_available;
    sl->topright_samples_available = topright_samples_available;
    sl->cbp = cbp;
    sl->intra_pcm_ptr = intra_pcm_ptr;
    for (int i = 0; i < 16; i++)
        sl->intra4x4_pred_mode_cache[scan8[i]] = intra4x4_pred_mode[i];
    memcpy(sl->non_zero_count_cache, nnzc, 15 * 8);
    memcpy(sl->mb, mb, (sizeof(int16_t) << ps) * 3 * 256);
    if (mb_luma_dc)
        memcpy(sl->mb_luma_dc[0], mb_luma_dc, (sizeof(int16_t) << ps) * 16);
    /* macroblock (0, 0) of planes that start at the macroblock: hl_decode_mb() adds (mb_x, mb_y) * 16 itself */
    f->data[0] = y - ((mb_x * 16 << ps) + mb_y * 16 * linesize);
    f->data[1] = cb - ((mb_x * 8 << ps) + mb_y * 8 * uvlinesize);
    f->data[2] = cr - ((mb_x * 8 << ps) + mb_y * 8 * uvlinesize);
    ff_h264_hl_decode_mb(h, sl);
    memcpy(mb, sl->mb, (sizeof(int16_t) << ps) * 3 * 256);
    av_frame_free(&f);
    av_free(mb_type);
    av_free(list_counts);
    av_free(pps);
    av_free(sps);
    av_free(sl);
    av_free(h);
    return 0;
}
