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

/* ==================================================================================================================================
 * A persistent "decoder" for whole pictures (round 4): ONE H264Context + H264SliceContext whose macroblock-level state the test
 * fills per macroblock — what the entropy decoder and ff_h264_decode_mb_*() / fill_decode_caches() would have left in sl-> — and on
 * which the reference's own ff_h264_hl_decode_mb() (libavcodec/h264_mb.c:800) runs, inter macroblocks included: hl_motion() ->
 * mc_part() -> mc_dir_part() / mc_part_weighted() with emulated_edge_mc() on UNPADDED reference pictures, then the residual.
 *
 * Two builds of this file:
 *   - in libffref.so: the dsp tables are the reference's C ones; the planes are host memory; the result is the expected picture;
 *   - in libffref_hip.so (-DFFREF_WITH_HIP, linked against libffhip.so): record mode — integration/avcodec_h264_picture_hip.c's
 *     recording members are installed instead, the planes are DEVICE addresses that are never dereferenced, and the same
 *     ff_h264_hl_decode_mb() drives them; the records land in a libffhip picture object.
 * The test's generator supplies decoder STATE (types, motion vectors, reference indices, weights, coefficients); every decision
 * about which dsp member runs with which operands is the reference's.
 * ================================================================================================================================== */
#include "libavcodec/avcodec.h"
#ifdef FFREF_WITH_HIP
#include "avcodec_h264_picture_hip.h"
#define FN(name) ffrefhip_##name
#else
#define FN(name) ffref_##name
#endif

typedef struct FFRefH264Dec {
    H264Context *h;
    H264SliceContext *sl;
    SPS *sps;
    PPS *pps;
    AVFrame *f;
    AVCodecContext *avctx;
    H264Picture *refpics;     /* parents of the H264Ref entries (await_references reads ->parent only under frame threading) */
    int record, bit_depth, mb_w, mb_h, cfmt;
#ifdef FFREF_WITH_HIP
    FFHipH264Recorder rec;
#endif
} FFRefH264Dec;

/* the header's values of the macroblock-type flags the generator combines (libavcodec/mpegutils.h, h264dec.h) */
int FN(h264dec_mb_type_bits)(int which)
{
    static const int v[] = { MB_TYPE_16x16, MB_TYPE_16x8, MB_TYPE_8x16, MB_TYPE_8x8, MB_TYPE_P0L0, MB_TYPE_P1L0, MB_TYPE_P0L1, MB_TYPE_P1L1,
                             MB_TYPE_8x8DCT, MB_TYPE_INTRA4x4, MB_TYPE_INTRA16x16, MB_TYPE_INTRA_PCM, MB_TYPE_DIRECT2, MB_TYPE_SKIP,
                             MB_TYPE_INTERLACED };
    return which >= 0 && which < (int)(sizeof(v) / sizeof(v[0])) ? v[which] : -1;
}

void FN(h264dec_close)(FFRefH264Dec *d)
{
    if (!d)
        return;
    if (d->h) {
        av_free(d->h->cur_pic.mb_type);
        av_free(d->h->cur_pic.qscale_table);
        av_free(d->h->list_counts);
        av_free(d->h->slice_table_base);
        av_free(d->h->non_zero_count);
        av_free(d->h->cbp_table);
    }
    if (d->sl) {
        av_free(d->sl->bipred_scratchpad);
        av_free(d->sl->edge_emu_buffer);
    }
    av_frame_free(&d->f);
    av_free(d->refpics);
    av_free(d->avctx);
    av_free(d->pps);
    av_free(d->sps);
    av_free(d->sl);
    av_free(d->h);
    av_free(d);
}

FFRefH264Dec *FN(h264dec_open_fmt)(int bit_depth, int mb_w, int mb_h, int linesize, int uvlinesize, int record, int chroma_format_idc);

/* linesize / uvlinesize in bytes.  record != 0 only in the hip build. */
FFRefH264Dec *FN(h264dec_open)(int bit_depth, int mb_w, int mb_h, int linesize, int uvlinesize, int record)
{
    return FN(h264dec_open_fmt)(bit_depth, mb_w, mb_h, linesize, uvlinesize, record, 1);
}

/* ... of sps->chroma_format_idc 1 (4:2:0) or 3 (4:4:4: ff_h264_hl_decode_mb() takes hl_decode_mb_444, h264_mb.c:807-811; the chroma
 * planes have the luma geometry and uvlinesize == linesize as the decoder allocates them) */
FFRefH264Dec *FN(h264dec_open_fmt)(int bit_depth, int mb_w, int mb_h, int linesize, int uvlinesize, int record, int chroma_format_idc)
{
    const int ps = bit_depth > 8;
    FFRefH264Dec *d = av_mallocz(sizeof(*d));
    H264Context *h;
    H264SliceContext *sl;
#ifndef FFREF_WITH_HIP
    if (record)
        return NULL;
#endif
    if (!d || chroma_format_idc < 0 || chroma_format_idc > 3 || (chroma_format_idc == 3 && uvlinesize != linesize)) {
        av_free(d);
        return NULL;
    }
    d->cfmt = chroma_format_idc;
    d->h = h = av_mallocz(sizeof(*h));
    d->sl = sl = av_mallocz(sizeof(*sl));
    d->sps = av_mallocz(sizeof(SPS));
    d->pps = av_mallocz(sizeof(PPS));
    d->avctx = av_mallocz(sizeof(AVCodecContext));
    d->refpics = av_calloc(2 * 48, sizeof(H264Picture));
    d->f = av_frame_alloc();
    if (!h || !sl || !d->sps || !d->pps || !d->avctx || !d->refpics || !d->f)
        abort();
    d->record = record;
    d->bit_depth = bit_depth;
    d->mb_w = mb_w;
    d->mb_h = mb_h;
    d->sps->chroma_format_idc = chroma_format_idc;
    d->sps->bit_depth_luma = d->sps->bit_depth_chroma = bit_depth;
    d->sps->profile_idc = chroma_format_idc == 3 ? 244 : chroma_format_idc == 2 ? 122 : 100;   /* (High covers monochrome) */
    d->sps->mb_width = mb_w;
    d->sps->mb_height = mb_h;
    for (int k = 0; k < 6; k++)
        d->pps->dequant4_coeff[k] = d->pps->dequant4_buffer[k];
    for (int k = 0; k < 2; k++)               /* chroma_qp_table: the identity is enough for the loop filter's index arithmetic */
        for (int q = 0; q < QP_MAX_NUM + 1; q++)
            d->pps->chroma_qp_table[k][q] = (uint8_t)FFMIN(q + 2 * k, QP_MAX_NUM);
    h->ps.sps = d->sps;
    h->ps.pps = d->pps;
    h->avctx = d->avctx;               /* active_thread_type = 0: hl_motion() does not wait for reference rows */
    h->pixel_shift = ps;
    h->chroma_x_shift = chroma_format_idc != 3;
    h->chroma_y_shift = chroma_format_idc <= 1;   /* (monochrome is decoded into 4:2:0 frames) */
    h->mb_width = mb_w;
    h->mb_height = mb_h;
    h->mb_stride = mb_w + 1;
    h->mb_num = mb_w * mb_h;
    h->picture_structure = PICT_FRAME;
    ff_h264dsp_init(&h->h264dsp, bit_depth, chroma_format_idc);
    ff_h264qpel_init(&h->h264qpel, bit_depth);
    ff_h264chroma_init(&h->h264chroma, bit_depth);
    ff_h264_pred_init(&h->hpc, AV_CODEC_ID_H264, bit_depth, chroma_format_idc);
    ff_videodsp_init(&h->vdsp, bit_depth);
#ifdef FFREF_WITH_HIP
    if (record)
        ff_h264_hip_recorder_install(h);
#endif
    /* h->block_offset (h264_slice.c init_scan_tables / ff_h264_slice_context_init): scan8-ordered 4x4 block positions */
    for (int i = 0; i < 16; i++) {
        const int x = 4 * ((i & 1) + ((i >> 2) & 1) * 2), yy = 4 * (((i >> 1) & 1) + ((i >> 3) & 1) * 2);
        h->block_offset[i] = (x << ps) + yy * linesize;
        h->block_offset[16 + i] = h->block_offset[32 + i] = (x << ps) + yy * uvlinesize;
        /* ... and of a field macroblock: every second line (h264_slice.c init_dimensions / ff_h264_field_start) */
        h->block_offset[48 + i] = (x << ps) + 2 * yy * linesize;
        h->block_offset[48 + 16 + i] = h->block_offset[48 + 32 + i] = (x << ps) + 2 * yy * uvlinesize;
    }
    h->cur_pic.f = d->f;
    h->cur_pic.mb_type = av_calloc((size_t)h->mb_stride * (mb_h + 1) + 1, sizeof(uint32_t));
    h->cur_pic.qscale_table = av_mallocz((size_t)h->mb_stride * (mb_h + 1) + 1);
    h->list_counts = av_mallocz((size_t)h->mb_stride * (mb_h + 1) + 1);
    h->non_zero_count = av_calloc((size_t)h->mb_stride * (mb_h + 1) + 1, 48);
    h->cbp_table = av_calloc((size_t)h->mb_stride * (mb_h + 1) + 1, sizeof(uint16_t));
    sl->h264 = h;
    sl->linesize = linesize;
    sl->uvlinesize = uvlinesize;
    sl->deblocking_filter = 0;   /* reconstruction here; the in-loop filter runs over the finished picture (h264_mb.c:528: no xchg_mb_border) */
    sl->is_complex = 0;
    sl->list_count = 2;
    sl->qscale = 26;
    sl->chroma_qp[0] = 27;
    sl->chroma_qp[1] = 28;
    /* ff_h264_slice_context_init / alloc_scratch_buffers (h264_slice.c:168-200): sizes as the decoder allocates them */
    {
        const int alloc_size = FFALIGN(FFABS(linesize) + 32, 32);
        sl->bipred_scratchpad = av_mallocz(16 * 6 * (size_t)alloc_size);
        sl->edge_emu_buffer = av_mallocz((size_t)alloc_size * 2 * 21);
        if (!sl->bipred_scratchpad || !sl->edge_emu_buffer)
            abort();
    }
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < 48; i++)
            sl->ref_list[l][i].parent = &d->refpics[l * 48 + i];
    return d;
}

/* the picture being reconstructed: host planes (C tables) or device addresses (record mode) */
void FN(h264dec_set_cur)(FFRefH264Dec *d, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    d->f->data[0] = y;
    d->f->data[1] = cb;
    d->f->data[2] = cr;
    d->f->linesize[0] = d->sl->linesize;
    d->f->linesize[1] = d->f->linesize[2] = d->sl->uvlinesize;
}

/* sl->ref_list[list][idx]: a frame reference (reference = PICT_FRAME), planes with the current picture's line sizes and NO border */
void FN(h264dec_set_ref)(FFRefH264Dec *d, int list, int idx, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    H264Ref *r = &d->sl->ref_list[list][idx];
    r->data[0] = y;
    r->data[1] = cb;
    r->data[2] = cr;
    r->linesize[0] = d->sl->linesize;
    r->linesize[1] = r->linesize[2] = d->sl->uvlinesize;
    r->reference = PICT_FRAME;
    if (d->sl->ref_count[list] < (unsigned)idx + 1)
        d->sl->ref_count[list] = idx + 1;
}

/* A FIELD picture (PAFF): picture_structure = PICT_TOP_FIELD (1) / PICT_BOTTOM_FIELD (2), PICT_FRAME (3) back to frames.  The decoder then
 * runs every macroblock as a field macroblock (sl->mb_field_decoding_flag, MB_FIELD(sl)): hl_decode_mb() doubles the line sizes, takes
 * block_offset[48..] and starts odd rows one line down (h264_mb_template.c:61-78); mb_y of the calls below is the decoder's own — 2 * the
 * field's macroblock row + (bottom field) (h264_slice.c:2676-2680,2759-2761) — and h->mb_height stays the FRAME's (even). */
void FN(h264dec_set_field)(FFRefH264Dec *d, int picture_structure)
{
    d->h->picture_structure = picture_structure;
    d->sl->mb_field_decoding_flag = picture_structure != PICT_FRAME;
    d->sl->is_complex = picture_structure != PICT_FRAME;   /* h264_cavlc.c / h264_cabac.c: FRAME_MBAFF || picture_structure != PICT_FRAME */
}

/* The lossless transform bypass (round 6): profile_idc != 0 makes the stream one with qpprime_y_zero_transform_bypass_flag (sps->transform_bypass;
 * profile_idc 244: the DPCM forms of vertically / horizontally predicted intra blocks as well); `on`: the macroblocks decoded from here on
 * have QP'Y = 0 — hl_decode_mb()'s transform_bypass (h264_mb_template.c:51) — else 26 as before. */
void FN(h264dec_set_bypass)(FFRefH264Dec *d, int profile_idc, int on)
{
    d->sps->transform_bypass = profile_idc != 0;
    if (profile_idc)
        d->sps->profile_idc = profile_idc;
    d->sl->qscale = on ? 0 : 26;
    d->h->x264_build = -1;          /* (no x264 SEI: Intra8x8 DPCM blocks start from the filtered edge, h264_mb.c:641-648) */
}

/* sl->ref_list[list][idx] as a FIELD of a frame whose planes are given (h264_refs.c pic_as_field(), :44-59): the bottom field starts one
 * line down, line sizes double, reference = the parity */
void FN(h264dec_set_ref_field)(FFRefH264Dec *d, int list, int idx, uint8_t *y, uint8_t *cb, uint8_t *cr, int parity)
{
    H264Ref *r = &d->sl->ref_list[list][idx];
    const int bottom = parity == PICT_BOTTOM_FIELD;
    r->data[0] = y + (bottom ? d->sl->linesize : 0);
    r->data[1] = cb + (bottom ? d->sl->uvlinesize : 0);
    r->data[2] = cr + (bottom ? d->sl->uvlinesize : 0);
    r->linesize[0] = 2 * d->sl->linesize;
    r->linesize[1] = r->linesize[2] = 2 * d->sl->uvlinesize;
    r->reference = parity;
    if (d->sl->ref_count[list] < (unsigned)idx + 1)
        d->sl->ref_count[list] = idx + 1;
}

/* sl->pwt as pred_weight_table() / implicit_weight_table() leave it (h264_parse.c:30-118, h264_slice.c:700-760): use_weight 0 none,
 * 1 explicit, 2 implicit.  luma_weight [48][2][2], chroma_weight [48][2][2][2], implicit_weight [48][48][2] ints. */
void FN(h264dec_set_pwt)(FFRefH264Dec *d, int use_weight, int use_weight_chroma, int luma_log2_weight_denom, int chroma_log2_weight_denom,
                         const int *luma_weight, const int *chroma_weight, const int *implicit_weight)
{
    H264PredWeightTable *p = &d->sl->pwt;
    p->use_weight = use_weight;
    p->use_weight_chroma = use_weight_chroma;
    p->luma_log2_weight_denom = luma_log2_weight_denom;
    p->chroma_log2_weight_denom = chroma_log2_weight_denom;
    if (luma_weight)
        memcpy(p->luma_weight, luma_weight, sizeof(p->luma_weight));
    if (chroma_weight)
        memcpy(p->chroma_weight, chroma_weight, sizeof(p->chroma_weight));
    if (implicit_weight)
        memcpy(p->implicit_weight, implicit_weight, sizeof(p->implicit_weight));
}

#ifdef FFREF_WITH_HIP
/* record mode: a new picture into `pic` (begin() already called); ref_base[pl] is what flush() will be given as ref[pl] */
void FN(h264dec_record_begin)(FFRefH264Dec *d, void *pic, const uint8_t *ref_y, const uint8_t *ref_cb, const uint8_t *ref_cr)
{
    const uint8_t *rb[3] = { ref_y, ref_cb, ref_cr };
    ff_h264_hip_recorder_begin(&d->rec, pic, d->h, d->sl, rb);
}
int FN(h264dec_record_error)(FFRefH264Dec *d) { return d->rec.error; }
#endif

static int run_hl_decode_mb(FFRefH264Dec *d)
{
#ifdef FFREF_WITH_HIP
    if (d->record)
        return ff_h264_hip_hl_decode_mb(&d->rec, d->h, d->sl);
#endif
    ff_h264_hl_decode_mb(d->h, d->sl);
    return 0;
}

static void set_mb(FFRefH264Dec *d, int mb_x, int mb_y, int mb_type, int cbp, const uint8_t *nnzc, const void *mb)
{
    H264SliceContext *sl = d->sl;
    const int ps = d->h->pixel_shift;
    sl->mb_x = mb_x;
    sl->mb_y = mb_y;
    sl->mb_xy = mb_x + mb_y * d->h->mb_stride;
    d->h->cur_pic.mb_type[sl->mb_xy] = mb_type;
    sl->cbp = cbp;
    memcpy(sl->non_zero_count_cache, nnzc, 15 * 8);
    memcpy(sl->mb, mb, (sizeof(int16_t) << ps) * 3 * 256);
}

/* An INTER macroblock: mb_type = MB_TYPE_* flags (partition shape, P?L? direction bits, 8x8DCT), sub_mb_type[4] likewise for
 * MB_TYPE_8x8, mv_cache [2][5*8][2] int16 and ref_cache [2][5*8] int8 in scan8 layout as fill_decode_caches() + the mv decode leave
 * them (h264_mvpred.h, h264_cavlc.c:870-1050), nnzc 15 x 8, mb 3 x 256 dctcoef (returned as hl_decode_mb() leaves it),
 * qmul_cb / qmul_cr = pps->dequant4_coeff[4 / 5][chroma_qp][0].  Returns 0, or the recorder's error. */
int FN(h264dec_decode_inter)(FFRefH264Dec *d, int mb_x, int mb_y, int mb_type, const uint16_t *sub_mb_type, const int16_t *mv_cache,
                             const int8_t *ref_cache, int cbp, const uint8_t *nnzc, void *mb, int qmul_cb, int qmul_cr)
{
    H264SliceContext *sl = d->sl;
    int r;
    set_mb(d, mb_x, mb_y, mb_type, cbp, nnzc, mb);
    for (int i = 0; i < 4; i++)
        sl->sub_mb_type[i] = sub_mb_type[i];
    memcpy(sl->mv_cache, mv_cache, sizeof(sl->mv_cache));
    memcpy(sl->ref_cache, ref_cache, sizeof(sl->ref_cache));
    /* (4:2:2: the chroma DC quantiser sits three steps up, h264_mb_template.c:232-236) */
    d->pps->dequant4_buffer[4][sl->chroma_qp[0] + (d->cfmt == 2 ? 3 : 0)][0] = qmul_cb;
    d->pps->dequant4_buffer[5][sl->chroma_qp[1] + (d->cfmt == 2 ? 3 : 0)][0] = qmul_cr;
    r = run_hl_decode_mb(d);
    memcpy(mb, sl->mb, (sizeof(int16_t) << d->h->pixel_shift) * 3 * 256);
    return r;
}

/* An INTRA macroblock through the same object (arguments as ffref_h264_hl_decode_intra_mb_bd). */
int FN(h264dec_decode_intra)(FFRefH264Dec *d, int mb_x, int mb_y, int type, int intra16x16_pred_mode, int chroma_pred_mode,
                             const uint8_t *intra4x4_pred_mode, unsigned topleft_samples_available, unsigned topright_samples_available,
                             const uint8_t *nnzc, int cbp, void *mb, const void *mb_luma_dc, const int *qmul, const uint8_t *intra_pcm_ptr)
{
    H264SliceContext *sl = d->sl;
    const int ps = d->h->pixel_shift;
    int r;
    set_mb(d, mb_x, mb_y, type == 0 ? MB_TYPE_INTRA16x16 : type == 1 ? MB_TYPE_INTRA4x4 : type == 2 ? (MB_TYPE_INTRA4x4 | MB_TYPE_8x8DCT)
                                                                                         : MB_TYPE_INTRA_PCM, cbp, nnzc, mb);
    sl->intra16x16_pred_mode = intra16x16_pred_mode;
    sl->chroma_pred_mode = chroma_pred_mode;
    sl->topleft_samples_available = topleft_samples_available;
    sl->topright_samples_available = topright_samples_available;
    sl->intra_pcm_ptr = intra_pcm_ptr;
    for (int i = 0; i < 16; i++)
        sl->intra4x4_pred_mode_cache[scan8[i]] = intra4x4_pred_mode[i];
    if (mb_luma_dc) /* 4:4:4: 3 x 16 dctcoef, plane by plane, into sl->mb_luma_dc[0..2] (each 16 * 2 int16 whatever the depth) */
        for (int p = 0; p < (d->cfmt == 3 ? 3 : 1); p++)
            memcpy(sl->mb_luma_dc[p], (const uint8_t *)mb_luma_dc + (sizeof(int16_t) << ps) * 16 * p, (sizeof(int16_t) << ps) * 16);
    d->pps->dequant4_buffer[0][sl->qscale][0] = qmul[0];
    d->pps->dequant4_buffer[1][sl->chroma_qp[0] + (d->cfmt == 2 ? 3 : 0)][0] = qmul[1];
    d->pps->dequant4_buffer[2][sl->chroma_qp[1] + (d->cfmt == 2 ? 3 : 0)][0] = qmul[2];
    r = run_hl_decode_mb(d);
    memcpy(mb, sl->mb, (sizeof(int16_t) << ps) * 3 * 256);
    return r;
}

/* ff_h264_filter_mb() (libavcodec/h264_loopfilter.c:716) on one macroblock, in raster order over a picture: the state is what
 * fill_filter_caches() (h264_slice.c:2313) leaves for it.  ints = { mb_type, left_type (0: no left neighbour to filter against), top_type,
 * qscale of this / the left / the top macroblock, cbp, list_count, slice_alpha_c0_offset, slice_beta_offset, chroma_qp[0], chroma_qp[1],
 * pps->cabac }; mv_cache [2][40][2] int16, caches = ref_cache [2][40] int8 followed by non_zero_count_cache [15 * 8] uint8 — the
 * macroblock's own entries AND the neighbours' (the cache's border row and column).  C build: filters the picture given to set_cur()
 * in place.  hip build: records the macroblock's edges. */
int FN(h264dec_filter_mb)(FFRefH264Dec *d, int mb_x, int mb_y, const int *ints, const int16_t *mv_cache, const uint8_t *caches)
{
    H264Context *h = d->h;
    H264SliceContext *sl = d->sl;
    const int fld = h->picture_structure != PICT_FRAME; /* a field picture: the row above is two rows up in the frame's numbering */
    const int mb_xy = mb_x + mb_y * h->mb_stride, top_xy = mb_xy - (h->mb_stride << fld), ps = h->pixel_shift;
    sl->mb_x = mb_x;
    sl->mb_y = mb_y;
    sl->mb_xy = mb_xy;
    h->cur_pic.mb_type[mb_xy] = ints[0];
    sl->left_type[LTOP] = sl->left_type[LBOT] = ints[1];
    sl->top_type = ints[2];
    sl->left_mb_xy[LTOP] = sl->left_mb_xy[LBOT] = mb_xy - 1;
    sl->top_mb_xy = top_xy;
    h->cur_pic.qscale_table[mb_xy] = (int8_t)ints[3];
    if (mb_x > 0) {
        h->cur_pic.mb_type[mb_xy - 1] = ints[1];
        h->cur_pic.qscale_table[mb_xy - 1] = (int8_t)ints[4];
    }
    if (mb_y > fld) {
        h->cur_pic.mb_type[top_xy] = ints[2];
        h->cur_pic.qscale_table[top_xy] = (int8_t)ints[5];
    }
    sl->cbp = ints[6];
    sl->list_count = ints[7];
    sl->slice_alpha_c0_offset = ints[8];
    sl->slice_beta_offset = ints[9];
    sl->chroma_qp[0] = ints[10];
    sl->chroma_qp[1] = ints[11];
    d->pps->cabac = ints[12];
    memcpy(sl->mv_cache, mv_cache, sizeof(sl->mv_cache));
    memcpy(sl->ref_cache, caches, sizeof(sl->ref_cache));
    memcpy(sl->non_zero_count_cache, caches + sizeof(sl->ref_cache), 15 * 8);
#ifdef FFREF_WITH_HIP
    if (d->record)
        return ff_h264_hip_filter_mb(&d->rec, h, sl, mb_x, mb_y);
#endif
    {
        /* loop_filter() (h264_slice.c:2470-2491): a field macroblock on an odd row starts one line below the row pair's first */
        const int cs = d->cfmt == 3 ? 16 : 8, ch = d->cfmt <= 1 ? 8 : 16; /* block_h = 16 >> chroma_y_shift */
        uint8_t *dy = d->f->data[0] + (((ptrdiff_t)mb_x << ps) + (ptrdiff_t)mb_y * sl->linesize) * 16;
        uint8_t *dcb = d->f->data[1] + ((ptrdiff_t)mb_x << ps) * cs + (ptrdiff_t)mb_y * sl->uvlinesize * ch;
        uint8_t *dcr = d->f->data[2] + ((ptrdiff_t)mb_x << ps) * cs + (ptrdiff_t)mb_y * sl->uvlinesize * ch;
        if (fld && (mb_y & 1)) {
            dy -= (ptrdiff_t)sl->linesize * 15;
            dcb -= (ptrdiff_t)sl->uvlinesize * (ch - 1);
            dcr -= (ptrdiff_t)sl->uvlinesize * (ch - 1);
        }
        sl->mb_linesize = sl->linesize << fld;
        sl->mb_uvlinesize = sl->uvlinesize << fld;
        ff_h264_filter_mb(h, sl, mb_x, mb_y, dy, dcb, dcr, sl->linesize << fld, sl->uvlinesize << fld);
    }
    return 0;
}

#ifndef FFREF_WITH_HIP
/* h->vdsp.emulated_edge_mc at the depth (libavcodec/videodsp_template.c:24): what the decoder's edge buffer holds */
void ffref_emulated_edge_mc(int bit_depth, uint8_t *buf, const uint8_t *src, ptrdiff_t buf_linesize, ptrdiff_t src_linesize, int block_w,
                            int block_h, int src_x, int src_y, int w, int h)
{
    VideoDSPContext v;
    ff_videodsp_init(&v, bit_depth);
    v.emulated_edge_mc(buf, src, buf_linesize, src_linesize, block_w, block_h, src_x, src_y, w, h);
}
#endif
