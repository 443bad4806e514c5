/*
 * ffref_shim_h264pred.c — flat accessors onto the reference's H264PredContext.  TEST INFRASTRUCTURE ONLY.
 * Its own translation unit: h264pred.h's mode macros (DC_PRED, ...) collide with vp9.h's enum of the same names.
 * Includes the reference's headers where they lie (-I/root/reference); contains no reference code.
 */
#include "config.h"
#include <stddef.h>
#include <stdint.h>
#include "libavutil/cpu.h"
#include "libavutil/log.h"
#include "libavcodec/codec_id.h"
#include <string.h>
#include "libavcodec/h264pred.h"

static void pure_c(void) { av_force_cpu_flags(0); av_log_set_level(AV_LOG_ERROR); }

/* ---- h264pred: H264PredContext of the H.264 codec, 4:2:0, at the depth ffref_h264_pred_set_bit_depth() chose (8 to start with) ---- */
static H264PredContext pred_ctx;
static int pred_ready;
static H264PredContext *h264pred(void)
{
    pure_c();
    if (!pred_ready) {
        ff_h264_pred_init(&pred_ctx, AV_CODEC_ID_H264, 8, 1);
        pred_ready = 1;
    }
    return &pred_ctx;
}
/* 8 / 9 / 10 / 12 / 14: the instantiations of h264pred_template.c (libavcodec/h264pred.c:448-538 per depth); samples are uint16_t and
 * the _add members' coefficients int32_t above 8 bits */
void ffref_h264_pred_set_bit_depth(int bit_depth)
{
    pure_c();
    ff_h264_pred_init(&pred_ctx, AV_CODEC_ID_H264, bit_depth, 1);
    pred_ready = 1;
}
/* the same with chroma_format_idc given: from 2 on pred8x8[] / pred8x8_add[] are the 8 x 16 forms (h264pred.c:478-535) */
void ffref_h264_pred_set_format(int bit_depth, int chroma_format_idc)
{
    pure_c();
    ff_h264_pred_init(&pred_ctx, AV_CODEC_ID_H264, bit_depth, chroma_format_idc);
    pred_ready = 1;
}
/* the table of another codec that shares H264PredContext (AV_CODEC_ID_SVQ3 / _RV40 / _VP7 / _VP8: h264pred.c:540-578), 8 bits, 4:2:0;
 * members the reference leaves unset for the codec are NULL (the context is cleared first): ffref_h264_pred_has() tells */
void ffref_h264_pred_set_codec(int codec_id)
{
    pure_c();
    memset(&pred_ctx, 0, sizeof(pred_ctx));
    ff_h264_pred_init(&pred_ctx, codec_id, 8, 1);
    pred_ready = 1;
}
int ffref_h264_pred_has(int table, int mode)
{
    const H264PredContext *c = h264pred();
    return table == 0 ? c->pred4x4[mode] != NULL : table == 2 ? c->pred8x8[mode] != NULL : table == 3 ? c->pred16x16[mode] != NULL : c->pred8x8l[mode] != NULL;
}
void ffref_h264_pred4x4(int mode, uint8_t *src, const uint8_t *topright, ptrdiff_t stride) { h264pred()->pred4x4[mode](src, topright, stride); }
void ffref_h264_pred8x8l(int mode, uint8_t *src, int has_topleft, int has_topright, ptrdiff_t stride)
{
    h264pred()->pred8x8l[mode](src, has_topleft, has_topright, stride);
}
void ffref_h264_pred8x8(int mode, uint8_t *src, ptrdiff_t stride) { h264pred()->pred8x8[mode](src, stride); }
void ffref_h264_pred16x16(int mode, uint8_t *src, ptrdiff_t stride) { h264pred()->pred16x16[mode](src, stride); }
void ffref_h264_pred4x4_add(int mode, uint8_t *pix, int16_t *block, ptrdiff_t stride) { h264pred()->pred4x4_add[mode](pix, block, stride); }
void ffref_h264_pred8x8l_add(int mode, uint8_t *pix, int16_t *block, ptrdiff_t stride) { h264pred()->pred8x8l_add[mode](pix, block, stride); }
void ffref_h264_pred8x8l_filter_add(int mode, uint8_t *pix, int16_t *block, int has_topleft, int has_topright, ptrdiff_t stride)
{
    h264pred()->pred8x8l_filter_add[mode](pix, block, has_topleft, has_topright, stride);
}
void ffref_h264_pred8x8_add(int mode, uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    h264pred()->pred8x8_add[mode](pix, block_offset, block, stride);
}
void ffref_h264_pred16x16_add(int mode, uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    h264pred()->pred16x16_add[mode](pix, block_offset, block, stride);
}
