/*
 * integration/avcodec_h264_hip.c — libavcodec/hip/h264{dsp,qpel,chroma,pred}_init.c of the FFmpeg-side patch: the `hip` arch hooks of
 * the four H.264 DSP tables.
 *
 * The reference ends each table's init with one `#if ARCH_X86 ... #elif ARCH_AARCH64 ...` chain (libavcodec/h264dsp.c:155-169,
 * h264qpel.c:105-119, h264chroma.c:54-66, h264pred.c:591-603).  A GPU arch is additional to the host ISA, so the patch is one more
 * call after the chain.  This build leaves the reference files untouched: they are compiled where they lie with their init renamed
 * (-Dff_h264dsp_init=ff_h264dsp_init_c, ...) and the functions below take the original names — call the reference, then the hook.
 *
 * libffhip's tables (include/ffhip.h) hold the members it replaces with the reference's exact signatures, so a hook is a
 * member-by-member copy in both directions: in (the C functions become libffhip's fallbacks), out (the hip faces).
 */
#include <string.h>

#include "libavutil/attributes.h"
#include "libavutil/cpu.h"
#include "libavcodec/codec_id.h"
#include "libavcodec/h264chroma.h"
#include "libavcodec/h264dsp.h"
#include "libavcodec/h264pred.h"
#include "libavcodec/h264qpel.h"

#include "ffhip.h"
#include "hip_cpu.h"

void ff_h264dsp_init_c(H264DSPContext *c, const int bit_depth, const int chroma_format_idc);
void ff_h264qpel_init_c(H264QpelContext *c, int bit_depth);
void ff_h264chroma_init_c(H264ChromaContext *c, int bit_depth);
void ff_h264_pred_init_c(H264PredContext *h, int codec_id, const int bit_depth, const int chroma_format_idc);

#define H264DSP_MEMBERS(X) X(v_loop_filter_luma) X(h_loop_filter_luma) X(v_loop_filter_luma_intra) X(h_loop_filter_luma_intra) \
    X(v_loop_filter_chroma) X(h_loop_filter_chroma) X(v_loop_filter_chroma_intra) X(h_loop_filter_chroma_intra) \
    X(idct_add) X(idct8_add) X(idct_dc_add) X(idct8_dc_add) X(idct_add16) X(idct8_add4) X(idct_add16intra) X(idct_add8) \
    X(luma_dc_dequant_idct) X(chroma_dc_dequant_idct) X(add_pixels8_clear) X(add_pixels4_clear) \
    X(h_loop_filter_luma_mbaff) X(h_loop_filter_luma_mbaff_intra) X(h_loop_filter_chroma_mbaff) X(h_loop_filter_chroma_mbaff_intra)

static av_cold void h264dsp_init_hip(H264DSPContext *c, const int bit_depth, const int chroma_format_idc)
{
    FFHipH264DSPContext h;
    FFHipH264WeightContext w;
#define GIVE(m) h.m = c->m;
#define TAKE(m) c->m = h.m;
    H264DSP_MEMBERS(GIVE)                       /* what the C / SIMD inits left: libffhip keeps these as its fallbacks */
    if (ff_h264dsp_init_hip(&h, bit_depth, chroma_format_idc) >= 0) {
        H264DSP_MEMBERS(TAKE)
    }                                           /* else: 4:2:2 / 4:4:4, a depth it does not take, no device — keep the pointers */
    for (int i = 0; i < 4; i++) {
        w.weight_pixels_tab[i]   = c->weight_pixels_tab[i];
        w.biweight_pixels_tab[i] = c->biweight_pixels_tab[i];
    }
    if (ff_h264dsp_weight_init_hip(&w, bit_depth) >= 0)
        for (int i = 0; i < 4; i++) {
            c->weight_pixels_tab[i]   = w.weight_pixels_tab[i];
            c->biweight_pixels_tab[i] = w.biweight_pixels_tab[i];
        }
}

av_cold void ff_h264dsp_init(H264DSPContext *c, const int bit_depth, const int chroma_format_idc)
{
    ff_h264dsp_init_c(c, bit_depth, chroma_format_idc);
    if (av_get_cpu_flags() & AV_CPU_FLAG_HIP)
        h264dsp_init_hip(c, bit_depth, chroma_format_idc);
}

/* put_/avg_h264_qpel_pixels_tab: [4][16] in the reference (h264qpel.h:27-30), sizes 16 / 8 / 4 (/ 2) — libffhip takes the first three */
av_cold void ff_h264qpel_init(H264QpelContext *c, int bit_depth)
{
    FFHipH264QpelContext h;
    ff_h264qpel_init_c(c, bit_depth);
    if (!(av_get_cpu_flags() & AV_CPU_FLAG_HIP))
        return;
    for (int s = 0; s < 3; s++)
        for (int i = 0; i < 16; i++) {
            h.put_h264_qpel_pixels_tab[s][i] = c->put_h264_qpel_pixels_tab[s][i];
            h.avg_h264_qpel_pixels_tab[s][i] = c->avg_h264_qpel_pixels_tab[s][i];
        }
    if (ff_h264qpel_init_hip(&h, bit_depth) < 0)
        return;
    for (int s = 0; s < 3; s++)
        for (int i = 0; i < 16; i++) {
            c->put_h264_qpel_pixels_tab[s][i] = h.put_h264_qpel_pixels_tab[s][i];
            c->avg_h264_qpel_pixels_tab[s][i] = h.avg_h264_qpel_pixels_tab[s][i];
        }
}

av_cold void ff_h264chroma_init(H264ChromaContext *c, int bit_depth)
{
    FFHipH264ChromaContext h;
    ff_h264chroma_init_c(c, bit_depth);
    if (!(av_get_cpu_flags() & AV_CPU_FLAG_HIP))
        return;
    for (int i = 0; i < 4; i++) {
        h.put_h264_chroma_pixels_tab[i] = c->put_h264_chroma_pixels_tab[i];
        h.avg_h264_chroma_pixels_tab[i] = c->avg_h264_chroma_pixels_tab[i];
    }
    if (ff_h264chroma_init_hip(&h, bit_depth) < 0)
        return;
    for (int i = 0; i < 4; i++) {
        c->put_h264_chroma_pixels_tab[i] = h.put_h264_chroma_pixels_tab[i];
        c->avg_h264_chroma_pixels_tab[i] = h.avg_h264_chroma_pixels_tab[i];
    }
}

/* H264PredContext and FFHipH264PredContext declare the same members in the same order with the same signatures (h264pred.h:92-116) */
_Static_assert(sizeof(FFHipH264PredContext) == sizeof(H264PredContext), "FFHipH264PredContext mirrors H264PredContext");
av_cold void ff_h264_pred_init(H264PredContext *h, int codec_id, const int bit_depth, const int chroma_format_idc)
{
    FFHipH264PredContext x;
    ff_h264_pred_init_c(h, codec_id, bit_depth, chroma_format_idc);
    if (!(av_get_cpu_flags() & AV_CPU_FLAG_HIP))
        return;
    memcpy(&x, h, sizeof(x));
    if (ff_h264_pred_init_hip(&x, codec_id, bit_depth, chroma_format_idc) >= 0)
        memcpy(h, &x, sizeof(x));
}
