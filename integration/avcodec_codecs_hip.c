/*
 * integration/avcodec_codecs_hip.c — libavcodec/hip/{me_cmp,hevcdsp,vp9dsp}_init.c of the FFmpeg-side patch.
 *
 * Same recipe as avcodec_h264_hip.c: the reference's ff_me_cmp_init / ff_hevc_dsp_init / ff_vp9dsp_init are compiled where they
 * lie under the names *_c, the functions here take the original names, call them and then the `hip` hook — the one call a
 * maintainer adds behind the ARCH_* chains (libavcodec/me_cmp.c:1014-1026, hevc/dsp.c after the per-depth switch, vp9dsp.c:88-112).
 */
#include <string.h>

#include "libavutil/attributes.h"
#include "libavutil/cpu.h"
#include "libavcodec/avcodec.h"
#include "libavcodec/hevc/dsp.h"
#include "libavcodec/me_cmp.h"
#include "libavcodec/vp9dsp.h"

#include "ffhip.h"
#include "hip_cpu.h"

void ff_me_cmp_init_c(MECmpContext *c, AVCodecContext *avctx);
void ff_hevc_dsp_init_c(HEVCDSPContext *hpc, int bit_depth);
void ff_vp9dsp_init_c(VP9DSPContext *dsp, int bpp, int bitexact);

/* me_cmp_func's first argument is the encoder context, unused by these metrics (checkasm passes NULL, tests/checkasm/motion.c) */
av_cold void ff_me_cmp_init(MECmpContext *c, AVCodecContext *avctx)
{
    FFHipMECmpContext h;
    ff_me_cmp_init_c(c, avctx);
    if (!(av_get_cpu_flags() & AV_CPU_FLAG_HIP))
        return;
    for (int i = 0; i < 2; i++) {
        h.sad[i]            = (ffhip_me_cmp_func)c->sad[i];
        h.hadamard8_diff[i] = (ffhip_me_cmp_func)c->hadamard8_diff[i];
        h.pix_abs[i][0]     = (ffhip_me_cmp_func)c->pix_abs[i][0];
        for (int k = 0; k < 3; k++)
            h.pix_abs_hpel[i][k] = (ffhip_me_cmp_func)c->pix_abs[i][k + 1];
        h.sse[i]            = (ffhip_me_cmp_func)c->sse[i];
        h.nsse[i]           = (ffhip_me_cmp_func)c->nsse[i];
    }
    if (ff_me_cmp_init_hip(&h) < 0)
        return;
    for (int i = 0; i < 2; i++) {
        c->sad[i]            = (me_cmp_func)h.sad[i];
        c->hadamard8_diff[i] = (me_cmp_func)h.hadamard8_diff[i];
        c->pix_abs[i][0]     = (me_cmp_func)h.pix_abs[i][0];
        for (int k = 0; k < 3; k++)
            c->pix_abs[i][k + 1] = (me_cmp_func)h.pix_abs_hpel[i][k];
        c->sse[i]            = (me_cmp_func)h.sse[i];
        c->nsse[i]           = (me_cmp_func)h.nsse[i];
    }
}

/* SAOParams is what FFHipSAOParams restates (hevc/dsp.h:35-46) */
_Static_assert(sizeof(FFHipSAOParams) == sizeof(SAOParams), "FFHipSAOParams mirrors SAOParams");

av_cold void ff_hevc_dsp_init(HEVCDSPContext *c, int bit_depth)
{
    FFHipHEVCDSPContext h;
    ff_hevc_dsp_init_c(c, bit_depth);
    if (!(av_get_cpu_flags() & AV_CPU_FLAG_HIP))
        return;
#define SCALAR(X) X(transform_4x4_luma) X(hevc_h_loop_filter_luma) X(hevc_v_loop_filter_luma) X(hevc_h_loop_filter_chroma) \
    X(hevc_v_loop_filter_chroma) X(hevc_h_loop_filter_luma_c) X(hevc_v_loop_filter_luma_c) X(hevc_h_loop_filter_chroma_c) \
    X(hevc_v_loop_filter_chroma_c) X(dequant) X(transform_rdpcm)
#define TABLE(X) X(add_residual) X(idct) X(idct_dc) X(sao_band_filter) X(sao_edge_filter) X(put_hevc_qpel) X(put_hevc_qpel_uni) \
    X(put_hevc_epel) X(put_hevc_epel_uni) X(put_hevc_qpel_uni_w) X(put_hevc_qpel_bi) X(put_hevc_qpel_bi_w) X(put_hevc_epel_uni_w) \
    X(put_hevc_epel_bi) X(put_hevc_epel_bi_w)
#define GIVE_S(m) h.m = (void *)c->m;
#define TAKE_S(m) c->m = (void *)h.m;
#define SAME_T(m) _Static_assert(sizeof(h.m) == sizeof(c->m), "table " #m " has the reference's shape");
#define GIVE_T(m) memcpy(h.m, c->m, sizeof(h.m));
#define TAKE_T(m) memcpy(c->m, h.m, sizeof(h.m));
    TABLE(SAME_T)
    _Static_assert(sizeof(h.sao_edge_restore) == sizeof(c->sao_edge_restore), "sao_edge_restore[2]");
    SCALAR(GIVE_S) TABLE(GIVE_T)
    memcpy(h.sao_edge_restore, c->sao_edge_restore, sizeof(h.sao_edge_restore));
    if (ff_hevc_dsp_init_hip(&h, bit_depth) < 0)
        return;                                   /* 9 bits, above 12, no device: keep the C pointers; put_pcm stays C anyway */
    SCALAR(TAKE_S) TABLE(TAKE_T)
    memcpy(c->sao_edge_restore, h.sao_edge_restore, sizeof(h.sao_edge_restore));
}

av_cold void ff_vp9dsp_init(VP9DSPContext *dsp, int bpp, int bitexact)
{
    FFHipVP9ItxfmContext it;
    FFHipVP9McContext mc;
    FFHipVP9ScaledMcContext smc;
    FFHipVP9LoopFilterContext lf;
    FFHipVP9IntraContext ip;
    ff_vp9dsp_init_c(dsp, bpp, bitexact);
    if (!(av_get_cpu_flags() & AV_CPU_FLAG_HIP))
        return;
    _Static_assert(sizeof(it.itxfm_add) == sizeof(dsp->itxfm_add) && sizeof(mc.mc) == sizeof(dsp->mc) && sizeof(smc.smc) == sizeof(dsp->smc) &&
                   sizeof(lf.loop_filter_8) == sizeof(dsp->loop_filter_8) && sizeof(lf.loop_filter_16) == sizeof(dsp->loop_filter_16) &&
                   sizeof(lf.loop_filter_mix2) == sizeof(dsp->loop_filter_mix2) && sizeof(ip.intra_pred) == sizeof(dsp->intra_pred),
                   "the five vp9dsp tables have the reference's shapes (vp9dsp.h:36-120)");
    memcpy(it.itxfm_add, dsp->itxfm_add, sizeof(it.itxfm_add));
    if (ff_vp9dsp_itxfm_init_hip(&it, bpp) >= 0)
        memcpy(dsp->itxfm_add, it.itxfm_add, sizeof(it.itxfm_add));
    memcpy(mc.mc, dsp->mc, sizeof(mc.mc));
    if (ff_vp9dsp_mc_init_hip(&mc, bpp) >= 0)
        memcpy(dsp->mc, mc.mc, sizeof(mc.mc));
    memcpy(smc.smc, dsp->smc, sizeof(smc.smc));
    if (ff_vp9dsp_scaled_mc_init_hip(&smc, bpp) >= 0)
        memcpy(dsp->smc, smc.smc, sizeof(smc.smc));
    memcpy(lf.loop_filter_8, dsp->loop_filter_8, sizeof(lf.loop_filter_8));
    memcpy(lf.loop_filter_16, dsp->loop_filter_16, sizeof(lf.loop_filter_16));
    memcpy(lf.loop_filter_mix2, dsp->loop_filter_mix2, sizeof(lf.loop_filter_mix2));
    if (ff_vp9dsp_loopfilter_init_hip(&lf, bpp) >= 0) {
        memcpy(dsp->loop_filter_8, lf.loop_filter_8, sizeof(lf.loop_filter_8));
        memcpy(dsp->loop_filter_16, lf.loop_filter_16, sizeof(lf.loop_filter_16));
        memcpy(dsp->loop_filter_mix2, lf.loop_filter_mix2, sizeof(lf.loop_filter_mix2));
    }
    memcpy(ip.intra_pred, dsp->intra_pred, sizeof(ip.intra_pred));
    if (ff_vp9dsp_intrapred_init_hip(&ip, bpp) >= 0)
        memcpy(dsp->intra_pred, ip.intra_pred, sizeof(ip.intra_pred));
}
