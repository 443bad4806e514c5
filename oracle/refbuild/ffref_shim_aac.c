/*
 * ffref_shim_aac.c — flat accessor onto the reference's AACDecDSP.imdct_and_windowing (float decoder).  TEST INFRASTRUCTURE ONLY.
 * The member is a static function of libavcodec/aac/aacdec_float.c (through aacdec_dsp_template.c); it is reached the way the
 * decoder reaches it: ff_aac_decode_init_float() fills AACDecContext.dsp and the window tables, the two inverse MDCTs are
 * created as ff_aac_decode_init() creates them (aacdec.c:1267-1285).  The rest of the decoder (bitstream, SBR) is stubbed out.
 * Includes the reference's headers where they lie (-I/root/reference); contains no reference code.
 */
#include "config.h"
#include <string.h>
#include "libavutil/cpu.h"
#include "libavutil/log.h"
#include "libavutil/mem.h"
#include "libavutil/tx.h"
#include "libavcodec/avcodec.h"
#include "libavcodec/aac/aacdec.h"
#include "libavcodec/aactab.h"
#include "libavcodec/sinewin.h"

int ff_aac_decode_init_float(AVCodecContext *avctx);
/* the decoder proper is not part of this build: ff_aac_decode_init_float() ends in it */
int ff_aac_decode_init(AVCodecContext *avctx) { return 0; }
void ff_ps_init_common(void) {} /* likewise parametric stereo's tables (aacdec_tab.c:778) */
void ff_aac_sbr_init(void) {} /* init_tables_float_fn() calls it; SBR is not part of this build */

static AACDecContext *aac(void)
{
    static AACDecContext *ac;
    if (!ac) {
        av_force_cpu_flags(0);
        av_log_set_level(AV_LOG_ERROR);
        AVCodecContext *avctx = av_mallocz(sizeof(*avctx));
        ac = av_mallocz(sizeof(*ac));
        avctx->priv_data = ac;
        ac->avctx = avctx;
        if (ff_aac_decode_init_float(avctx) < 0)
            return NULL;
        float s128 = (1.0 / 128) / 32768.0f, s1024 = (1.0 / 1024) / 32768.0f; /* MDCT_INIT's scale_float */
        if (av_tx_init(&ac->mdct128, &ac->mdct128_fn, AV_TX_FLOAT_MDCT, 1, 128, &s128, 0) < 0 ||
            av_tx_init(&ac->mdct1024, &ac->mdct1024_fn, AV_TX_FLOAT_MDCT, 1, 1024, &s1024, 0) < 0)
            return NULL;
    }
    return ac;
}

/* which: 0 sine_1024, 1 sine_128, 2 kbd_long_1024, 3 kbd_short_128 */
const float *ffref_aac_window(int which)
{
    if (!aac())
        return NULL;
    return which == 0 ? ff_sine_1024 : which == 1 ? ff_sine_128 : which == 2 ? ff_aac_kbd_long_1024 : ff_aac_kbd_short_128;
}

/* one channel, one frame: seq / kb = {current, previous}; saved[512] is the overlap state in and out */
int ffref_aac_imdct_and_windowing(const float *coeffs, const int seq[2], const int kb[2], float *saved, float *out)
{
    AACDecContext *ac = aac();
    static SingleChannelElement *sce;
    if (!ac)
        return -1;
    if (!sce)
        sce = av_mallocz(sizeof(*sce));
    sce->ics.window_sequence[0] = seq[0]; sce->ics.window_sequence[1] = seq[1];
    sce->ics.use_kb_window[0] = kb[0];    sce->ics.use_kb_window[1] = kb[1];
    memcpy(sce->coeffs, coeffs, 1024 * sizeof(float));
    memcpy(sce->saved, saved, 512 * sizeof(float));
    sce->output = sce->ret_buf;
    ac->dsp.imdct_and_windowing(ac, sce);
    memcpy(out, sce->output, 1024 * sizeof(float));
    memcpy(saved, sce->saved, 512 * sizeof(float));
    return 0;
}

/* AACDecDSP.apply_tns on one channel-frame (coef[1024] in place); the arrays are TemporalNoiseShaping's / IndividualChannelStream's */
int ffref_aac_apply_tns(float *coef, const int n_filt[8], const int length[8][4], const int direction[8][4], const int order[8][4],
                        const float tcoef[8][4][20], int num_windows, int num_swb, const uint16_t *swb_offset, int tns_max_bands, int max_sfb,
                        int decode)
{
    AACDecContext *ac = aac();
    static TemporalNoiseShaping tns;
    static IndividualChannelStream ics;
    if (!ac)
        return -1;
    tns.present = 1;
    memcpy(tns.n_filt, n_filt, sizeof(tns.n_filt));
    memcpy(tns.length, length, sizeof(tns.length));
    memcpy(tns.direction, direction, sizeof(tns.direction));
    memcpy(tns.order, order, sizeof(tns.order));
    memcpy(tns.coef, tcoef, sizeof(tns.coef));
    ics.num_windows = num_windows;
    ics.num_swb = num_swb;
    ics.swb_offset = swb_offset;
    ics.tns_max_bands = tns_max_bands;
    ics.max_sfb = max_sfb;
    ac->dsp.apply_tns(coef, &tns, &ics, decode);
    return 0;
}

/* ---- the stereo tools and long-term prediction (AACDecDSP.apply_mid_side_stereo / apply_intensity_stereo / apply_ltp /
 *      update_ltp): the members run on a ChannelElement filled in from flat arguments ---- */
static ChannelElement *cpe_scratch(void)
{
    static ChannelElement *cpe;
    if (!cpe)
        cpe = av_mallocz(sizeof(*cpe));
    return cpe;
}

static void ics_groups(IndividualChannelStream *ics, int num_window_groups, const uint8_t *group_len, int max_sfb, const uint16_t *swb_offset)
{
    ics->num_window_groups = num_window_groups;
    memcpy(ics->group_len, group_len, num_window_groups);
    ics->max_sfb = max_sfb;
    ics->swb_offset = swb_offset;
}

int ffref_aac_apply_mid_side_stereo(float *ch0, float *ch1, int num_window_groups, const uint8_t *group_len, int max_sfb_ste,
                                    const uint8_t *ms_mask, const int *band_type0, const int *band_type1, const uint16_t *swb_offset)
{
    AACDecContext *ac = aac();
    ChannelElement *cpe = cpe_scratch();
    if (!ac || !cpe)
        return -1;
    ics_groups(&cpe->ch[0].ics, num_window_groups, group_len, max_sfb_ste, swb_offset);
    cpe->max_sfb_ste = max_sfb_ste;
    memcpy(cpe->ms_mask, ms_mask, 128);
    for (int i = 0; i < 128; i++) {
        cpe->ch[0].band_type[i] = band_type0[i];
        cpe->ch[1].band_type[i] = band_type1[i];
    }
    memcpy(cpe->ch[0].coeffs, ch0, 1024 * sizeof(float));
    memcpy(cpe->ch[1].coeffs, ch1, 1024 * sizeof(float));
    ac->dsp.apply_mid_side_stereo(ac, cpe);
    memcpy(ch0, cpe->ch[0].coeffs, 1024 * sizeof(float));
    memcpy(ch1, cpe->ch[1].coeffs, 1024 * sizeof(float));
    return 0;
}

int ffref_aac_apply_intensity_stereo(const float *coef0, float *coef1, int num_window_groups, const uint8_t *group_len, int max_sfb,
                                     int ms_present, const uint8_t *ms_mask, const int *band_type1, const float *sf1,
                                     const uint16_t *swb_offset)
{
    AACDecContext *ac = aac();
    ChannelElement *cpe = cpe_scratch();
    if (!ac || !cpe)
        return -1;
    ics_groups(&cpe->ch[1].ics, num_window_groups, group_len, max_sfb, swb_offset);
    memcpy(cpe->ms_mask, ms_mask, 128);
    for (int i = 0; i < 128; i++)
        cpe->ch[1].band_type[i] = band_type1[i];
    memcpy(cpe->ch[1].sf, sf1, 128 * sizeof(float));
    memcpy(cpe->ch[0].coeffs, coef0, 1024 * sizeof(float));
    memcpy(cpe->ch[1].coeffs, coef1, 1024 * sizeof(float));
    ac->dsp.apply_intensity_stereo(ac, cpe, ms_present);
    memcpy(coef1, cpe->ch[1].coeffs, 1024 * sizeof(float));
    return 0;
}

/* tns_present = 0: no TNS on the prediction; otherwise the TemporalNoiseShaping arrays as in ffref_aac_apply_tns */
int ffref_aac_apply_ltp(float *coeffs, const float *ltp_state, int lag, float coef, const int8_t *used, const int seq[2], const int kb[2],
                        int max_sfb, int num_swb, int tns_max_bands, const uint16_t *swb_offset, int tns_present, const int n_filt[8],
                        const int length[8][4], const int direction[8][4], const int order[8][4], const float tcoef[8][4][20], float *predFreq)
{
    AACDecContext *ac = aac();
    SingleChannelElement *sce = &cpe_scratch()->ch[0];
    if (!ac)
        return -1;
    if (!ac->mdct_ltp) {
        float s = -32786.0 * 2 + 36; /* ff_aac_decode_init's scale_float for the LTP transform (aacdec.c:1288-1291) */
        if (av_tx_init(&ac->mdct_ltp, &ac->mdct_ltp_fn, AV_TX_FLOAT_MDCT, 0, 1024, &s, 0) < 0)
            return -1;
    }
    memset(&sce->ics, 0, sizeof(sce->ics));
    sce->ics.window_sequence[0] = seq[0]; sce->ics.window_sequence[1] = seq[1];
    sce->ics.use_kb_window[0] = kb[0];    sce->ics.use_kb_window[1] = kb[1];
    sce->ics.max_sfb = max_sfb;
    sce->ics.num_swb = num_swb;
    sce->ics.num_windows = 1;
    sce->ics.tns_max_bands = tns_max_bands;
    sce->ics.swb_offset = swb_offset;
    sce->ics.ltp.present = 1;
    sce->ics.ltp.lag = lag;
    sce->ics.ltp.coef = coef;
    memcpy(sce->ics.ltp.used, used, MAX_LTP_LONG_SFB);
    sce->tns.present = tns_present;
    if (tns_present) {
        memcpy(sce->tns.n_filt, n_filt, sizeof(sce->tns.n_filt));
        memcpy(sce->tns.length, length, sizeof(sce->tns.length));
        memcpy(sce->tns.direction, direction, sizeof(sce->tns.direction));
        memcpy(sce->tns.order, order, sizeof(sce->tns.order));
        memcpy(sce->tns.coef, tcoef, sizeof(sce->tns.coef));
    }
    memcpy(sce->coeffs, coeffs, 1024 * sizeof(float));
    memcpy(sce->ltp_state, ltp_state, 3072 * sizeof(float));
    sce->output = sce->ret_buf;
    ac->dsp.apply_ltp(ac, sce);
    memcpy(coeffs, sce->coeffs, 1024 * sizeof(float));
    memcpy(predFreq, ac->buf_mdct, 1024 * sizeof(float));
    return 0;
}

int ffref_aac_update_ltp(float *ltp_state, const float *buf_mdct, const float *saved, const float *output, int seq0, int kb0)
{
    AACDecContext *ac = aac();
    SingleChannelElement *sce = &cpe_scratch()->ch[0];
    if (!ac)
        return -1;
    sce->ics.window_sequence[0] = seq0;
    sce->ics.use_kb_window[0] = kb0;
    memcpy(ac->buf_mdct, buf_mdct, 1024 * sizeof(float));
    memcpy(sce->saved, saved, 512 * sizeof(float));
    sce->output = sce->ret_buf;
    memcpy(sce->output, output, 1024 * sizeof(float));
    memcpy(sce->ltp_state, ltp_state, 3072 * sizeof(float));
    ac->dsp.update_ltp(ac, sce);
    memcpy(ltp_state, sce->ltp_state, 3072 * sizeof(float));
    return 0;
}

/* ---- the 960- and 768-sample frame variants (AACDecDSP.imdct_and_windowing_960 / _768).  Their window tables are file-static in
 *      aacdec_float.c; the same public generators fill copies here (ff_sine_window_init / ff_kbd_window_init, as
 *      init_tables_float_fn does for 960 / 120, aacdec_float.c:63-67), so what the caller is handed is bit for bit what the member
 *      uses ---- */
#include "libavcodec/kbdwin.h"
const float *ffref_aac_window_len(int L, int which)
{
    static float tab[4][1024], zero[1024];
    static int done;
    if (L == 768) /* this reference never fills sine_768 / sine_96 / aac_kbd_long_768 / aac_kbd_short_96 (file-static, no
                   * ff_*_window_init call anywhere in libavcodec): the member runs on all-zero tables, and so must its checker */
        return zero;
    if (L != 960)
        return NULL;
    if (!done) {
        ff_sine_window_init(tab[0], 960);
        ff_sine_window_init(tab[1], 120);
        ff_kbd_window_init(tab[2], 4.0, 960);
        ff_kbd_window_init(tab[3], 6.0, 120);
        done = 1;
    }
    return tab[which & 3];
}

int ffref_aac_imdct_and_windowing_len(int L, const float *coeffs, const int seq[2], const int kb[2], float *saved, float *out)
{
    AACDecContext *ac = aac();
    SingleChannelElement *sce = &cpe_scratch()->ch[0];
    if (!ac || (L != 960 && L != 768))
        return -1;
    if (!ac->mdct960) {
        float s96 = (1.0 / 96) / 32768.0f, s120 = (1.0 / 120) / 32768.0f, s768 = (1.0 / 768) / 32768.0f, s960 = (1.0 / 960) / 32768.0f;
        if (av_tx_init(&ac->mdct96, &ac->mdct96_fn, AV_TX_FLOAT_MDCT, 1, 96, &s96, 0) < 0 ||
            av_tx_init(&ac->mdct120, &ac->mdct120_fn, AV_TX_FLOAT_MDCT, 1, 120, &s120, 0) < 0 ||
            av_tx_init(&ac->mdct768, &ac->mdct768_fn, AV_TX_FLOAT_MDCT, 1, 768, &s768, 0) < 0 ||
            av_tx_init(&ac->mdct960, &ac->mdct960_fn, AV_TX_FLOAT_MDCT, 1, 960, &s960, 0) < 0)
            return -1;
    }
    sce->ics.window_sequence[0] = seq[0]; sce->ics.window_sequence[1] = seq[1];
    sce->ics.use_kb_window[0] = kb[0];    sce->ics.use_kb_window[1] = kb[1];
    memcpy(sce->coeffs, coeffs, 1024 * sizeof(float));
    memcpy(sce->saved, saved, (L / 2) * sizeof(float));
    sce->output = sce->ret_buf;
    if (L == 960)
        ac->dsp.imdct_and_windowing_960(ac, sce);
    else
        ac->dsp.imdct_and_windowing_768(ac, sce);
    memcpy(out, sce->output, L * sizeof(float));
    memcpy(saved, sce->saved, (L / 2) * sizeof(float));
    return 0;
}

/* ---- AAC-LD / AAC-ELD: AACDecDSP.imdct_and_windowing_ld / _eld ---- */
/* which: 0 ff_sine_512, 1 ff_sine_128, 2 ff_aac_eld_window_512, 3 ff_aac_eld_window_480 */
const float *ffref_aac_ld_table(int which)
{
    if (!aac())
        return NULL;
    return which == 0 ? ff_sine_512 : which == 1 ? ff_sine_128 : which == 2 ? ff_aac_eld_window_512 : ff_aac_eld_window_480;
}

static int ld_tx(AACDecContext *ac)
{
    if (!ac->mdct512) {
        float s480 = (1.0 / 480) / 32768.0f, s512 = (1.0 / 512) / 32768.0f;
        if (av_tx_init(&ac->mdct480, &ac->mdct480_fn, AV_TX_FLOAT_MDCT, 1, 480, &s480, 0) < 0 ||
            av_tx_init(&ac->mdct512, &ac->mdct512_fn, AV_TX_FLOAT_MDCT, 1, 512, &s512, 0) < 0)
            return -1;
    }
    return 0;
}

int ffref_aac_imdct_and_windowing_ld(const float *coeffs, int kb_prev, float *saved, float *out)
{
    AACDecContext *ac = aac();
    SingleChannelElement *sce = &cpe_scratch()->ch[0];
    if (!ac || ld_tx(ac) < 0)
        return -1;
    sce->ics.use_kb_window[1] = kb_prev;
    memcpy(sce->coeffs, coeffs, 1024 * sizeof(float));
    memcpy(sce->saved, saved, 256 * sizeof(float));
    sce->output = sce->ret_buf;
    ac->dsp.imdct_and_windowing_ld(ac, sce);
    memcpy(out, sce->output, 512 * sizeof(float));
    memcpy(saved, sce->saved, 256 * sizeof(float));
    return 0;
}

int ffref_aac_imdct_and_windowing_eld(int n, const float *coeffs, float *saved, float *out)
{
    AACDecContext *ac = aac();
    SingleChannelElement *sce = &cpe_scratch()->ch[0];
    if (!ac || ld_tx(ac) < 0 || (n != 512 && n != 480))
        return -1;
    ac->oc[1].m4ac.frame_length_short = n == 480;
    memcpy(sce->coeffs, coeffs, 1024 * sizeof(float));
    memcpy(sce->saved, saved, 3 * n * sizeof(float));
    sce->output = sce->ret_buf;
    ac->dsp.imdct_and_windowing_eld(ac, sce);
    memcpy(out, sce->output, n * sizeof(float));
    memcpy(saved, sce->saved, 3 * n * sizeof(float));
    return 0;
}

/* ---- AAC Main prediction and channel coupling: AACDecDSP.apply_prediction / apply_dependent_coupling / apply_independent_coupling ---- */
/* sampling_index: pred_sfb_max = ff_aac_pred_sfb_max[sampling_index] is matched by the caller choosing the index */
int ffref_aac_apply_prediction(float *ps, float *coef, int is_long, int *initialized, int predictor_present, const uint8_t *prediction_used,
                               int sampling_index, const uint16_t *swb_offset, int reset_group)
{
    AACDecContext *ac = aac();
    SingleChannelElement *sce = &cpe_scratch()->ch[0];
    static float *st; /* MAX_PREDICTORS PredictorState records of 8 floats (libavcodec/aac_defines.h:130-139; the type itself needs USE_FIXED) */
    if (!ac)
        return -1;
    if (!st)
        st = av_mallocz(MAX_PREDICTORS * 8 * sizeof(float));
    memcpy(st, ps, MAX_PREDICTORS * 8 * sizeof(float));
    sce->predictor_state = (struct PredictorState *)st;
    ac->oc[1].m4ac.sampling_index = sampling_index;
    sce->ics.window_sequence[0] = is_long ? ONLY_LONG_SEQUENCE : EIGHT_SHORT_SEQUENCE;
    sce->ics.predictor_initialized = *initialized;
    sce->ics.predictor_present = predictor_present;
    sce->ics.predictor_reset_group = reset_group;
    memcpy(sce->ics.prediction_used, prediction_used, 41);
    sce->ics.swb_offset = swb_offset;
    memcpy(sce->coeffs, coef, 1024 * sizeof(float));
    ac->dsp.apply_prediction(ac, sce);
    memcpy(coef, sce->coeffs, 1024 * sizeof(float));
    memcpy(ps, st, MAX_PREDICTORS * 8 * sizeof(float));
    *initialized = sce->ics.predictor_initialized;
    return ff_aac_pred_sfb_max[sampling_index];
}

int ffref_aac_apply_dependent_coupling(float *dest, const float *src, int num_window_groups, const uint8_t *group_len, int max_sfb,
                                       const int *band_type, const float *gain, const uint16_t *swb_offset)
{
    AACDecContext *ac = aac();
    static ChannelElement *cce;
    SingleChannelElement *target = &cpe_scratch()->ch[0];
    if (!ac)
        return -1;
    if (!cce)
        cce = av_mallocz(sizeof(*cce));
    ac->oc[1].m4ac.object_type = AOT_AAC_MAIN;
    ics_groups(&cce->ch[0].ics, num_window_groups, group_len, max_sfb, swb_offset);
    for (int i = 0; i < 128; i++)
        cce->ch[0].band_type[i] = band_type[i];
    memcpy(cce->coup.gain[3], gain, 120 * sizeof(float));
    memcpy(cce->ch[0].coeffs, src, 1024 * sizeof(float));
    memcpy(target->coeffs, dest, 1024 * sizeof(float));
    ac->dsp.apply_dependent_coupling(ac, target, cce, 3);
    memcpy(dest, target->coeffs, 1024 * sizeof(float));
    return 0;
}

int ffref_aac_apply_independent_coupling(float *dest, const float *src, float gain, int len)
{
    AACDecContext *ac = aac();
    static ChannelElement *cce;
    SingleChannelElement *target = &cpe_scratch()->ch[0];
    if (!ac || (len != 1024 && len != 2048))
        return -1;
    if (!cce)
        cce = av_mallocz(sizeof(*cce));
    ac->oc[1].m4ac.sbr = len == 2048;
    cce->coup.gain[5][0] = gain;
    cce->ch[0].output = cce->ch[0].ret_buf;
    target->output = target->ret_buf;
    memcpy(cce->ch[0].output, src, len * sizeof(float));
    memcpy(target->output, dest, len * sizeof(float));
    ac->dsp.apply_independent_coupling(ac, target, cce, 5);
    memcpy(dest, target->output, len * sizeof(float));
    ac->oc[1].m4ac.sbr = 0;
    return 0;
}
