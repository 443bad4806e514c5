/*
 * ffo_aac.c — CPU restatement of AACDecDSP.imdct_and_windowing, float AAC-LC / 1024-sample frames
 * (libavcodec/aac/aacdec_dsp_template.c:325-387) and of the window tables it reads (libavcodec/sinewin_tablegen.h:57-64,
 * libavcodec/kbdwin.c:29-56).  TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg are the
 * only callers.  Pinned bit-exact to the reference's own member built in place (tests/test_oracle_vs_ref.py::test_aac_*); the
 * Kaiser-Bessel table is pinned to <= 1 ulp only (the reference evaluates I0 with its own rational approximation,
 * libavutil/mathematics.c av_bessel_i0, this file with the power series), so parity tests take the tables as inputs from
 * tests/golden/aac.npz — as the hip path takes them from its caller.
 *
 * Restated as what it computes rather than as the call sequence: the frame's 1024 "time" samples buf[] (one long or eight short
 * inverse MDCTs), then out[] = the overlap of the previous frame's tail `saved` with buf's head under the previous frame's window
 * shape, and the new tail, which depends on buf[] alone.
 */
#include <math.h>
#include <string.h>
#include "ffo.h"

enum { ONLY_LONG, LONG_START, EIGHT_SHORT, LONG_STOP }; /* enum WindowSequence, libavcodec/aac.h:63-68 */

void ffo_aac_sine_window(float *w, int n)
{
    for (int i = 0; i < n; i++)
        w[i] = sinf((i + 0.5) * (M_PI / (2.0 * n)));
}

static double bessel_i0(double x)
{
    double q = x * x / 4, term = 1, sum = 1;
    for (int k = 1; k < 500 && term > sum * 1e-18; k++) {
        term *= q / ((double)k * k);
        sum += term;
    }
    return sum;
}

void ffo_aac_kbd_window(float *w, float alpha, int n)
{
    double t[513], sum = 0, scale = 0;
    const double a2 = 4 * (alpha * M_PI / n) * (alpha * M_PI / n);
    for (int i = 0; i <= n / 2; i++) {
        t[i] = bessel_i0(sqrt(i * (double)(n - i) * a2));
        scale += t[i] * (1 + (i && i < n / 2));
    }
    scale = 1.0 / (scale + 1);
    for (int i = 0; i < n; i++) {
        sum += t[i <= n / 2 ? i : n - i];
        w[i] = (float)sqrt(sum * scale);
    }
}

/* AVFloatDSPContext.vector_fmul_window (libavutil/float_dsp.c:79-97): 2 len outputs from len + len inputs under a 2 len window */
static void window_overlap(float *dst, const float *src0, const float *src1, const float *win, int len)
{
    for (int t = 0; t < len; t++) {
        const float a = src0[t], b = src1[len - 1 - t], wi = win[t], wj = win[2 * len - 1 - t];
        dst[t] = a * wj - b * wi;
        dst[2 * len - 1 - t] = a * wi + b * wj;
    }
}

/* windows[]: sine_<L>, sine_<L/8>, kbd_long_<L>, kbd_short_<L/8>; seq / kb = { this frame, previous frame }.
 * L = 1024: imdct_and_windowing (aacdec_dsp_template.c:325-387); 960: _960 (:453-512; a short window's coefficients sit 128 apart:
 * in_short_stride = 128); 768: _768 (:389-448; 96 apart).  saved holds L / 2, out L samples. */
void ffo_aac_imdct_and_windowing_len(int L, int in_short_stride, const FfoTx *mdct_long, const FfoTx *mdct_short, const float *const windows[4],
                                     const float *coeffs, const int seq[2], const int kb[2], float *saved, float *out)
{
    const int H = L / 2, S = L / 8, S2 = S / 2, A = H - S2; /* 1024: 512, 128, 64, 448 */
    const float *swindow = windows[kb[0] ? 3 : 1], *lwindow_prev = windows[kb[1] ? 2 : 0], *swindow_prev = windows[kb[1] ? 3 : 1];
    float buf[1024], tail[128];
    if (seq[0] == EIGHT_SHORT)
        for (int i = 0; i < 8; i++)
            ffo_mdct_run(mdct_short, buf + i * S, coeffs + i * in_short_stride, sizeof(float));
    else
        ffo_mdct_run(mdct_long, buf, coeffs, sizeof(float));

    const int long_prev = seq[1] == ONLY_LONG || seq[1] == LONG_STOP, long_cur = seq[0] == ONLY_LONG || seq[0] == LONG_START;
    if (long_prev && long_cur) {
        window_overlap(out, saved, buf, lwindow_prev, H);
    } else {
        memcpy(out, saved, A * sizeof(float));
        if (seq[0] == EIGHT_SHORT) {
            window_overlap(out + A, saved + A, buf, swindow_prev, S2);
            for (int b = 1; b < 4; b++)
                window_overlap(out + A + b * S, buf + (b - 1) * S + S2, buf + b * S, swindow, S2);
            window_overlap(tail, buf + 3 * S + S2, buf + 4 * S, swindow, S2);
            memcpy(out + A + 4 * S, tail, S2 * sizeof(float));
        } else {
            window_overlap(out + A, saved + A, buf, swindow_prev, S2);
            memcpy(out + A + S, buf + S2, A * sizeof(float));
        }
    }
    if (seq[0] == EIGHT_SHORT) {
        memcpy(saved, tail + S2, S2 * sizeof(float));
        for (int b = 4; b < 7; b++)
            window_overlap(saved + S2 + (b - 4) * S, buf + b * S + S2, buf + (b + 1) * S, swindow, S2);
        memcpy(saved + A, buf + 7 * S + S2, S2 * sizeof(float));
    } else if (seq[0] == LONG_START) {
        memcpy(saved, buf + H, A * sizeof(float));
        memcpy(saved + A, buf + 7 * S + S2, S2 * sizeof(float));
    } else {
        memcpy(saved, buf + H, H * sizeof(float));
    }
}

void ffo_aac_imdct_and_windowing(const FfoTx *mdct1024, const FfoTx *mdct128, const float *const windows[4], const float *coeffs,
                                 const int seq[2], const int kb[2], float *saved, float *out)
{
    ffo_aac_imdct_and_windowing_len(1024, 128, mdct1024, mdct128, windows, coeffs, seq, kb, saved, out);
}

/*
 * AACDecDSP.apply_tns, float (libavcodec/aac/aacdec_dsp_template.c:164-223; LPC from the transmitted reflection coefficients:
 * compute_lpc_coefs, libavcodec/lpc_functions.h:54-103 with normalize = 0).  Restated in two steps: the walk over windows and
 * filters that yields each filter's range (a "filter record"), and the filter itself on one record.
 */
int ffo_aac_tns_filters(FfoAacTnsFilter *out, const int n_filt[8], const int length[8][4], const int direction[8][4], const int order[8][4],
                        const float coef[8][4][20], int num_windows, int num_swb, const uint16_t *swb_offset, int tns_max_bands, int max_sfb)
{
    const int mmm = tns_max_bands < max_sfb ? tns_max_bands : max_sfb;
    int n = 0;
    if (!mmm)
        return 0;
    for (int w = 0; w < num_windows; w++) {
        int bottom = num_swb;
        for (int f = 0; f < n_filt[w]; f++) {
            const int top = bottom;
            bottom = top - length[w][f] > 0 ? top - length[w][f] : 0;
            if (!order[w][f])
                continue;
            const int start = swb_offset[bottom < mmm ? bottom : mmm], end = swb_offset[top < mmm ? top : mmm];
            if (end - start <= 0)
                continue;
            FfoAacTnsFilter *r = &out[n++];
            r->size = end - start;
            r->inc = direction[w][f] ? -1 : 1;
            r->start = (direction[w][f] ? end - 1 : start) + w * 128;
            r->order = order[w][f];
            memcpy(r->coef, coef[w][f], sizeof(r->coef));
        }
    }
    return n;
}

void ffo_aac_tns_run(float *coef, const FfoAacTnsFilter *r, int decode)
{
    float lpc[20], hist[21] = { 0 };
    for (int i = 0; i < r->order; i++) { /* the step-up recursion, in place */
        const float k = -r->coef[i];
        lpc[i] = k;
        for (int j = 0; j < (i + 1) >> 1; j++) {
            const float f = lpc[j], b = lpc[i - 1 - j];
            lpc[j] = f + k * b;
            lpc[i - 1 - j] = b + k * f;
        }
    }
    int p = r->start;
    for (int m = 0; m < r->size; m++, p += r->inc) {
        const int lim = m < r->order ? m : r->order;
        float x = coef[p];
        hist[0] = x; /* MA: the inputs' history; AR: overwritten with the output below */
        for (int i = 1; i <= lim; i++)
            x = decode ? x - hist[i] * lpc[i - 1] : x + hist[i] * lpc[i - 1];
        coef[p] = x;
        if (decode)
            hist[0] = x;
        for (int i = r->order; i > 0; i--)
            hist[i] = hist[i - 1];
    }
}

/*
 * The stereo tools and long-term prediction of AACDecDSP, float (test infrastructure like the rest of this file).
 */

/* AACDecDSP.apply_mid_side_stereo (aacdec_dsp_template.c:83-111); the band loop body is AVFloatDSPContext.butterflies_float
 * (libavutil/float_dsp.c:112-122).  band_type*: enum BandType per (group, sfb) (libavcodec/aac.h:66-78: NOISE_BT = 13) */
void ffo_aac_apply_mid_side_stereo(float *ch0, float *ch1, int num_window_groups, const uint8_t *group_len, int max_sfb_ste,
                                   const uint8_t *ms_mask, const int *band_type0, const int *band_type1, const uint16_t *swb_offset)
{
    for (int g = 0; g < num_window_groups; g++) {
        for (int sfb = 0; sfb < max_sfb_ste; sfb++) {
            const int idx = g * max_sfb_ste + sfb;
            if (!ms_mask[idx] || band_type0[idx] >= 13 || band_type1[idx] >= 13)
                continue;
            for (int w = 0; w < group_len[g]; w++)
                for (int i = swb_offset[sfb]; i < swb_offset[sfb + 1]; i++) {
                    float *a = ch0 + w * 128 + i, *b = ch1 + w * 128 + i;
                    const float t = *a - *b;
                    *a += *b;
                    *b = t;
                }
        }
        ch0 += group_len[g] * 128;
        ch1 += group_len[g] * 128;
    }
}

/* AACDecDSP.apply_intensity_stereo (aacdec_dsp_template.c:120-160): INTENSITY_BT2 = 14 (out of phase), INTENSITY_BT = 15; the band
 * loop body is vector_fmul_scalar (libavutil/float_dsp.c:45-51) */
void ffo_aac_apply_intensity_stereo(const float *coef0, float *coef1, int num_window_groups, const uint8_t *group_len, int max_sfb,
                                    int ms_present, const uint8_t *ms_mask, const int *band_type1, const float *sf1, const uint16_t *swb_offset)
{
    for (int g = 0; g < num_window_groups; g++) {
        for (int sfb = 0; sfb < max_sfb; sfb++) {
            const int idx = g * max_sfb + sfb;
            if (band_type1[idx] != 14 && band_type1[idx] != 15)
                continue;
            int c = -1 + 2 * (band_type1[idx] - 14);
            if (ms_present)
                c *= 1 - 2 * ms_mask[idx];
            const float scale = c * sf1[idx];
            for (int w = 0; w < group_len[g]; w++)
                for (int i = swb_offset[sfb]; i < swb_offset[sfb + 1]; i++)
                    coef1[w * 128 + i] = coef0[w * 128 + i] * scale;
        }
        coef0 += group_len[g] * 128;
        coef1 += group_len[g] * 128;
    }
}

/* AACDecDSP.apply_ltp (aacdec_dsp_template.c:252-282) with windowing_and_mdct_ltp (:225-247) inlined; mdct_ltp: the forward
 * 1024-point MDCT of ff_aac_decode_init (aacdec.c:1288-1291); tns: the channel's filter records (ffo_aac_tns_filters), ntns = 0
 * when sce->tns.present is 0; predFreq[1024] is handed back for inspection */
void ffo_aac_apply_ltp(const FfoTx *mdct_ltp, const float *const windows[4], float *coeffs, const float *ltp_state, int lag, float coef,
                       const int8_t *used, const int seq[2], const int kb[2], int max_sfb, const uint16_t *swb_offset,
                       const FfoAacTnsFilter *tns, int ntns, float *predFreq)
{
    if (seq[0] == EIGHT_SHORT)
        return;
    const float *lwindow = windows[kb[0] ? 2 : 0], *swindow = windows[kb[0] ? 3 : 1];
    const float *lwindow_prev = windows[kb[1] ? 2 : 0], *swindow_prev = windows[kb[1] ? 3 : 1];
    float in[2048];
    const int num_samples = lag < 1024 ? lag + 1024 : 2048;
    for (int i = 0; i < 2048; i++)
        in[i] = i < num_samples ? ltp_state[i + 2048 - lag] * coef : 0.0f;
    if (seq[0] != LONG_STOP) {
        for (int i = 0; i < 1024; i++)
            in[i] = in[i] * lwindow_prev[i];
    } else {
        memset(in, 0, 448 * sizeof(float));
        for (int i = 0; i < 128; i++)
            in[448 + i] = in[448 + i] * swindow_prev[i];
    }
    if (seq[0] != LONG_START) {
        for (int i = 0; i < 1024; i++)
            in[1024 + i] = in[1024 + i] * lwindow[1023 - i];
    } else {
        for (int i = 0; i < 128; i++)
            in[1024 + 448 + i] = in[1024 + 448 + i] * swindow[127 - i];
        memset(in + 1024 + 576, 0, 448 * sizeof(float));
    }
    ffo_mdct_run(mdct_ltp, predFreq, in, sizeof(float));
    for (int k = 0; k < ntns; k++)
        ffo_aac_tns_run(predFreq, &tns[k], 0);
    for (int sfb = 0; sfb < (max_sfb < 40 ? max_sfb : 40); sfb++)
        if (used[sfb])
            for (int i = swb_offset[sfb]; i < swb_offset[sfb + 1]; i++)
                coeffs[i] += predFreq[i];
}

/* AACDecDSP.update_ltp (aacdec_dsp_template.c:287-320): buf_mdct = the frame's inverse-MDCT output (ac->buf_mdct after
 * imdct_and_windowing), saved / output = the channel's overlap state and samples after it; ltp_state[3072] in and out */
void ffo_aac_update_ltp(const float *const windows[4], float *ltp_state, const float *buf_mdct, const float *saved, const float *output,
                        int seq0, int kb0)
{
    const float *lwindow = windows[kb0 ? 2 : 0], *swindow = windows[kb0 ? 3 : 1];
    float saved_ltp[1024];
    if (seq0 == EIGHT_SHORT || seq0 == LONG_START) {
        memcpy(saved_ltp, seq0 == EIGHT_SHORT ? saved : buf_mdct + 512, (seq0 == EIGHT_SHORT ? 512 : 448) * sizeof(float));
        memset(saved_ltp + 576, 0, 448 * sizeof(float));
        for (int i = 0; i < 64; i++)
            saved_ltp[448 + i] = buf_mdct[960 + i] * swindow[64 + 63 - i];
        for (int i = 0; i < 64; i++)
            saved_ltp[512 + i] = buf_mdct[1023 - i] * swindow[63 - i];
    } else {
        for (int i = 0; i < 512; i++)
            saved_ltp[i] = buf_mdct[512 + i] * lwindow[512 + 511 - i];
        for (int i = 0; i < 512; i++)
            saved_ltp[512 + i] = buf_mdct[1023 - i] * lwindow[511 - i];
    }
    memmove(ltp_state, ltp_state + 1024, 1024 * sizeof(float));
    memcpy(ltp_state + 1024, output, 1024 * sizeof(float));
    memcpy(ltp_state + 2048, saved_ltp, 1024 * sizeof(float));
}

/* AACDecDSP.imdct_and_windowing_ld (aacdec_dsp_template.c:516-541): one 512-point inverse MDCT per frame, the overlap under
 * sine_512 or — when the PREVIOUS frame asked for it — the low-overlap window (sine_128 in the middle); saved holds 256, out 512 */
void ffo_aac_imdct_and_windowing_ld(const FfoTx *mdct512, const float *sine_512, const float *sine_128, const float *coeffs, int kb_prev,
                                    float *saved, float *out)
{
    float buf[512];
    ffo_mdct_run(mdct512, buf, coeffs, sizeof(float));
    if (kb_prev) {
        memcpy(out, saved, 192 * sizeof(float));
        window_overlap(out + 192, saved + 192, buf, sine_128, 64);
        memcpy(out + 320, buf + 64, 192 * sizeof(float));
    } else {
        window_overlap(out, saved, buf, sine_512, 256);
    }
    memcpy(saved, buf + 256, 256 * sizeof(float));
}

/* AACDecDSP.imdct_and_windowing_eld (aacdec_dsp_template.c:543-602), n = 512 or 480: the coefficient shuffle that maps the ELD
 * filterbank onto a conventional inverse MDCT, sign flips, four-term window sums over the history of three frames (saved[3 n],
 * newest first); window = ff_aac_eld_window_512 / _480.  The reference shuffles sce->coeffs in place; this takes a copy. */
void ffo_aac_imdct_and_windowing_eld(int n, const FfoTx *mdct, const float *window, const float *coeffs, float *saved, float *out)
{
    const int n2 = n >> 1, n4 = n >> 2;
    float in[512], buf[512];
    memcpy(in, coeffs, n * sizeof(float));
    for (int i = 0; i < n2; i += 2) {
        float t = in[i];
        in[i] = -in[n - 1 - i];
        in[n - 1 - i] = t;
        t = -in[i + 1];
        in[i + 1] = in[n - 2 - i];
        in[n - 2 - i] = t;
    }
    ffo_mdct_run(mdct, buf, in, sizeof(float));
    for (int i = 0; i < n; i += 2)
        buf[i] = -buf[i];
    for (int i = n4; i < n2; i++)
        out[i - n4] = buf[n2 - 1 - i] * window[i - n4] + saved[i + n2] * window[i + n - n4] + -saved[n + n2 - 1 - i] * window[i + 2 * n - n4] +
                      -saved[2 * n + n2 + i] * window[i + 3 * n - n4];
    for (int i = 0; i < n2; i++)
        out[n4 + i] = buf[i] * window[i + n2 - n4] + -saved[n - 1 - i] * window[i + n2 + n - n4] + -saved[n + i] * window[i + n2 + 2 * n - n4] +
                      saved[2 * n + n - 1 - i] * window[i + n2 + 3 * n - n4];
    for (int i = 0; i < n4; i++)
        out[n2 + n4 + i] = buf[i + n2] * window[i + n - n4] + -saved[n2 - 1 - i] * window[i + 2 * n - n4] + -saved[n + n2 + i] * window[i + 3 * n - n4];
    memmove(saved + n, saved, 2 * n * sizeof(float));
    memcpy(saved, buf, n * sizeof(float));
}

/* AACDecDSP.apply_prediction, float (AAC Main; aacdec_dsp_template.c:636-664 with predict(), aacdec_float_prediction.h:35-85).
 * ps: 672 PredictorState records of 8 floats (cor0, cor1, var0, var1, r0, r1, k1, x_est: aac_defines.h:130-139) */
static float f16_round(float pf) { uint32_t i; memcpy(&i, &pf, 4); i = (i + 0x00008000U) & 0xFFFF0000U; memcpy(&pf, &i, 4); return pf; }
static float f16_even(float pf)
{
    uint32_t i;
    memcpy(&i, &pf, 4);
    i = (i + 0x00007FFFU + (i & (0x00010000U >> 16))) & 0xFFFF0000U; /* the reference's precedence: i & 1 */
    memcpy(&pf, &i, 4);
    return pf;
}
static float f16_trunc(float pf) { uint32_t i; memcpy(&i, &pf, 4); i &= 0xFFFF0000U; memcpy(&pf, &i, 4); return pf; }

static void pred_reset(float *ps)
{
    ps[0] = ps[1] = 0.0f;
    ps[2] = ps[3] = 1.0f;
    ps[4] = ps[5] = 0.0f;
}

void ffo_aac_apply_prediction(float *ps, float *coef, int is_long, int *initialized, int predictor_present, const uint8_t *prediction_used,
                              int pred_sfb_max, const uint16_t *swb_offset, int reset_group)
{
    if (!*initialized) {
        for (int i = 0; i < 672; i++)
            pred_reset(ps + 8 * i);
        *initialized = 1;
    }
    if (!is_long) {
        for (int i = 0; i < 672; i++)
            pred_reset(ps + 8 * i);
        return;
    }
    const float a = 0.953125f, alpha = 0.90625f;
    for (int sfb = 0; sfb < pred_sfb_max; sfb++)
        for (int k = swb_offset[sfb]; k < swb_offset[sfb + 1]; k++) {
            float *p = ps + 8 * k;
            const float r0 = p[4], r1 = p[5], cor0 = p[0], cor1 = p[1], var0 = p[2], var1 = p[3];
            const float k1 = var0 > 1 ? cor0 * f16_even(a / var0) : 0;
            const float k2 = var1 > 1 ? cor1 * f16_even(a / var1) : 0;
            const float pv = f16_round(k1 * r0 + k2 * r1);
            if (predictor_present && prediction_used[sfb])
                coef[k] += pv;
            const float e0 = coef[k], e1 = e0 - k1 * r0;
            p[1] = f16_trunc(alpha * cor1 + r1 * e1);
            p[3] = f16_trunc(alpha * var1 + 0.5f * (r1 * r1 + e1 * e1));
            p[0] = f16_trunc(alpha * cor0 + r0 * e0);
            p[2] = f16_trunc(alpha * var0 + 0.5f * (r0 * r0 + e0 * e0));
            p[5] = f16_trunc(a * (r0 - k1 * e0));
            p[4] = f16_trunc(a * e0);
        }
    if (reset_group)
        for (int i = reset_group - 1; i < 672; i += 30)
            pred_reset(ps + 8 * i);
}

/* AACDecDSP.apply_dependent_coupling (aacdec_float_coupling.h:42-71; band_type 0 = ZERO_BT) and apply_independent_coupling (:78-88) */
void ffo_aac_apply_dependent_coupling(float *dest, const float *src, int num_window_groups, const uint8_t *group_len, int max_sfb,
                                      const int *band_type, const float *gain, const uint16_t *swb_offset)
{
    int idx = 0;
    for (int g = 0; g < num_window_groups; g++) {
        for (int i = 0; i < max_sfb; i++, idx++)
            if (band_type[idx] != 0)
                for (int w = 0; w < group_len[g]; w++)
                    for (int k = swb_offset[i]; k < swb_offset[i + 1]; k++)
                        dest[w * 128 + k] += gain[idx] * src[w * 128 + k];
        dest += group_len[g] * 128;
        src += group_len[g] * 128;
    }
}

void ffo_aac_apply_independent_coupling(float *dest, const float *src, float gain, int len)
{
    for (int i = 0; i < len; i++)
        dest[i] += src[i] * gain;
}
