/*
 * ffo_tx.c — CPU restatement of the reference's float MDCT (av_tx, AV_TX_FLOAT_MDCT, power-of-two).
 * TEST INFRASTRUCTURE ONLY (see ffo.h).
 *
 *   ff_tx_mdct_init / _fwd / _inv     libavutil/tx_template.c:1223-1342
 *   ff_tx_mdct_gen_exp                libavutil/tx_template.c:2107-2134
 *   split-radix FFT codelets          libavutil/tx_template.c:540-722 (BUTTERFLIES/TRANSFORM, sr_combine)
 *   cosine tables                     libavutil/tx_template.c:65-77
 *   input permutation                 libavutil/tx.c:125-155
 *
 * The reference unrolls the split-radix recursion into fft2/4/8/16 base cases and an 8-way
 * combine loop; the recursion below performs the same per-element float operations in the same
 * expression order (multiplications by the exact table values 1 and 0 included), so — built
 * without FMA contraction — it is expected to match the reference C bit for bit; the test-suite
 * checks that and the stated tolerance separately.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ffo.h"

typedef struct { float re, im; } cpx;

struct FfoTx {
    int   len;      /* av_tx len: forward output count */
    int   inv;
    int  *map;      /* len/2 entries */
    cpx  *exp;      /* len/2 (fwd) or len (inv) entries */
    float *cos_tab[20]; /* cos_tab[log2 n][k] = cos(2*pi*k/n), k <= n/4 */
};

static int sr_perm(int i, int len, int inv)
{
    len >>= 1;
    if (len <= 1)
        return i & 1;
    if (!(i & len))
        return sr_perm(i, len, inv) * 2;
    len >>= 1;
    return sr_perm(i, len, inv) * 4 + 1 - 2 * (!(i & len) ^ inv);
}

static void sr_fft(const struct FfoTx *s, cpx *z, int n, int lg)
{
    if (n == 1)
        return;
    if (n == 2) {
        cpx d = { z[0].re - z[1].re, z[0].im - z[1].im };
        z[0].re = z[0].re + z[1].re;
        z[0].im = z[0].im + z[1].im;
        z[1] = d;
        return;
    }
    const int q = n >> 2;
    sr_fft(s, z, n >> 1, lg - 1);
    sr_fft(s, z + 2 * q, q, lg - 2);
    sr_fft(s, z + 3 * q, q, lg - 2);
    const float *tab = s->cos_tab[lg];
    for (int k = 0; k < q; k++) {
        cpx *a0 = z + k, *a1 = z + k + q, *a2 = z + k + 2 * q, *a3 = z + k + 3 * q;
        float wre = tab[k], wim = tab[q - k], nwim = -wim;
        float t1 = a2->re * wre - a2->im * nwim;
        float t2 = a2->re * nwim + a2->im * wre;
        float t5 = a3->re * wre - a3->im * wim;
        float t6 = a3->re * wim + a3->im * wre;
        float r0 = a0->re, i0 = a0->im, r1 = a1->re, i1 = a1->im;
        float t3 = t5 - t1;
        t5 = t5 + t1;
        a2->re = r0 - t5;
        a0->re = r0 + t5;
        a3->im = i1 - t3;
        a1->im = i1 + t3;
        float t4 = t2 - t6;
        t6 = t2 + t6;
        a3->re = r1 - t4;
        a1->re = r1 + t4;
        a2->im = i0 - t6;
        a0->im = i0 + t6;
    }
}

FfoTx *ffo_mdct_create(int inv, int len, float scale_f)
{
    if (len < 4 || (len & (len - 1)))
        return NULL;
    struct FfoTx *s = calloc(1, sizeof(*s));
    const int n = len >> 1; /* complex FFT size */
    double scale = scale_f;
    int lg = 0;
    while ((1 << lg) < n)
        lg++;
    s->len = len;
    s->inv = inv;
    s->map = malloc(sizeof(int) * n);
    s->exp = malloc(sizeof(cpx) * (inv ? 2 * n : n));
    for (int l = 2; l <= lg; l++) {
        int m = 1 << l;
        double freq = 2 * M_PI / m;
        s->cos_tab[l] = malloc(sizeof(float) * (m / 4 + 1));
        for (int i = 0; i < m / 4; i++)
            s->cos_tab[l][i] = (float)cos(i * freq);
        s->cos_tab[l][m / 4] = 0;
    }
    /* forward MDCT asks for a SCATTER map, inverse for GATHER (tx_template.c:1231-1233) */
    for (int i = 0; i < n; i++) {
        int p = -sr_perm(i, n, inv) & (n - 1);
        if (!inv)
            s->map[p] = i;
        else
            s->map[i] = p;
    }
    {
        const double theta = (scale < 0 ? n : 0) + 1.0 / 8.0;
        const double sc = sqrt(fabs(scale));
        cpx *e = s->exp + (inv ? n : 0);
        for (int i = 0; i < n; i++) {
            const double alpha = M_PI_2 * (i + theta) / n;
            e[i].re = (float)(cos(alpha) * sc);
            e[i].im = (float)(sin(alpha) * sc);
        }
        if (inv)
            for (int i = 0; i < n; i++)
                s->exp[i] = s->exp[n + s->map[i]];
    }
    return s;
}

void ffo_mdct_free(FfoTx *s)
{
    if (!s)
        return;
    for (int l = 0; l < 20; l++)
        free(s->cos_tab[l]);
    free(s->map);
    free(s->exp);
    free(s);
}

/* stride in bytes, as av_tx_fn; forward: output stride, inverse: input stride */
void ffo_mdct_run(const FfoTx *s, float *out, const float *in, ptrdiff_t stride)
{
    const int n = s->len >> 1, q = s->len >> 2, len3 = 3 * n;
    const cpx *exp = s->exp;
    int lg = 0;
    while ((1 << lg) < n)
        lg++;
    stride /= (ptrdiff_t)sizeof(float);
    cpx *z = malloc(sizeof(cpx) * n);
    if (!s->inv) {
        for (int i = 0; i < n; i++) {
            const int k = 2 * i;
            float re, im;
            if (k < n) {
                re = -in[n + k] + in[n - 1 - k];
                im = -in[len3 + k] + -in[len3 - 1 - k];
            } else {
                re = -in[n + k] + -in[5 * n - 1 - k];
                im = in[k - n] + -in[len3 - 1 - k];
            }
            cpx *d = z + s->map[i];
            d->im = re * exp[i].re - im * exp[i].im;
            d->re = re * exp[i].im + im * exp[i].re;
        }
        sr_fft(s, z, n, lg);
        for (int i = 0; i < q; i++) {
            const int i0 = q + i, i1 = q - i - 1;
            cpx s1 = z[i1], s0 = z[i0];
            out[(2 * i1 + 1) * stride] = s0.re * exp[i0].im - s0.im * exp[i0].re;
            out[2 * i0 * stride]       = s0.re * exp[i0].re + s0.im * exp[i0].im;
            out[(2 * i0 + 1) * stride] = s1.re * exp[i1].im - s1.im * exp[i1].re;
            out[2 * i1 * stride]       = s1.re * exp[i1].re + s1.im * exp[i1].im;
        }
    } else {
        const float *in2 = in + (2 * n - 1) * stride;
        cpx *o = (cpx *)out;
        for (int i = 0; i < n; i++) {
            const int k = s->map[i] << 1;
            float tre = in2[-k * stride], tim = in[k * stride];
            z[i].re = tre * exp[i].re - tim * exp[i].im;
            z[i].im = tre * exp[i].im + tim * exp[i].re;
        }
        sr_fft(s, z, n, lg);
        exp += n;
        for (int i = 0; i < q; i++) {
            const int i0 = q + i, i1 = q - i - 1;
            cpx s1 = { z[i1].im, z[i1].re }, s0 = { z[i0].im, z[i0].re };
            o[i1].re = s1.re * exp[i1].im - s1.im * exp[i1].re;
            o[i0].im = s1.re * exp[i1].re + s1.im * exp[i1].im;
            o[i0].re = s0.re * exp[i0].im - s0.im * exp[i0].re;
            o[i1].im = s0.re * exp[i0].re + s0.im * exp[i0].im;
        }
    }
    free(z);
}

/*
 * AV_TX_FLOAT_FFT, power-of-two: the out-of-place wrapper gathers the input through the split-radix permutation
 * (ff_tx_fft, libavutil/tx_template.c:735-749; map from ff_tx_gen_ptwo_revtab, tx.c:125-155) and runs the same
 * "no shuffle" split-radix network the MDCT uses; the inverse differs by the permutation only.  Complex interleaved
 * (re, im) floats in and out, len of each; unnormalised.
 */
void ffo_fft_run(int inv, int len, float *out, const float *in)
{
    struct FfoTx s;
    memset(&s, 0, sizeof(s));
    int lg = 0;
    while ((1 << lg) < len)
        lg++;
    for (int l = 2; l <= lg; l++) {
        const int m = 1 << l;
        const double freq = 2 * M_PI / m;
        s.cos_tab[l] = malloc(sizeof(float) * (m / 4 + 1));
        for (int i = 0; i < m / 4; i++)
            s.cos_tab[l][i] = (float)cos(i * freq);
        s.cos_tab[l][m / 4] = 0;
    }
    cpx *z = malloc(sizeof(cpx) * len);
    const cpx *src = (const cpx *)in;
    for (int i = 0; i < len; i++)
        z[i] = src[-sr_perm(i, len, inv) & (len - 1)];
    sr_fft(&s, z, len, lg);
    memcpy(out, z, sizeof(cpx) * len);
    free(z);
    for (int l = 0; l < 20; l++)
        free(s.cos_tab[l]);
}

/* ff_tx_mdct_naive_fwd: tx_template.c:1144-1163 — in 2*len, out len (double results) */
void ffo_mdct_naive_fwd(int len, double scale, double *out, const float *in)
{
    const double phase = M_PI / (4.0 * len);
    for (int i = 0; i < len; i++) {
        double sum = 0.0;
        for (int j = 0; j < 2 * len; j++) {
            int a = (2 * j + 1 + len) * (2 * i + 1);
            sum += in[j] * cos(a * phase);
        }
        out[i] = sum * scale;
    }
}

/* ff_tx_mdct_naive_inv: tx_template.c:1165-1193 — in len coefficients, out len samples */
void ffo_mdct_naive_inv(int len, double scale, double *out, const float *in)
{
    const int h = len >> 1;
    const double phase = M_PI / (4.0 * len);
    for (int i = 0; i < h; i++) {
        double sd = 0.0, su = 0.0;
        double id = phase * (4 * h - 2 * i - 1), iu = phase * (3 * len + 2 * i + 1);
        for (int j = 0; j < len; j++) {
            double a = 2 * j + 1;
            sd += cos(a * id) * in[j];
            su += cos(a * iu) * in[j];
        }
        out[i] = sd * scale;
        out[i + h] = -su * scale;
    }
}
