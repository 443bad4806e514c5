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
    int  *map;      /* len/2 entries (15xM: in_map, then out_map: 2 * len/2) */
    cpx  *exp;      /* len/2 (fwd) or len (inv) entries */
    float *cos_tab[20]; /* cos_tab[log2 n][k] = cos(2*pi*k/n), k <= n/4 */
    int   pfa_m;    /* 0: power of two; else len/2 = pfa_f * pfa_m (ff_tx_mdct_pfa_<f>xM_*) */
    int   pfa_f;    /* the prime-factor codelet's small factor: 3, 5, 7, 9 or 15 */
    int  *sub_map;  /* NxM: the sub-transform's scatter map (pfa_m entries) */
    float tab53[12];/* ff_tx_tab_53 */
    float tab7[6], tab9[8]; /* ff_tx_tab_7, ff_tx_tab_9 */
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

static FfoTx *pfa_create(int inv, int len, int factor, float scale_f);

/* which ff_tx_mdct_pfa_<N>xM codelet av_tx_init ends up with for len/2 = N * 2^k (candidates DECL_COMP_MDCT(3 / 5 / 7 / 9 / 15),
 * libavutil/tx_template.c:1595-1599, ranked in libavutil/tx.c:get_codelet_prio "larger factors are generally better"): 0 = none */
int ffo_mdct_pfa_factor(int len)
{
    static const int f[5] = { 15, 9, 7, 5, 3 };
    for (int i = 0; i < 5; i++) {
        const int m = len / (2 * f[i]);
        if (len % (2 * f[i]) == 0 && m >= 2 && !(m & (m - 1)))
            return f[i];
    }
    return 0;
}

FfoTx *ffo_mdct_create(int inv, int len, float scale_f)
{
    if (ffo_mdct_pfa_factor(len))
        return pfa_create(inv, len, ffo_mdct_pfa_factor(len), scale_f);
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
    free(s->sub_map);
    free(s->exp);
    free(s);
}

static void pfa_run(const FfoTx *s, float *out, const float *in, ptrdiff_t stride);

/* AV_TX_FULL_IMDCT: ff_tx_mdct_inv_full (libavutil/tx_template.c:1391-1408): the half inverse lands in the middle of the
 * 2 * len outputs and is mirrored outwards (first quarter negated).  s must be an inverse context; contiguous data. */
void ffo_imdct_full_run(const FfoTx *s, float *out, const float *in)
{
    const int len = s->len << 1, len2 = len >> 1, len4 = len >> 2;
    ffo_mdct_run(s, out + len4, in, sizeof(float));
    for (int i = 0; i < len4; i++) {
        out[i] = -out[len2 - i - 1];
        out[len - i - 1] = out[len2 + i];
    }
}

/* stride in bytes, as av_tx_fn; forward: output stride, inverse: input stride */
void ffo_mdct_run(const FfoTx *s, float *out, const float *in, ptrdiff_t stride)
{
    if (s->pfa_m) {
        pfa_run(s, out, in, stride);
        return;
    }
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
 * MDCT lengths 2 * 15 * 2^k (Opus / CELT 120..960, AAC-960 240 and 1920): ff_tx_mdct_pfa_15xM_fwd / _inv, the codelet av_tx picks
 * for them ("larger factors are better", libavutil/tx.c:391-395).
 *
 *   ff_tx_mdct_pfa_init, DECL_COMP_IMDCT / DECL_COMP_MDCT   libavutil/tx_template.c:1425-1600
 *   fft3, fft5_m1..3, fft15, ff_tx_tab_53                   libavutil/tx_template.c:92-107,175-245,463-476
 *   ff_tx_gen_compound_mapping (GATHER), mulinv             libavutil/tx.c:34-42,75-121
 *   TX_EMBED_INPUT_PFA_MAP                                  libavutil/tx_priv.h:275-284
 *   sub-transform: fftM_ns, in place, SCATTER revtab        libavutil/tx_template.c:590-629, tx.c:136-155
 * Prime-factor decomposition: M 15-point transforms over the Ruritanian input map (pre-twiddled), then 15 in-place M-point
 * split-radix transforms, then the post-twiddle through the CRT output map.
 */
static int mulinv(int n, int m)
{
    n = n % m;
    for (int x = 1; x < m; x++)
        if (((n * x) % m) == 1)
            return x;
    return 0;
}

static FfoTx *pfa_create(int inv, int len, int F, float scale_f)
{
    struct FfoTx *s = calloc(1, sizeof(*s));
    const int n = len >> 1, m = n / F; /* n = F m complex points */
    const double scale = scale_f;
    int lg = 0;
    while ((1 << lg) < m)
        lg++;
    s->len = len;
    s->inv = inv;
    s->pfa_m = m;
    s->pfa_f = F;
    for (int l = 2; l <= lg; l++) {
        const int mm = 1 << l;
        const double freq = 2 * M_PI / mm;
        s->cos_tab[l] = malloc(sizeof(float) * (mm / 4 + 1));
        for (int i = 0; i < mm / 4; i++)
            s->cos_tab[l][i] = (float)cos(i * freq);
        s->cos_tab[l][mm / 4] = 0;
    }
    s->tab53[0] = s->tab53[1] = (float)cos(2 * M_PI / 5);
    s->tab53[2] = s->tab53[3] = (float)cos(2 * M_PI / 10);
    s->tab53[4] = s->tab53[5] = (float)sin(2 * M_PI / 5);
    s->tab53[6] = s->tab53[7] = (float)sin(2 * M_PI / 10);
    s->tab53[8] = s->tab53[9] = (float)cos(2 * M_PI / 12);
    s->tab53[10] = (float)cos(2 * M_PI / 6);
    s->tab53[11] = (float)cos(8 * M_PI / 6);
    /* ff_tx_init_tab_7 / _9 (tx_template.c:110-130): pairs (re, im) */
    s->tab7[0] = (float)cos(2 * M_PI / 7);  s->tab7[1] = (float)sin(2 * M_PI / 7);
    s->tab7[2] = (float)sin(2 * M_PI / 28); s->tab7[3] = (float)cos(2 * M_PI / 28);
    s->tab7[4] = (float)cos(2 * M_PI / 14); s->tab7[5] = (float)sin(2 * M_PI / 14);
    s->tab9[0] = (float)cos(2 * M_PI / 3);  s->tab9[1] = (float)sin(2 * M_PI / 3);
    s->tab9[2] = (float)cos(2 * M_PI / 9);  s->tab9[3] = (float)sin(2 * M_PI / 9);
    s->tab9[4] = (float)cos(2 * M_PI / 36); s->tab9[5] = (float)sin(2 * M_PI / 36);
    s->tab9[6] = s->tab9[2] + s->tab9[5];
    s->tab9[7] = s->tab9[3] - s->tab9[4];
    /* the sub-transform permutes on output: SCATTER revtab of the M-point split-radix FFT */
    s->sub_map = malloc(sizeof(int) * m);
    for (int i = 0; i < m; i++)
        s->sub_map[-sr_perm(i, m, inv) & (m - 1)] = i;
    /* compound map, opts == NULL: in_map gathers */
    s->map = malloc(sizeof(int) * 2 * n);
    int *in_map = s->map, *out_map = s->map + n;
    const int m_inv = mulinv(m, F), n_inv = mulinv(F, m);
    for (int j = 0; j < m; j++)
        for (int i = 0; i < F; i++) {
            in_map[j * F + i] = (i * m + j * F) % n;
            out_map[(i * m * m_inv + j * F * n_inv) % n] = i * m + j;
        }
    if (inv)
        for (int i = 0; i < m; i++) {
            int *in = &in_map[i * F + 1]; /* skip the DC */
            for (int j = 0; j < (F - 1) >> 1; j++) {
                const int t = in[j];
                in[j] = in[F - j - 2];
                in[F - j - 2] = t;
            }
        }
    /* the 15-point transform is itself 3 x 5: embed its input map */
    if (F == 15)
        for (int k = 0; k < n; k += 15) {
            int mtmp[15];
            memcpy(mtmp, &in_map[k], sizeof(mtmp));
            for (int mm = 0; mm < 5; mm++)
                for (int nn = 0; nn < 3; nn++)
                    in_map[k + mm * 3 + nn] = mtmp[(mm * 3 + nn * 5) % 15];
        }
    {
        const double theta = (scale < 0 ? n : 0) + 1.0 / 8.0;
        const double sc = sqrt(fabs(scale));
        s->exp = malloc(sizeof(cpx) * (inv ? 2 * n : n));
        cpx *e = s->exp + (inv ? n : 0);
        for (int i = 0; i < n; i++) {
            const double alpha = M_PI_2 * (i + theta) / n;
            e[i].re = (float)(cos(alpha) * sc);
            e[i].im = (float)(sin(alpha) * sc);
        }
        if (inv)
            for (int i = 0; i < n; i++)
                s->exp[i] = s->exp[n + in_map[i]];
    }
    for (int i = 0; i < n; i++)
        in_map[i] <<= 1;
    return s;
}

#define BF(x, y, a, b) do { x = (a) - (b); y = (a) + (b); } while (0)
#define CMUL(dre, dim, are, aim, bre, bim) do { (dre) = (are) * (bre) - (aim) * (bim); (dim) = (are) * (bim) + (aim) * (bre); } while (0)
#define SMUL(dre, dim, are, aim, bre, bim) do { (dre) = (are) * (bre) - (aim) * (bim); (dim) = (are) * (bim) - (aim) * (bre); } while (0)

static void fft3(const float *tab, cpx *out, const cpx *in, int stride)
{
    cpx tmp[3];
    tmp[0] = in[0];
    BF(tmp[1].re, tmp[2].im, in[1].im, in[2].im);
    BF(tmp[1].im, tmp[2].re, in[1].re, in[2].re);
    out[0 * stride].re = tmp[0].re + tmp[2].re;
    out[0 * stride].im = tmp[0].im + tmp[2].im;
    tmp[1].re = tab[8] * tmp[1].re;
    tmp[1].im = tab[9] * tmp[1].im;
    tmp[2].re = tab[10] * tmp[2].re;
    tmp[2].im = tab[10] * tmp[2].im;
    out[1 * stride].re = tmp[0].re - tmp[2].re + tmp[1].re;
    out[1 * stride].im = tmp[0].im - tmp[2].im - tmp[1].im;
    out[2 * stride].re = tmp[0].re - tmp[2].re - tmp[1].re;
    out[2 * stride].im = tmp[0].im - tmp[2].im + tmp[1].im;
}

/* fft5_m1 / _m2 / _m3: the same butterfly with the outputs at d[0..4] * stride */
static void fft5(const float *tab, cpx *out, const cpx *in, int stride, const int *d)
{
    cpx dc, z0[4], t[6];
    dc = in[0];
    BF(t[1].im, t[0].re, in[1].re, in[4].re);
    BF(t[1].re, t[0].im, in[1].im, in[4].im);
    BF(t[3].im, t[2].re, in[2].re, in[3].re);
    BF(t[3].re, t[2].im, in[2].im, in[3].im);
    out[d[0] * stride].re = dc.re + t[0].re + t[2].re;
    out[d[0] * stride].im = dc.im + t[0].im + t[2].im;
    SMUL(t[4].re, t[0].re, tab[0], tab[2], t[2].re, t[0].re);
    SMUL(t[4].im, t[0].im, tab[0], tab[2], t[2].im, t[0].im);
    CMUL(t[5].re, t[1].re, tab[4], tab[6], t[3].re, t[1].re);
    CMUL(t[5].im, t[1].im, tab[4], tab[6], t[3].im, t[1].im);
    BF(z0[0].re, z0[3].re, t[0].re, t[1].re);
    BF(z0[0].im, z0[3].im, t[0].im, t[1].im);
    BF(z0[2].re, z0[1].re, t[4].re, t[5].re);
    BF(z0[2].im, z0[1].im, t[4].im, t[5].im);
    out[d[1] * stride].re = dc.re + z0[3].re;
    out[d[1] * stride].im = dc.im + z0[0].im;
    out[d[2] * stride].re = dc.re + z0[2].re;
    out[d[2] * stride].im = dc.im + z0[1].im;
    out[d[3] * stride].re = dc.re + z0[1].re;
    out[d[3] * stride].im = dc.im + z0[2].im;
    out[d[4] * stride].re = dc.re + z0[0].re;
    out[d[4] * stride].im = dc.im + z0[3].im;
}

static void fft15(const float *tab, cpx *out, const cpx *in, int stride)
{
    static const int d1[5] = { 0, 6, 12, 3, 9 }, d2[5] = { 10, 1, 7, 13, 4 }, d3[5] = { 5, 11, 2, 8, 14 };
    cpx tmp[15];
    for (int i = 0; i < 5; i++)
        fft3(tab, tmp + i, in + i * 3, 5);
    fft5(tab, out, tmp + 0, stride, d1);
    fft5(tab, out, tmp + 5, stride, d2);
    fft5(tab, out, tmp + 10, stride, d3);
}

/*
 * fft7 (tx_template.c:250-340, float branch).  The 7-point DFT on the sums p[k] = in[k+1] + in[6-k] and differences
 * m[k] = in[k+1] - in[6-k] of the mirrored inputs: output pair (k, 7 - k) is dc + C_k -/+ i S_k, each C / S a three-term
 * expression whose term ORDER is the reference's (float addition is not associative): rows below list (table entry, operand)
 * in evaluation order, the sign applying to the product.  tab = { c1, s1, s3', c3', c2', s2' } with the reference's odd
 * angle choices (ff_tx_init_tab_7).
 */
static void fft7(const float *tab, cpx *out, const cpx *in, int stride)
{
    const float c[3] = { tab[0], tab[2], tab[4] }, sn[3] = { tab[1], tab[3], tab[5] };
    float pre[3], pim[3], mre[3], mim[3];
    for (int k = 0; k < 3; k++) {
        pre[k] = in[k + 1].re + in[6 - k].re;
        pim[k] = in[k + 1].im + in[6 - k].im;
        mre[k] = in[k + 1].re - in[6 - k].re;
        mim[k] = in[k + 1].im - in[6 - k].im;
    }
    out[0].re = in[0].re + pre[0] + pre[1] + pre[2];
    out[0].im = in[0].im + pim[0] + pim[1] + pim[2];
    /* cosine parts: z[k] = c0 * p[a] - c[b1] * p[b2] - c[c1] * p[c2] */
    static const int zre[3][5] = { { 0, 2, 2, 1, 1 }, { 2, 1, 0, 2, 1 }, { 1, 2, 0, 1, 2 } };
    static const int zim[3][5] = { { 0, 1, 1, 2, 2 }, { 2, 1, 0, 2, 1 }, { 1, 2, 0, 1, 2 } };
    float zr[3], zi[3];
    for (int k = 0; k < 3; k++) {
        zr[k] = c[0] * pre[zre[k][0]] - c[zre[k][1]] * pre[zre[k][2]] - c[zre[k][3]] * pre[zre[k][4]];
        zi[k] = c[0] * pim[zim[k][0]] - c[zim[k][1]] * pim[zim[k][2]] - c[zim[k][3]] * pim[zim[k][4]];
    }
    /* sine parts */
    const float t0re = sn[2] * mim[0] + sn[1] * mim[2] - sn[0] * mim[1];
    const float t2re = sn[0] * mim[2] + sn[2] * mim[1] - sn[1] * mim[0];
    const float t4re = sn[2] * mim[2] + sn[1] * mim[1] + sn[0] * mim[0];
    const float t0im = sn[0] * mre[0] + sn[1] * mre[1] + sn[2] * mre[2];
    const float t2im = sn[2] * mre[1] + sn[0] * mre[2] - sn[1] * mre[0];
    const float t4im = sn[2] * mre[0] + sn[1] * mre[2] - sn[0] * mre[1];
    const float are[3] = { t4re, t2re, t0re }, aim[3] = { t0im, t2im, t4im };
    const float dre = in[0].re, dim_ = in[0].im;
    for (int k = 0; k < 3; k++) {
        const float lo_re = zr[k] - are[k], hi_re = zr[k] + are[k], lo_im = zi[k] - aim[k], hi_im = zi[k] + aim[k];
        /* outputs 1, 3 take (re +, im -), output 2 the opposite; the mirrored output the other pair */
        const int o = k == 0 ? 1 : k == 1 ? 2 : 3;
        if (k == 1) {
            out[o * stride].re = dre + lo_re;
            out[o * stride].im = dim_ + hi_im;
            out[(7 - o) * stride].re = dre + hi_re;
            out[(7 - o) * stride].im = dim_ + lo_im;
        } else {
            out[o * stride].re = dre + hi_re;
            out[o * stride].im = dim_ + lo_im;
            out[(7 - o) * stride].re = dre + lo_re;
            out[(7 - o) * stride].im = dim_ + hi_im;
        }
    }
}

/* fft9 (tx_template.c:342-461, float branch): 3 x 3 with the twiddles folded in; tab = ff_tx_tab_9 as (re, im) pairs */
static void fft9(const float *tab, cpx *out, const cpx *in, int stride)
{
    const float t0r = tab[0], t0i = tab[1], t1r = tab[2], t1i = tab[3], t2r = tab[4], t2i = tab[5], t3r = tab[6], t3i = tab[7];
    cpx p[4], q[4]; /* sums / differences of in[k+1], in[8-k] */
    for (int k = 0; k < 4; k++) {
        p[k].re = in[k + 1].re + in[8 - k].re;
        p[k].im = in[k + 1].im + in[8 - k].im;
        q[k].re = in[k + 1].re - in[8 - k].re;
        q[k].im = in[k + 1].im - in[8 - k].im;
    }
    const cpx w0 = { p[0].re - p[3].re, p[0].im - p[3].im }, w1 = { p[1].re - p[3].re, p[1].im - p[3].im };
    const cpx w2 = { q[0].re - q[3].re, q[0].im - q[3].im }, w3 = { q[1].re + q[3].re, q[1].im + q[3].im };
    cpx z0 = { in[0].re + p[2].re, in[0].im + p[2].im };
    const cpx z1 = { p[0].re + p[1].re + p[3].re, p[0].im + p[1].im + p[3].im };
    out[0].re = z0.re + z1.re;
    out[0].im = z0.im + z1.im;
    cpx x[5], y[5];
    y[3].re = t0i * (q[0].re - q[1].re + q[3].re);
    y[3].im = t0i * (q[0].im - q[1].im + q[3].im);
    x[3].re = z0.re + t0r * z1.re;
    x[3].im = z0.im + t0r * z1.im;
    z0.re = in[0].re + t0r * p[2].re;
    z0.im = in[0].im + t0r * p[2].im;
    x[1].re = t1r * w0.re + t2i * w1.re;
    x[1].im = t1r * w0.im + t2i * w1.im;
    x[2].re = t2i * w0.re - t3r * w1.re;
    x[2].im = t2i * w0.im - t3r * w1.im;
    y[1].re = t1i * w2.re + t2r * w3.re;
    y[1].im = t1i * w2.im + t2r * w3.im;
    y[2].re = t2r * w2.re - t3i * w3.re;
    y[2].im = t2r * w2.im - t3i * w3.im;
    y[0].re = t0i * q[2].re;
    y[0].im = t0i * q[2].im;
    x[4].re = x[1].re + x[2].re;
    x[4].im = x[1].im + x[2].im;
    y[4].re = y[1].re - y[2].re;
    y[4].im = y[1].im - y[2].im;
    x[1].re = z0.re + x[1].re;
    x[1].im = z0.im + x[1].im;
    y[1].re = y[0].re + y[1].re;
    y[1].im = y[0].im + y[1].im;
    x[2].re = z0.re + x[2].re;
    x[2].im = z0.im + x[2].im;
    y[2].re = y[2].re - y[0].re;
    y[2].im = y[2].im - y[0].im;
    x[4].re = z0.re - x[4].re;
    x[4].im = z0.im - x[4].im;
    y[4].re = y[0].re - y[4].re;
    y[4].im = y[0].im - y[4].im;
    for (int k = 1; k <= 4; k++) {
        out[k * stride].re = x[k].re + y[k].im;
        out[k * stride].im = x[k].im - y[k].re;
        out[(9 - k) * stride].re = x[k].re - y[k].im;
        out[(9 - k) * stride].im = x[k].im + y[k].re;
    }
}

static void fft5_plain(const float *tab, cpx *out, const cpx *in, int stride)
{
    static const int d[5] = { 0, 1, 2, 3, 4 };
    fft5(tab, out, in, stride, d);
}

static void fft_small(const FfoTx *s, cpx *out, const cpx *in, int stride)
{
    switch (s->pfa_f) {
    case 3:  fft3(s->tab53, out, in, stride);       break;
    case 5:  fft5_plain(s->tab53, out, in, stride); break;
    case 7:  fft7(s->tab7, out, in, stride);        break;
    case 9:  fft9(s->tab9, out, in, stride);        break;
    default: fft15(s->tab53, out, in, stride);      break;
    }
}

static void pfa_run(const FfoTx *s, float *out, const float *in, ptrdiff_t stride)
{
    const int n = s->len >> 1, m = s->pfa_m, F = s->pfa_f;
    const int *in_map = s->map, *out_map = s->map + n;
    const cpx *exp = s->exp;
    cpx *tmp = malloc(sizeof(cpx) * n), f[15];
    int lg = 0;
    while ((1 << lg) < m)
        lg++;
    stride /= (ptrdiff_t)sizeof(float);
    if (s->inv) {
        const float *in1 = in, *in2 = in + (F * m * 2 - 1) * stride;
        cpx *z = (cpx *)out;
        const int len4 = s->len >> 2;
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < F; j++) {
                const int k = in_map[i * F + j];
                const cpx t = { in2[-k * stride], in1[k * stride] };
                CMUL(f[j].re, f[j].im, t.re, t.im, exp[i * F + j].re, exp[i * F + j].im);
            }
            fft_small(s, tmp + s->sub_map[i], f, m);
        }
        for (int i = 0; i < F; i++)
            sr_fft(s, tmp + m * i, m, lg);
        exp += n;
        for (int i = 0; i < len4; i++) {
            const int i0 = len4 + i, i1 = len4 - i - 1;
            const int s0 = out_map[i0], s1 = out_map[i1];
            const cpx src1 = { tmp[s1].im, tmp[s1].re }, src0 = { tmp[s0].im, tmp[s0].re };
            CMUL(z[i1].re, z[i0].im, src1.re, src1.im, exp[i1].im, exp[i1].re);
            CMUL(z[i0].re, z[i1].im, src0.re, src0.im, exp[i0].im, exp[i0].re);
        }
    } else {
        const int len4 = n, len3 = len4 * 3, len8 = s->len >> 2;
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < F; j++) {
                const int k = in_map[i * F + j];
                cpx t;
                if (k < len4) {
                    t.re = -in[len4 + k] + in[1 * len4 - 1 - k];
                    t.im = -in[len3 + k] + -in[1 * len3 - 1 - k];
                } else {
                    t.re = -in[len4 + k] + -in[5 * len4 - 1 - k];
                    t.im = in[-len4 + k] + -in[1 * len3 - 1 - k];
                }
                CMUL(f[j].im, f[j].re, t.re, t.im, exp[k >> 1].re, exp[k >> 1].im);
            }
            fft_small(s, tmp + s->sub_map[i], f, m);
        }
        for (int i = 0; i < F; i++)
            sr_fft(s, tmp + m * i, m, lg);
        for (int i = 0; i < len8; i++) {
            const int i0 = len8 + i, i1 = len8 - i - 1;
            const int s0 = out_map[i0], s1 = out_map[i1];
            const cpx src1 = tmp[s1], src0 = tmp[s0];
            CMUL(out[(2 * i1 + 1) * stride], out[2 * i0 * stride], src0.re, src0.im, exp[i0].im, exp[i0].re);
            CMUL(out[(2 * i0 + 1) * stride], out[2 * i1 * stride], src1.re, src1.im, exp[i1].im, exp[i1].re);
        }
    }
    free(tmp);
}
#undef BF
#undef CMUL
#undef SMUL

/*
 * AV_TX_FLOAT_FFT, power-of-two: the out-of-place wrapper gathers the input through the split-radix permutation
 * (ff_tx_fft, libavutil/tx_template.c:735-749; map from ff_tx_gen_ptwo_revtab, tx.c:125-155) and runs the same
 * "no shuffle" split-radix network the MDCT uses; the inverse differs by the permutation only.  Complex interleaved
 * (re, im) floats in and out, len of each; unnormalised.
 */
/*
 * AV_TX_FLOAT_FFT, len = F * 2^k with F = 15 / 9 / 7 / 5 / 3: ff_tx_fft_pfa over fft<F>_ns and the 2^k-point split-radix codelet
 * (libavutil/tx_template.c:948-1080; the tree av_tx_init builds for these lengths, e.g. 960 = fft15_ns x fft64_ns).  The compound
 * map is generated for the forward direction (ff_tx_gen_compound_mapping(s, opts, 0, n, m), :1032) and its input half flattened
 * through the F-point codelet's own map (:1041-1045), which carries the direction: ff_tx_gen_default_map (tx.c:525-542) for
 * 3 / 5 / 7 / 9, ff_tx_gen_pfa_input_map(3, 5) (tx.c:44-72) for 15.
 */
int ffo_fft_pfa_factor(int len)
{
    static const int f[5] = { 15, 9, 7, 5, 3 };
    for (int i = 0; i < 5; i++) {
        const int m = len / f[i];
        if (len % f[i] == 0 && m >= 4 && !(m & (m - 1)))
            return f[i];
    }
    return 0;
}

static void fft_pfa_run(int inv, int len, int F, float *out, const float *in)
{
    FfoTx *s = pfa_create(inv, 2 * len, F, 1.0f); /* the MDCT of twice the length shares the tables (cos, tab53 / 7 / 9, sub_map, out_map) */
    const int m = len / F;
    int *in_map = malloc(sizeof(int) * len), fm[15];
    const int *out_map = s->map + len;
    const cpx *src = (const cpx *)in;
    cpx *dst = (cpx *)out, *tmp = malloc(sizeof(cpx) * len), f[15];
    int lg = 0;
    while ((1 << lg) < m)
        lg++;
    for (int j = 0; j < m; j++)
        for (int i = 0; i < F; i++)
            in_map[j * F + i] = (i * m + j * F) % len;
    fm[0] = 0;
    for (int i = 1; i < F; i++)
        fm[i] = inv ? F - i : i;
    if (F == 15) {
        for (int a = 0; a < 5; a++)
            for (int b = 0; b < 3; b++) {
                if (inv)
                    fm[(a * 3 + b * 5) % 15] = a * 3 + b;
                else
                    fm[a * 3 + b] = (a * 3 + b * 5) % 15;
            }
        if (inv)
            for (int w = 1; w <= 7; w++) {
                const int t = fm[w];
                fm[w] = fm[15 - w];
                fm[15 - w] = t;
            }
    }
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < F; j++)
            f[j] = src[in_map[i * F + fm[j]]];
        fft_small(s, tmp + s->sub_map[i], f, m);
    }
    for (int i = 0; i < F; i++)
        sr_fft(s, tmp + m * i, m, lg);
    for (int i = 0; i < len; i++)
        dst[i] = tmp[out_map[i]];
    free(tmp);
    free(in_map);
    ffo_mdct_free(s);
}

void ffo_fft_run(int inv, int len, float *out, const float *in)
{
    if (ffo_fft_pfa_factor(len)) {
        fft_pfa_run(inv, len, ffo_fft_pfa_factor(len), out, in);
        return;
    }
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

/*
 * AV_TX_FLOAT_RDFT, power-of-two: ff_tx_rdft_r2c / _c2r (libavutil/tx_template.c:1601-1716).  len real samples <-> len/2 + 1
 * complex bins through a len/2-point complex FFT (the out-of-place AV_TX_FLOAT_FFT above) and one pass that separates /
 * merges the even and odd halves.  fact[] and the twiddles are computed in double and stored as float (ff_tx_rdft_init,
 * :1601-1655).  The reference's c2r works inside its input buffer; this restatement copies it.
 */
void ffo_rdft_run(int inv, int len, float scale, float *out, const float *in)
{
    const int len2 = len >> 1, len4 = len >> 2;
    const double f = 2 * M_PI / len, m = inv ? 2 * (double)scale : (double)scale;
    float fact[8];
    float *tcos = malloc(sizeof(float) * 2 * len4), *tsin = tcos + len4;
    cpx *data = malloc(sizeof(cpx) * (len2 + 1)), t[3];
    fact[0] = (float)((inv ? 0.5 : 1.0) * m);
    fact[1] = (float)(inv ? 0.5 * m : 1.0 * m);
    fact[2] = (float)m;
    fact[3] = (float)-m;
    fact[4] = (float)((0.5 - 0.0) * m);
    fact[5] = (float)((0.0 - 0.5) * m);
    fact[6] = (float)((0.5 - inv) * m);
    fact[7] = (float)(-(0.5 - inv) * m);
    for (int i = 0; i < len4; i++) {
        tcos[i] = (float)cos(i * f);
        tsin[i] = (float)(cos(((len - i * 4) / 4.0) * f) * (inv ? 1 : -1));
    }
    if (!inv) {
        ffo_fft_run(0, len2, (float *)data, in);
    } else {
        memcpy(data, in, sizeof(cpx) * (len2 + 1));
        data[0].im = data[len2].re;
    }
    t[0].re = data[0].re;
    data[0].re = t[0].re + data[0].im;
    data[0].im = t[0].re - data[0].im;
    data[0].re = fact[0] * data[0].re;
    data[0].im = fact[1] * data[0].im;
    data[len4].re = fact[2] * data[len4].re;
    data[len4].im = fact[3] * data[len4].im;
    for (int i = 1; i < len4; i++) {
        t[0].re = fact[4] * (data[i].re + data[len2 - i].re);
        t[0].im = fact[5] * (data[i].im - data[len2 - i].im);
        t[1].re = fact[6] * (data[i].im + data[len2 - i].im);
        t[1].im = fact[7] * (data[i].re - data[len2 - i].re);
        t[2].re = t[1].re * tcos[i] - t[1].im * tsin[i];
        t[2].im = t[1].re * tsin[i] + t[1].im * tcos[i];
        data[i].re = t[0].re + t[2].re;
        data[i].im = t[2].im - t[0].im;
        data[len2 - i].re = t[0].re - t[2].re;
        data[len2 - i].im = t[2].im + t[0].im;
    }
    if (inv) {
        ffo_fft_run(1, len2, out, (const float *)data);
    } else {
        data[len2].re = data[0].im;
        data[0].im = data[len2].im = 0;
        memcpy(out, data, sizeof(cpx) * (len2 + 1));
    }
    free(data);
    free(tcos);
}

/*
 * AV_TX_FLOAT_RDFT with AV_TX_REAL_TO_REAL (mode 1) / AV_TX_REAL_TO_IMAGINARY (mode 2), len % 4 == 0, power of two:
 * ff_tx_rdft_r2r / ff_tx_rdft_r2i (libavutil/tx_template.c:1718-1830; forward only).  len reals -> the real parts of bins
 * 0 .. len/2 (len/2 + 1 floats) resp. len/2 floats of imaginary parts.  The reference runs in place over the FFT's output,
 * one float array aliasing the complex one; it is restated the same way (the buffer is the FFT output, `o` aliases it), so
 * the table quirks come out by themselves: bin len/4 enters the loop after its two scalings, tcos[len/4] is tsin[0],
 * tsin[len/4] is the zero of the reference's over-sized, zeroed table, and r2i's last output is the FFT's own
 * data[len/2 - 1].im, which the copy loop picks up from a slot the main loop never writes.
 */
void ffo_rdft_half_run(int mode, int len, float scale, float *out, const float *in)
{
    const int len2 = len >> 1, len4 = len >> 2;
    const double f = 2 * M_PI / len, m = (double)scale;
    float fact[8];
    float *tcos = calloc(2 * len4 + 1, sizeof(float)), *tsin = tcos + len4;
    cpx *data = malloc(sizeof(cpx) * (len2 + 1));
    float *o = (float *)data;
    float tmp_dc;
    fact[0] = (float)(1.0 * m);
    fact[1] = (float)(1.0 * m);
    fact[2] = (float)m;
    fact[3] = (float)-m;
    fact[4] = (float)((0.5 - 0.0) * m);
    fact[5] = mode == 1 ? 1 / scale : (float)((0.0 - 0.5) * m);
    fact[6] = (float)((0.5 - 0) * m);
    fact[7] = (float)(-(0.5 - 0) * m);
    for (int i = 0; i < len4; i++) {
        tcos[i] = (float)cos(i * f);
        tsin[i] = (float)(cos(((len - i * 4) / 4.0) * f) * -1);
    }
    ffo_fft_run(0, len2, (float *)data, in);
    tmp_dc = data[0].re;
    data[0].re = tmp_dc + data[0].im;
    tmp_dc = tmp_dc - data[0].im;
    data[0].re = fact[0] * data[0].re;
    tmp_dc = fact[1] * tmp_dc;
    data[len4].re = fact[2] * data[len4].re;
    data[len4].im = fact[3] * data[len4].im;
    for (int i = 1; i <= len4; i++) {
        float t[4];
        const cpx sf = data[i], sl = data[len2 - i];
        if (mode == 1)
            t[0] = fact[4] * (sf.re + sl.re);
        else
            t[0] = fact[5] * (sf.im - sl.im);
        t[1] = fact[6] * (sf.im + sl.im);
        t[2] = fact[7] * (sf.re - sl.re);
        if (mode == 1) {
            t[3] = t[1] * tcos[i] - t[2] * tsin[i];
            o[i] = t[0] + t[3];
            o[len - i] = t[0] - t[3];
        } else {
            t[3] = t[1] * tsin[i] + t[2] * tcos[i];
            o[i - 1] = t[3] - t[0];
            o[len - i - 1] = t[0] + t[3];
        }
    }
    for (int i = 1; i < len4 + (mode == 2); i++)
        o[len2 - i] = o[len - i];
    if (mode == 1)
        o[len2] = tmp_dc;
    memcpy(out, o, sizeof(float) * (len2 + (mode == 1)));
    free(data);
    free(tcos);
}

/*
 * AV_TX_FLOAT_DCT_I / AV_TX_FLOAT_DST_I, forward (ff_tx_dctI / ff_tx_dstI on ff_tx_dcstI_init, libavutil/tx_template.c:2006-2075): n
 * even.  The input mirrored into 2 (n - 1) reals resp. 2 (n + 1) reals with the odd symmetry, then the half-complex RDFT of that
 * length — 2 mod 4, so the _mod2 forms of ff_tx_rdft_r2r / _r2i (:1718-1830) with the pair in the middle handled before the loop
 * (after data[len4].re took its factor) and, for r2r, out[len4 + 1] = tmp_mid * (1 / scale).  Restated in floats in the reference's
 * order, in place over the FFT's output as there; the FFT itself has n -+ 1 points, an odd number the reference serves with a
 * prime-factor or the naive codelet: here the naive sum in double precision, rounded to float (ff_tx_fft_naive, :1016-1046, sums in
 * float; the difference is rounding, which the tests' tolerance covers).  is_dst: 0 DCT-I, 1 DST-I.
 */
void ffo_dcst1_run(int is_dst, int n, float scale, float *out, const float *in, ptrdiff_t stride)
{
    const int ln = is_dst ? n + 1 : n - 1, len = 2 * ln, len2 = len >> 1, len4 = len >> 2, al4 = (len + 3) / 4;
    const int mode = is_dst ? 2 : 1;
    const double f = 2 * M_PI / len, m = (double)scale;
    float fact[8];
    float *tmp = calloc(len + 2, sizeof(float));
    float *tcos = calloc(2 * al4, sizeof(float)), *tsin = tcos + al4;
    cpx *data = calloc(len2 + 2, sizeof(cpx));
    float *o = (float *)data;
    float tmp_dc, tmp_mid, t[4];
    cpx sf, sl;
    stride /= (ptrdiff_t)sizeof(float);
    if (!is_dst) {
        for (int i = 0; i < ln; i++)
            tmp[i] = tmp[2 * ln - i] = in[i * stride];
        tmp[ln] = in[ln * stride];
    } else {
        tmp[0] = 0;
        for (int i = 1; i < ln; i++) {
            const float a = in[(i - 1) * stride];
            tmp[i] = -a;
            tmp[2 * ln - i] = a;
        }
        tmp[ln] = 0;
    }
    fact[0] = (float)(1.0 * m);
    fact[1] = (float)(1.0 * m);
    fact[2] = (float)m;
    fact[3] = (float)-m;
    fact[4] = (float)((0.5 - 0.0) * m);
    fact[5] = mode == 1 ? 1 / scale : (float)((0.0 - 0.5) * m);
    fact[6] = (float)((0.5 - 0) * m);
    fact[7] = (float)(-(0.5 - 0) * m);
    for (int i = 0; i < al4; i++) {
        tcos[i] = (float)cos(i * f);
        tsin[i] = (float)cos(((len - i * 4) / 4.0) * f) * -1;
    }
    for (int k = 0; k < len2; k++) {
        double re = 0, im = 0;
        for (int j = 0; j < len2; j++) {
            const double a = -2 * M_PI * (double)((long)j * k % len2) / len2, c = cos(a), sn = sin(a);
            re += tmp[2 * j] * c - tmp[2 * j + 1] * sn;
            im += tmp[2 * j] * sn + tmp[2 * j + 1] * c;
        }
        data[k].re = (float)re;
        data[k].im = (float)im;
    }
    tmp_dc = data[0].re;
    data[0].re = tmp_dc + data[0].im;
    tmp_dc = tmp_dc - data[0].im;
    data[0].re = fact[0] * data[0].re;
    tmp_dc = fact[1] * tmp_dc;
    data[len4].re = fact[2] * data[len4].re;
    sf = data[len4];
    sl = data[len4 + 1];
    if (mode == 1)
        t[0] = fact[4] * (sf.re + sl.re);
    else
        t[0] = fact[5] * (sf.im - sl.im);
    t[1] = fact[6] * (sf.im + sl.im);
    t[2] = fact[7] * (sf.re - sl.re);
    if (mode == 1) {
        t[3] = t[1] * tcos[len4] - t[2] * tsin[len4];
        tmp_mid = t[0] - t[3];
    } else {
        t[3] = t[1] * tsin[len4] + t[2] * tcos[len4];
        tmp_mid = t[0] + t[3];
    }
    for (int i = 1; i <= len4; i++) {
        sf = data[i];
        sl = data[len2 - i];
        if (mode == 1)
            t[0] = fact[4] * (sf.re + sl.re);
        else
            t[0] = fact[5] * (sf.im - sl.im);
        t[1] = fact[6] * (sf.im + sl.im);
        t[2] = fact[7] * (sf.re - sl.re);
        if (mode == 1) {
            t[3] = t[1] * tcos[i] - t[2] * tsin[i];
            o[i] = t[0] + t[3];
            o[len - i] = t[0] - t[3];
        } else {
            t[3] = t[1] * tsin[i] + t[2] * tcos[i];
            o[i - 1] = t[3] - t[0];
            o[len - i - 1] = t[0] + t[3];
        }
    }
    for (int i = 1; i < len4 + (mode == 2); i++)
        o[len2 - i] = o[len - i];
    if (mode == 1) {
        o[len2] = tmp_dc;
        o[len4 + 1] = tmp_mid * fact[5];
    } else {
        o[len4] = tmp_mid;
    }
    memcpy(out, o, sizeof(float) * n);
    free(data);
    free(tcos);
    free(tmp);
}

/*
 * AV_TX_FLOAT_DCT, power-of-two: ff_tx_dctII (forward) / ff_tx_dctIII (inverse) on top of the RDFT
 * (libavutil/tx_template.c:1832-2002).  n is the number of real samples (av_tx_init is handed n for the forward and n / 2 for
 * the inverse transform, ff_tx_dct_init doubles it); the RDFT runs with scale resp. scale / 2.  Tables in double, stored as
 * float.  The forward transform's odd outputs are a running sum from the Nyquist bin down: the order of those additions is part
 * of the result.  Out of place here; the reference also clobbers its input, which no caller can rely on.
 */
void ffo_dct_run(int inv, int n, float scale, float *out, const float *in)
{
    const int h = n / 2;
    const double freq = M_PI / (n * 2);
    float *ex = malloc(sizeof(float) * (n + h)), *buf = calloc(n + 2, sizeof(float));
    for (int i = 0; i < n; i++)
        ex[i] = (float)(cos(i * freq) * (!inv + 1));
    for (int i = 0; i < h; i++)
        ex[n + i] = inv ? (float)(0.5 / sin((2 * i + 1) * freq)) : (float)cos((n - 2 * i - 1) * freq);
    if (!inv) {
        float *bins = malloc(sizeof(float) * (n + 2));
        for (int i = 0; i < h; i++) {
            const float a = in[i], b = in[n - 1 - i];
            const float t1 = (a + b) * 0.5f, t2 = (a - b) * ex[n + i];
            buf[i] = t1 + t2;
            buf[n - 1 - i] = t1 - t2;
        }
        ffo_rdft_run(0, n, scale, bins, buf);
        float next = bins[n];
        for (int i = n - 2; i > 0; i -= 2) {
            const float re = bins[i], im = bins[i + 1];
            const float t = ex[n - i] * re - ex[i] * im;
            out[i] = ex[n - i] * im + ex[i] * re;
            out[i + 1] = next;
            next += t;
        }
        out[0] = ex[0] * bins[0];
        out[1] = next;
        free(bins);
    } else {
        buf[0] = in[0];
        buf[1] = in[1];
        buf[n] = 2 * in[n - 1];
        buf[n + 1] = 0;
        for (int i = 2; i < n; i += 2) {
            const float v1 = in[i], v2 = in[i - 1] - in[i + 1];
            buf[i + 1] = ex[n - i] * v1 - ex[i] * v2;
            buf[i] = ex[n - i] * v2 + ex[i] * v1;
        }
        ffo_rdft_run(1, n, scale * 0.5f, out, buf);
        for (int i = 0; i < h; i++) {
            const float a = out[i], b = out[n - 1 - i];
            const float t1 = a + b, t2 = (a - b) * ex[n + i];
            out[i] = t1 + t2;
            out[n - 1 - i] = t1 - t2;
        }
    }
    free(buf);
    free(ex);
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
