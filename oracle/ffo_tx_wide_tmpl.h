/*
 * ffo_tx_wide_tmpl.h — one sample type of oracle/ffo_tx_wide.c (included twice: W_INT 0 = double, 1 = int32).  TEST INFRASTRUCTURE.
 * Restates libavutil/tx_template.c as compiled by tx_double.c / tx_int32.c:
 *   CMUL, BF, FOLD, RESCALE                 libavutil/tx_priv.h:88-147
 *   ff_tx_init_tab_<len>                    libavutil/tx_template.c:69-79
 *   fft2 / fft4 / fft8 / fft16 codelets     libavutil/tx_template.c:612-700
 *   ff_tx_fft_sr_combine, DECL_SR_CODELET   libavutil/tx_template.c:540-610
 *   ff_tx_fft (the gather through the map)  libavutil/tx_template.c:735-749; split_radix_permutation libavutil/tx.c:125-155
 *   ff_tx_mdct_fwd / _inv, gen_exp          libavutil/tx_template.c:1272-1342,2107-2134
 */
#if W_INT
#define WT int32_t
#define WN(x) x##_i32
static inline void WN(cmul)(WT *dre, WT *dim, WT are, WT aim, WT bre, WT bim)
{
    int64_t accu = (int64_t)bre * are;
    accu -= (int64_t)bim * aim;
    *dre = (int)((accu + 0x40000000) >> 31);
    accu = (int64_t)bim * are;
    accu += (int64_t)bre * aim;
    *dim = (int)((accu + 0x40000000) >> 31);
}
static inline WT WN(sub)(WT a, WT b) { return (WT)((uint32_t)a - (uint32_t)b); }
static inline WT WN(add)(WT a, WT b) { return (WT)((uint32_t)a + (uint32_t)b); }
static inline WT WN(neg)(WT a) { return (WT)(0u - (uint32_t)a); }
static inline WT WN(fold)(WT a, WT b) { return (int32_t)((uint32_t)a + (uint32_t)b + 32u) >> 6; }
static inline WT WN(rescale)(double x)
{
    long long v = llrintf((float)(x * 2147483648.0));
    return (WT)(v < INT32_MIN ? INT32_MIN : v > INT32_MAX ? INT32_MAX : v);
}
#else
#define WT double
#define WN(x) x##_f64
static inline void WN(cmul)(WT *dre, WT *dim, WT are, WT aim, WT bre, WT bim)
{
    *dre = are * bre - aim * bim;
    *dim = are * bim + aim * bre;
}
static inline WT WN(sub)(WT a, WT b) { return a - b; }
static inline WT WN(add)(WT a, WT b) { return a + b; }
static inline WT WN(neg)(WT a) { return -a; }
static inline WT WN(fold)(WT a, WT b) { return a + b; }
static inline WT WN(rescale)(double x) { return x; }
#endif

typedef struct { WT re, im; } WN(cpx);

/* BUTTERFLIES with t1, t2, t5, t6 given (tx_template.c:512-524) */
static inline void WN(butterflies)(WN(cpx) *a0, WN(cpx) *a1, WN(cpx) *a2, WN(cpx) *a3, WT t1, WT t2, WT t5, WT t6)
{
    const WT r0 = a0->re, i0 = a0->im, r1 = a1->re, i1 = a1->im;
    const WT t3 = WN(sub)(t5, t1), s5 = WN(add)(t5, t1);
    a2->re = WN(sub)(r0, s5); a0->re = WN(add)(r0, s5);
    a3->im = WN(sub)(i1, t3); a1->im = WN(add)(i1, t3);
    const WT t4 = WN(sub)(t2, t6), s6 = WN(add)(t2, t6);
    a3->re = WN(sub)(r1, t4); a1->re = WN(add)(r1, t4);
    a2->im = WN(sub)(i0, s6); a0->im = WN(add)(i0, s6);
}
static inline void WN(transform)(WN(cpx) *a0, WN(cpx) *a1, WN(cpx) *a2, WN(cpx) *a3, WT wre, WT wim)
{
    WT t1, t2, t5, t6;
    WN(cmul)(&t1, &t2, a2->re, a2->im, wre, WN(neg)(wim));
    WN(cmul)(&t5, &t6, a3->re, a3->im, wre, wim);
    WN(butterflies)(a0, a1, a2, a3, t1, t2, t5, t6);
}

static void WN(fft2)(WN(cpx) *z)
{
    const WN(cpx) d = { WN(sub)(z[0].re, z[1].re), WN(sub)(z[0].im, z[1].im) };
    z[0].re = WN(add)(z[0].re, z[1].re);
    z[0].im = WN(add)(z[0].im, z[1].im);
    z[1] = d;
}
static void WN(fft4)(WN(cpx) *z) /* ff_tx_fft4_ns */
{
    const WT t3 = WN(sub)(z[0].re, z[1].re), t1 = WN(add)(z[0].re, z[1].re);
    const WT t8 = WN(sub)(z[3].re, z[2].re), t6 = WN(add)(z[3].re, z[2].re);
    const WT t4 = WN(sub)(z[0].im, z[1].im), t2 = WN(add)(z[0].im, z[1].im);
    const WT t7 = WN(sub)(z[2].im, z[3].im), t5 = WN(add)(z[2].im, z[3].im);
    z[2].re = WN(sub)(t1, t6); z[0].re = WN(add)(t1, t6);
    z[3].im = WN(sub)(t4, t8); z[1].im = WN(add)(t4, t8);
    z[3].re = WN(sub)(t3, t7); z[1].re = WN(add)(t3, t7);
    z[2].im = WN(sub)(t2, t5); z[0].im = WN(add)(t2, t5);
}
static void WN(fft8)(WN(cpx) *z, WT *const *tabs) /* ff_tx_fft8_ns */
{
    const WT c = tabs[3][1];
    WN(fft4)(z);
    const WT t1 = WN(add)(z[4].re, z[5].re), t2 = WN(add)(z[4].im, z[5].im);
    const WT t5 = WN(add)(z[6].re, z[7].re), t6 = WN(add)(z[6].im, z[7].im);
    z[5].re = WN(sub)(z[4].re, z[5].re); z[5].im = WN(sub)(z[4].im, z[5].im);
    z[7].re = WN(sub)(z[6].re, z[7].re); z[7].im = WN(sub)(z[6].im, z[7].im);
    WN(butterflies)(&z[0], &z[2], &z[4], &z[6], t1, t2, t5, t6);
    WN(transform)(&z[1], &z[3], &z[5], &z[7], c, c);
}
static void WN(fft16)(WN(cpx) *z, WT *const *tabs) /* ff_tx_fft16_ns */
{
    const WT *c = tabs[4];
    WN(fft8)(z, tabs);
    WN(fft4)(z + 8);
    WN(fft4)(z + 12);
    WN(butterflies)(&z[0], &z[4], &z[8], &z[12], z[8].re, z[8].im, z[12].re, z[12].im);
    WN(transform)(&z[2], &z[6], &z[10], &z[14], c[2], c[2]);
    WN(transform)(&z[1], &z[5], &z[9], &z[13], c[1], c[3]);
    WN(transform)(&z[3], &z[7], &z[11], &z[15], c[3], c[1]);
}
static void WN(sr_fft)(WN(cpx) *z, int n, int lg, WT *const *tabs)
{
    if (n == 1) return;
    if (n == 2) { WN(fft2)(z); return; }
    if (n == 4) { WN(fft4)(z); return; }
    if (n == 8) { WN(fft8)(z, tabs); return; }
    if (n == 16) { WN(fft16)(z, tabs); return; }
    const int q = n >> 2;
    WN(sr_fft)(z, n >> 1, lg - 1, tabs);
    WN(sr_fft)(z + 2 * q, q, lg - 2, tabs);
    WN(sr_fft)(z + 3 * q, q, lg - 2, tabs);
    for (int k = 0; k < q; k++) /* ff_tx_fft_sr_combine: TRANSFORM at every k, k = 0 included */
        WN(transform)(&z[k], &z[k + q], &z[k + 2 * q], &z[k + 3 * q], tabs[lg][k], tabs[lg][q - k]);
}

static WT **WN(make_tabs)(int lg)
{
    WT **tabs = calloc(24, sizeof(*tabs));
    for (int l = 3; l <= lg; l++) {
        const int m = 1 << l;
        const double freq = 2 * M_PI / m;
        tabs[l] = malloc(sizeof(WT) * (m / 4 + 1));
        for (int i = 0; i < m / 4; i++)
            tabs[l][i] = WN(rescale)(cos(i * freq));
        tabs[l][m / 4] = 0;
    }
    return tabs;
}
static void WN(free_tabs)(WT **tabs)
{
    for (int l = 0; l < 24; l++)
        free(tabs[l]);
    free(tabs);
}

static void WN(fft_run)(int inv, int len, void *out, const void *in)
{
    int lg = 0;
    while ((1 << lg) < len)
        lg++;
    WT **tabs = WN(make_tabs)(lg);
    const WN(cpx) *src = in;
    WN(cpx) *z = out;
    for (int i = 0; i < len; i++)
        z[i] = src[-txw_sr_perm(i, len, inv) & (len - 1)];
    WN(sr_fft)(z, len, lg, tabs);
    WN(free_tabs)(tabs);
}

static void WN(mdct_run)(int inv, int len, double scale, void *out_, const void *in_)
{
    const int n = len >> 1, q = len >> 2, len3 = 3 * n;
    int lg = 0;
    while ((1 << lg) < n)
        lg++;
    WT **tabs = WN(make_tabs)(lg);
    int *map = malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++)
        map[i] = -txw_sr_perm(i, n, inv) & (n - 1);      /* ff_tx_gen_ptwo_revtab: GATHER map[i] = k ... */
    if (!inv) {                                          /* ... SCATTER map[k] = i */
        int *sc = malloc(sizeof(int) * n);
        for (int i = 0; i < n; i++)
            sc[map[i]] = i;
        free(map);
        map = sc;
    }
    WN(cpx) *ex = malloc(sizeof(*ex) * n), *z = malloc(sizeof(*z) * n);
    const double theta = (scale < 0 ? n : 0) + 1.0 / 8.0, rt = sqrt(fabs(scale));
    for (int i = 0; i < n; i++) {
        const double alpha = M_PI_2 * (i + theta) / n;
        double sn, cs;
        sincos(alpha, &sn, &cs);   /* what gcc makes of the reference's cos(alpha), sin(alpha) pair: not always cos()'s last bit */
        ex[i].re = WN(rescale)(cs * rt);
        ex[i].im = WN(rescale)(sn * rt);
    }
    const WT *in = in_;
    WT *out = out_;
    if (!inv) {
        for (int i = 0; i < n; i++) {
            const int k = 2 * i;
            WT re, im;
            if (k < n) {
                re = WN(fold)(WN(neg)(in[n + k]), in[n - 1 - k]);
                im = WN(fold)(WN(neg)(in[len3 + k]), WN(neg)(in[len3 - 1 - k]));
            } else {
                re = WN(fold)(WN(neg)(in[n + k]), WN(neg)(in[5 * n - 1 - k]));
                im = WN(fold)(in[k - n], WN(neg)(in[len3 - 1 - k]));
            }
            WN(cmul)(&z[map[i]].im, &z[map[i]].re, re, im, ex[i].re, ex[i].im);
        }
        WN(sr_fft)(z, n, lg, tabs);
        for (int i = 0; i < q; i++) {
            const int i0 = q + i, i1 = q - i - 1;
            const WN(cpx) s1 = z[i1], s0 = z[i0];
            WN(cmul)(&out[2 * i1 + 1], &out[2 * i0], s0.re, s0.im, ex[i0].im, ex[i0].re);
            WN(cmul)(&out[2 * i0 + 1], &out[2 * i1], s1.re, s1.im, ex[i1].im, ex[i1].re);
        }
    } else {
        const WT *in2 = in + 2 * n - 1;
        WN(cpx) *o = (WN(cpx) *)out;
        for (int i = 0; i < n; i++) {
            const int k = map[i] << 1;
            WN(cmul)(&z[i].re, &z[i].im, in2[-k], in[k], ex[map[i]].re, ex[map[i]].im);
        }
        WN(sr_fft)(z, n, lg, tabs);
        for (int i = 0; i < q; i++) {
            const int i0 = q + i, i1 = q - i - 1;
            const WN(cpx) s1 = { z[i1].im, z[i1].re }, s0 = { z[i0].im, z[i0].re };
            WN(cmul)(&o[i1].re, &o[i0].im, s1.re, s1.im, ex[i1].im, ex[i1].re);
            WN(cmul)(&o[i0].re, &o[i1].im, s0.re, s0.im, ex[i0].im, ex[i0].re);
        }
    }
    free(z); free(ex); free(map);
    WN(free_tabs)(tabs);
}
#undef WT
#undef WN
