/*
 * ffo_vp9.c — CPU restatement of the VP9 inverse transforms, 8 bits: VP9DSPContext.itxfm_add[tx][txtp]
 * (libavcodec/vp9dsp_template.c:1155-1776).  TEST INFRASTRUCTURE ONLY (see ffo.h).
 *
 * The reference spells each 1-D transform out as one long butterfly listing.  Every multiplication is followed by its own
 * rounding ((x + 2^13) >> 14) and intermediates wrap in 32 bits, so the network — not just the matrix — is the definition.
 * It is restated here from its structure:
 *   an N-point inverse DCT is the N/2-point one on the even inputs plus an "odd part" on the odd inputs,
 *   out[i] = E[i] + O[i], out[N-1-i] = E[i] - O[i]  (idct4 inside idct8 inside idct16 inside idct32; :1202-1716);
 *   the odd parts are stages of plane rotations ROT / NROT, sum-difference butterflies and 1/sqrt2 scalings HALF on arrays;
 *   the ADSTs (:1218-1232,1272-1314,1406-1507) keep their products unrounded across the first butterfly, as the reference does.
 * All arithmetic is unsigned 32-bit with an arithmetic right shift of the reinterpreted sum, which is what the reference's
 * (dctint)(... U ...) >> 14 expressions do; between the two passes values are stored as int16 (dctcoef).
 */
#include <stdint.h>
#include <string.h>

#include "ffo.h"

typedef uint32_t u32;
#define R14(x) ((int32_t)((u32)(x) + (1u << 13)) >> 14)

/* (a c - b s, a s + b c), each rounded */
static void rot(int32_t a, int32_t b, u32 c, u32 s, int32_t *lo, int32_t *hi)
{
    *lo = R14((u32)a * c - (u32)b * s);
    *hi = R14((u32)a * s + (u32)b * c);
}
/* (-(a s + b c), a c - b s), each rounded: the negation happens before the rounding */
static void nrot(int32_t a, int32_t b, u32 c, u32 s, int32_t *lo, int32_t *hi)
{
    *lo = R14(-((u32)a * s + (u32)b * c));
    *hi = R14((u32)a * c - (u32)b * s);
}
/* ((a - b), (a + b)) / sqrt 2 */
static void half(int32_t a, int32_t b, int32_t *lo, int32_t *hi)
{
    *lo = R14(((u32)a - (u32)b) * 11585u);
    *hi = R14(((u32)a + (u32)b) * 11585u);
}

/* ---- inverse DCT: even part by recursion, odd parts below; o[] of an odd part is ordered so that out[i] = e[i] + o[i] ---- */
static void idct2_even(int32_t x0, int32_t x1, int32_t *e) /* the 2-point core: x0, x1 = inputs 0 and N/2 */
{
    half(x0, x1, &e[1], &e[0]);
}

static void idct4_core(const int32_t *x, int32_t *out) /* x[0..3] natural order */
{
    int32_t e[2], t2, t3;
    idct2_even(x[0], x[2], e);
    rot(x[1], x[3], 6270, 15137, &t2, &t3);
    out[0] = e[0] + t3;
    out[1] = e[1] + t2;
    out[2] = e[1] - t2;
    out[3] = e[0] - t3;
}

static void odd8(const int32_t *x, int32_t *o) /* x = inputs 1, 3, 5, 7 */
{
    int32_t a4, a7, a5, a6, d5, d6;
    rot(x[0], x[3], 3196, 16069, &a4, &a7);
    rot(x[2], x[1], 13623, 9102, &a5, &a6);
    o[3] = a4 + a5;
    d5 = a4 - a5;
    o[0] = a7 + a6;
    d6 = a7 - a6;
    half(d6, d5, &o[2], &o[1]);
}

static void odd16(const int32_t *x, int32_t *o) /* x = inputs 1, 3, ..., 15 */
{
    int32_t a[8], t[8]; /* a[k] = t(8+k)a, t[k] = t(8+k) of the listing */
    rot(x[0], x[7], 1606, 16305, &a[0], &a[7]);
    rot(x[4], x[3], 12665, 10394, &a[1], &a[6]);
    rot(x[2], x[5], 7723, 14449, &a[2], &a[5]);
    rot(x[6], x[1], 15679, 4756, &a[3], &a[4]);
    t[0] = a[0] + a[1]; t[1] = a[0] - a[1]; t[2] = a[3] - a[2]; t[3] = a[3] + a[2];
    t[4] = a[4] + a[5]; t[5] = a[4] - a[5]; t[6] = a[7] - a[6]; t[7] = a[7] + a[6];
    rot(t[6], t[1], 6270, 15137, &a[1], &a[6]);
    nrot(t[5], t[2], 6270, 15137, &a[2], &a[5]);
    a[0] = t[0] + t[3]; a[3] = t[0] - t[3];
    t[1] = a[1] + a[2]; t[2] = a[1] - a[2];
    a[4] = t[7] - t[4]; a[7] = t[7] + t[4];
    t[5] = a[6] - a[5]; t[6] = a[6] + a[5];
    half(t[5], t[2], &a[2], &a[5]);
    half(a[4], a[3], &t[3], &t[4]);
    o[0] = a[7]; o[1] = t[6]; o[2] = a[5]; o[3] = t[4]; o[4] = t[3]; o[5] = a[2]; o[6] = t[1]; o[7] = a[0];
}

static void odd32(const int32_t *x, int32_t *o) /* x = inputs 1, 3, ..., 31 */
{
    int32_t a[16], t[16]; /* index k stands for t(16+k) */
    rot(x[0], x[15], 804, 16364, &a[0], &a[15]);
    rot(x[8], x[7], 12140, 11003, &a[1], &a[14]);
    rot(x[4], x[11], 7005, 14811, &a[2], &a[13]);
    rot(x[12], x[3], 15426, 5520, &a[3], &a[12]);
    rot(x[2], x[13], 3981, 15893, &a[4], &a[11]);
    rot(x[10], x[5], 14053, 8423, &a[5], &a[10]);
    rot(x[6], x[9], 9760, 13160, &a[6], &a[9]);
    rot(x[14], x[1], 16207, 2404, &a[7], &a[8]);
    for (int k = 0; k < 16; k += 4) { /* pairs (k, k+1) sum / difference, (k+2, k+3) mirrored */
        t[k] = a[k] + a[k + 1];
        t[k + 1] = a[k] - a[k + 1];
        t[k + 2] = a[k + 3] - a[k + 2];
        t[k + 3] = a[k + 3] + a[k + 2];
    }
    rot(t[14], t[1], 3196, 16069, &a[1], &a[14]);
    nrot(t[13], t[2], 3196, 16069, &a[2], &a[13]);
    rot(t[10], t[5], 13623, 9102, &a[5], &a[10]);
    nrot(t[9], t[6], 13623, 9102, &a[6], &a[9]);
    a[0] = t[0] + t[3];   a[3] = t[0] - t[3];
    t[1] = a[1] + a[2];   t[2] = a[1] - a[2];
    a[4] = t[7] - t[4];   a[7] = t[7] + t[4];
    t[5] = a[6] - a[5];   t[6] = a[6] + a[5];
    a[8] = t[8] + t[11];  a[11] = t[8] - t[11];
    t[9] = a[9] + a[10];  t[10] = a[9] - a[10];
    a[12] = t[15] - t[12]; a[15] = t[15] + t[12];
    t[13] = a[14] - a[13]; t[14] = a[14] + a[13];
    rot(t[13], t[2], 6270, 15137, &a[2], &a[13]);
    rot(a[12], a[3], 6270, 15137, &t[3], &t[12]);
    nrot(a[11], a[4], 6270, 15137, &t[4], &t[11]);
    nrot(t[10], t[5], 6270, 15137, &a[5], &a[10]);
    {
        int32_t n[16];
        n[0] = a[0] + a[7];    n[7] = a[0] - a[7];
        n[1] = t[1] + t[6];    n[6] = t[1] - t[6];
        n[2] = a[2] + a[5];    n[5] = a[2] - a[5];
        n[3] = t[3] + t[4];    n[4] = t[3] - t[4];
        n[8] = a[15] - a[8];   n[15] = a[15] + a[8];
        n[9] = t[14] - t[9];   n[14] = t[14] + t[9];
        n[10] = a[13] - a[10]; n[13] = a[13] + a[10];
        n[11] = t[12] - t[11]; n[12] = t[12] + t[11];
        half(n[11], n[4], &n[4], &n[11]);
        half(n[10], n[5], &n[5], &n[10]);
        half(n[9], n[6], &n[6], &n[9]);
        half(n[8], n[7], &n[7], &n[8]);
        for (int k = 0; k < 16; k++)
            o[k] = n[15 - k];
    }
}

static void idct_n(int n, const int32_t *x, int32_t *out)
{
    if (n == 4) {
        idct4_core(x, out);
        return;
    }
    int32_t ev[16], od[16], e[16], o[16];
    for (int k = 0; k < n / 2; k++) {
        ev[k] = x[2 * k];
        od[k] = x[2 * k + 1];
    }
    idct_n(n / 2, ev, e);
    if (n == 8)
        odd8(od, o);
    else if (n == 16)
        odd16(od, o);
    else
        odd32(od, o);
    for (int k = 0; k < n / 2; k++) {
        out[k] = e[k] + o[k];
        out[n - 1 - k] = e[k] - o[k];
    }
}

/* ---- inverse ADST ---- */
static void iadst4(const int32_t *x, int32_t *out)
{
    const u32 x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    const u32 t0 = 5283u * x0 + 15212u * x2 + 9929u * x3;
    const u32 t1 = 9929u * x0 - 5283u * x2 - 15212u * x3;
    const u32 t2 = 13377u * (x0 - x2 + x3);
    const u32 t3 = 13377u * x1;
    out[0] = R14(t0 + t3);
    out[1] = R14(t1 + t3);
    out[2] = R14(t2);
    out[3] = R14(t0 + t1 - t3);
}

static void iadst8(const int32_t *x, int32_t *out)
{
    static const u32 c[4][2] = { { 16305, 1606 }, { 14449, 7723 }, { 10394, 12665 }, { 4756, 15679 } };
    u32 p[8];
    int32_t t[8];
    for (int k = 0; k < 4; k++) { /* pairs (x[7-2k], x[2k]) */
        const u32 a = x[7 - 2 * k], b = x[2 * k];
        p[2 * k] = c[k][0] * a + c[k][1] * b;
        p[2 * k + 1] = c[k][1] * a - c[k][0] * b;
    }
    for (int k = 0; k < 4; k++) {
        t[k] = R14(p[k] + p[k + 4]);
        t[k + 4] = R14(p[k] - p[k + 4]);
    }
    {
        const u32 q4 = 15137u * (u32)t[4] + 6270u * (u32)t[5], q5 = 6270u * (u32)t[4] - 15137u * (u32)t[5];
        const u32 q6 = 15137u * (u32)t[7] - 6270u * (u32)t[6], q7 = 6270u * (u32)t[7] + 15137u * (u32)t[6];
        const int32_t s2 = t[0] - t[2], s3 = t[1] - t[3];
        const int32_t s6 = R14(q4 - q6), s7 = R14(q5 - q7);
        out[0] = t[0] + t[2];
        out[7] = -(t[1] + t[3]);
        out[1] = -R14(q4 + q6);
        out[6] = R14(q5 + q7);
        out[3] = -R14(((u32)s2 + (u32)s3) * 11585u);
        out[4] = R14(((u32)s2 - (u32)s3) * 11585u);
        out[2] = R14(((u32)s6 + (u32)s7) * 11585u);
        out[5] = -R14(((u32)s6 - (u32)s7) * 11585u);
    }
}

static void iadst16(const int32_t *x, int32_t *out)
{
    static const u32 c[8][2] = { { 16364, 804 }, { 15893, 3981 }, { 14811, 7005 }, { 13160, 9760 },
                                 { 11003, 12140 }, { 8423, 14053 }, { 5520, 15426 }, { 2404, 16207 } };
    u32 p[16], q[16];
    int32_t a[16], t[16];
    for (int k = 0; k < 8; k++) { /* pairs (x[15-2k], x[2k]) */
        const u32 u = x[15 - 2 * k], v = x[2 * k];
        p[2 * k] = c[k][0] * u + c[k][1] * v;
        p[2 * k + 1] = c[k][1] * u - c[k][0] * v;
    }
    for (int k = 0; k < 8; k++) {
        a[k] = R14(p[k] + p[k + 8]);
        a[k + 8] = R14(p[k] - p[k + 8]);
    }
    q[8] = (u32)a[8] * 16069u + (u32)a[9] * 3196u;
    q[9] = (u32)a[8] * 3196u - (u32)a[9] * 16069u;
    q[10] = (u32)a[10] * 9102u + (u32)a[11] * 13623u;
    q[11] = (u32)a[10] * 13623u - (u32)a[11] * 9102u;
    q[12] = (u32)a[13] * 16069u - (u32)a[12] * 3196u;
    q[13] = (u32)a[13] * 3196u + (u32)a[12] * 16069u;
    q[14] = (u32)a[15] * 9102u - (u32)a[14] * 13623u;
    q[15] = (u32)a[15] * 13623u + (u32)a[14] * 9102u;
    for (int k = 0; k < 4; k++) {
        t[k] = a[k] + a[k + 4];
        t[k + 4] = a[k] - a[k + 4];
        a[k + 8] = R14(q[k + 8] + q[k + 12]);
        a[k + 12] = R14(q[k + 8] - q[k + 12]);
    }
    {
        const u32 r4 = (u32)t[4] * 15137u + (u32)t[5] * 6270u, r5 = (u32)t[4] * 6270u - (u32)t[5] * 15137u;
        const u32 r6 = (u32)t[7] * 15137u - (u32)t[6] * 6270u, r7 = (u32)t[7] * 6270u + (u32)t[6] * 15137u;
        const u32 r12 = (u32)a[12] * 15137u + (u32)a[13] * 6270u, r13 = (u32)a[12] * 6270u - (u32)a[13] * 15137u;
        const u32 r14 = (u32)a[15] * 15137u - (u32)a[14] * 6270u, r15 = (u32)a[15] * 6270u + (u32)a[14] * 15137u;
        const int32_t s2 = t[0] - t[2], s3 = t[1] - t[3];
        const int32_t s6 = R14(r4 - r6), s7 = R14(r5 - r7);
        const int32_t s10 = a[8] - a[10], s11 = a[9] - a[11];
        const int32_t s14 = R14(r12 - r14), s15 = R14(r13 - r15);
        out[0] = t[0] + t[2];
        out[15] = -(t[1] + t[3]);
        out[3] = -R14(r4 + r6);
        out[12] = R14(r5 + r7);
        out[1] = -(a[8] + a[10]);
        out[14] = a[9] + a[11];
        out[2] = R14(r12 + r14);
        out[13] = -R14(r13 + r15);
        out[7] = R14(-((u32)s2 + (u32)s3) * 11585u);
        out[8] = R14(((u32)s2 - (u32)s3) * 11585u);
        out[4] = R14(((u32)s7 + (u32)s6) * 11585u);
        out[11] = R14(((u32)s7 - (u32)s6) * 11585u);
        out[6] = R14(((u32)s11 + (u32)s10) * 11585u);
        out[9] = R14(((u32)s11 - (u32)s10) * 11585u);
        out[5] = R14(-((u32)s14 + (u32)s15) * 11585u);
        out[10] = R14(((u32)s14 - (u32)s15) * 11585u);
    }
}

/* lossless mode: the Walsh-Hadamard transform (:1719-1750); pass 0 scales its inputs down by 4 */
static void iwht4(const int32_t *x, int32_t *out, int pass)
{
    int32_t t0 = x[0], t1 = x[3], t2 = x[1], t3 = x[2], t4;
    if (!pass) {
        t0 >>= 2; t1 >>= 2; t2 >>= 2; t3 >>= 2;
    }
    t0 += t2;
    t3 -= t1;
    t4 = (t0 - t3) >> 1;
    t1 = t4 - t1;
    t2 = t4 - t2;
    t0 -= t1;
    t3 += t2;
    out[0] = t0; out[1] = t1; out[2] = t2; out[3] = t3;
}

static void tx1d(int kind, int n, const int32_t *x, int32_t *out, int pass)
{
    if (kind == 2)
        iwht4(x, out, pass);
    else if (kind == 0)
        idct_n(n, x, out);
    else if (n == 4)
        iadst4(x, out);
    else if (n == 8)
        iadst8(x, out);
    else
        iadst16(x, out);
}

static uint8_t clip_px(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

/*
 * itxfm_add[tx][txtp](dst, stride, block, eob): tx 0..3 = 4x4 .. 32x32, 4 = lossless 4x4 WHT; txtp 0 DCT_DCT, 1 DCT_ADST,
 * 2 ADST_DCT, 3 ADST_ADST (libavcodec/vp9.h: enum TxfmType) — DCT_ADST runs the ADST in the FIRST pass (iadst_idct_*,
 * vp9dsp_template.c:1756-1776); 32x32 and the WHT have one function in all four slots.  The block is consumed (zeroed;
 * the dc-only shortcut of DCT_DCT with eob == 1 clears block[0] only).
 */
void ffo_vp9_itxfm_add(int tx, int txtp, uint8_t *dst, ptrdiff_t stride, int16_t *block, int eob)
{
    const int wht = tx == 4, n = wht ? 4 : 4 << tx, bits = wht ? 0 : tx == 0 ? 4 : tx == 1 ? 5 : 6;
    const int first = wht ? 2 : (tx == 3 ? 0 : (txtp == 1 || txtp == 3)), second = wht ? 2 : (tx == 3 ? 0 : (txtp == 2 || txtp == 3));
    int16_t tmp[32 * 32];
    if (!wht && !first && !second && eob == 1) {
        const int32_t t = R14((u32)R14((u32)block[0] * 11585u) * 11585u);
        block[0] = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                dst[j * stride + i] = clip_px(dst[j * stride + i] + (bits ? (int32_t)((u32)t + (1u << (bits - 1))) >> bits : t));
        return;
    }
    for (int i = 0; i < n; i++) {
        int32_t x[32], o[32];
        for (int k = 0; k < n; k++)
            x[k] = block[i + k * n];
        tx1d(first, n, x, o, 0);
        for (int k = 0; k < n; k++)
            tmp[i * n + k] = (int16_t)o[k];
    }
    memset(block, 0, sizeof(int16_t) * n * n);
    for (int i = 0; i < n; i++) {
        int32_t x[32], o[32];
        for (int k = 0; k < n; k++)
            x[k] = tmp[i + k * n];
        tx1d(second, n, x, o, 1);
        for (int j = 0; j < n; j++) {
            const int32_t v = (int16_t)o[j]; /* out[] is dctcoef too */
            dst[j * stride + i] = clip_px(dst[j * stride + i] + (bits ? (int32_t)((u32)v + (1u << (bits - 1))) >> bits : v));
        }
    }
}

/*
 * VP9 motion compensation, 8 bits: VP9DSPContext.mc[size][filter][avg][!!mx][!!my] (libavcodec/vp9dsp_template.c:1966-2293;
 * taps: ff_vp9_subpel_filters, libavcodec/vp9dsp.c:32-86; enum FilterMode, libavcodec/vp9.h:64-70: 0 smooth, 1 regular,
 * 2 sharp, 3 bilinear).  Each 8-tap pass is clip_u8((sum + 64) >> 7); the 2-D form filters rows -3..h+3 horizontally into 8-bit
 * temporaries first.  Bilinear: a + ((m (b - a) + 8) >> 4), rows 0..h.  avg: (dst + v + 1) >> 1.  Stated per output sample.
 */
static const int8_t vp9_taps[3][16][8] = {
    { { 0, 0, 0, 127, 0, 0, 0, 0 }, /* index 0 is never used (full-pel copies); 128 does not fit, see vp9_tap() */
      { -3, -1, 32, 64, 38, 1, -3, 0 }, { -2, -2, 29, 63, 41, 2, -3, 0 }, { -2, -2, 26, 63, 43, 4, -4, 0 }, { -2, -3, 24, 62, 46, 5, -4, 0 },
      { -2, -3, 21, 60, 49, 7, -4, 0 }, { -1, -4, 18, 59, 51, 9, -4, 0 }, { -1, -4, 16, 57, 53, 12, -4, -1 }, { -1, -4, 14, 55, 55, 14, -4, -1 },
      { -1, -4, 12, 53, 57, 16, -4, -1 }, { 0, -4, 9, 51, 59, 18, -4, -1 }, { 0, -4, 7, 49, 60, 21, -3, -2 }, { 0, -4, 5, 46, 62, 24, -3, -2 },
      { 0, -4, 4, 43, 63, 26, -2, -2 }, { 0, -3, 2, 41, 63, 29, -2, -2 }, { 0, -3, 1, 38, 64, 32, -1, -3 } },
    { { 0, 0, 0, 127, 0, 0, 0, 0 },
      { 0, 1, -5, 126, 8, -3, 1, 0 }, { -1, 3, -10, 122, 18, -6, 2, 0 }, { -1, 4, -13, 118, 27, -9, 3, -1 }, { -1, 4, -16, 112, 37, -11, 4, -1 },
      { -1, 5, -18, 105, 48, -14, 4, -1 }, { -1, 5, -19, 97, 58, -16, 5, -1 }, { -1, 6, -19, 88, 68, -18, 5, -1 }, { -1, 6, -19, 78, 78, -19, 6, -1 },
      { -1, 5, -18, 68, 88, -19, 6, -1 }, { -1, 5, -16, 58, 97, -19, 5, -1 }, { -1, 4, -14, 48, 105, -18, 5, -1 }, { -1, 4, -11, 37, 112, -16, 4, -1 },
      { -1, 3, -9, 27, 118, -13, 4, -1 }, { 0, 2, -6, 18, 122, -10, 3, -1 }, { 0, 1, -3, 8, 126, -5, 1, 0 } },
    { { 0, 0, 0, 127, 0, 0, 0, 0 },
      { -1, 3, -7, 127, 8, -3, 1, 0 }, { -2, 5, -13, 125, 17, -6, 3, -1 }, { -3, 7, -17, 121, 27, -10, 5, -2 }, { -4, 9, -20, 115, 37, -13, 6, -2 },
      { -4, 10, -23, 108, 48, -16, 8, -3 }, { -4, 10, -24, 100, 59, -19, 9, -3 }, { -4, 11, -24, 90, 70, -21, 10, -4 }, { -4, 11, -23, 80, 80, -23, 11, -4 },
      { -4, 10, -21, 70, 90, -24, 11, -4 }, { -3, 9, -19, 59, 100, -24, 10, -4 }, { -3, 8, -16, 48, 108, -23, 10, -4 }, { -2, 6, -13, 37, 115, -20, 9, -4 },
      { -2, 5, -10, 27, 121, -17, 7, -3 }, { -1, 3, -6, 17, 125, -13, 5, -2 }, { 0, 1, -3, 8, 127, -7, 3, -1 } },
};

static int vp9_tap8(int filter, int m, const uint8_t *s, ptrdiff_t step)
{
    int sum = 64;
    for (int k = 0; k < 8; k++)
        sum += vp9_taps[filter][m][k] * s[(k - 3) * step];
    sum >>= 7;
    return sum < 0 ? 0 : sum > 255 ? 255 : sum;
}
static int vp9_bilin(int m, const uint8_t *s, ptrdiff_t step) { return s[0] + ((m * (s[step] - s[0]) + 8) >> 4); }

/* width 4..64 (multiple of 4), height 1..64, filter 0..3, mx / my 0..15 */
void ffo_vp9_mc(int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width, int height,
                int mx, int my)
{
    uint8_t tmp[71 * 64];
    const int bil = filter == 3;
    if (mx && my) { /* rows -3..h+3 (bilinear: 0..h) through the horizontal filter */
        const int r0 = bil ? 0 : -3, rows = bil ? height + 1 : height + 7;
        for (int r = 0; r < rows; r++)
            for (int x = 0; x < width; x++) {
                const uint8_t *s = src + (r + r0) * srcstride + x;
                tmp[r * 64 + x] = bil ? vp9_bilin(mx, s, 1) : vp9_tap8(filter, mx, s, 1);
            }
    }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const uint8_t *s = src + y * srcstride + x;
            int v;
            if (mx && my)
                v = bil ? vp9_bilin(my, tmp + y * 64 + x, 64) : vp9_tap8(filter, my, tmp + (y + 3) * 64 + x, 64);
            else if (mx)
                v = bil ? vp9_bilin(mx, s, 1) : vp9_tap8(filter, mx, s, 1);
            else if (my)
                v = bil ? vp9_bilin(my, s, srcstride) : vp9_tap8(filter, my, s, srcstride);
            else
                v = s[0];
            dst[y * dststride + x] = avg ? (dst[y * dststride + x] + v + 1) >> 1 : v;
        }
}

/*
 * VP9 loop filter, 8 bits: one 8-sample segment of an edge, loop_filter() (libavcodec/vp9dsp_template.c:1780-1889) as
 * loop_filter_8[wd][dir], loop_filter_16[dir] (two segments) and loop_filter_mix2[wd1][wd2][dir] (two segments, the limits
 * packed in the two low bytes) call it (:1891-1966).  dir 0 = "h": a column edge, the segment runs down (next line = + stride,
 * across = 1); dir 1 = "v": a row edge.  wd = 4, 8 or 16.
 * The two flat filters are stated as what they are: a window of radius 3 / 7 around the sample over the 8 / 16 samples
 * p3..q3 / p7..q7 with the ends repeated, the centre counted twice.
 */
static int iabs(int v) { return v < 0 ? -v : v; }
static int clip_i8(int v) { return v < -128 ? -128 : v > 127 ? 127 : v; }

void ffo_vp9_loop_filter(int wd, int dir, uint8_t *dst, ptrdiff_t stride, int E, int I, int H)
{
    const ptrdiff_t along = dir ? 1 : stride, across = dir ? stride : 1;
    for (int i = 0; i < 8; i++, dst += along) {
        int px[16]; /* p7 .. p0, q0 .. q7 */
        const int r = wd >= 16 ? 8 : 4;
        for (int k = -r; k < r; k++)
            px[8 + k] = dst[k * across];
        const int p3 = px[4], p2 = px[5], p1 = px[6], p0 = px[7], q0 = px[8], q1 = px[9], q2 = px[10], q3 = px[11];
        if (!(iabs(p3 - p2) <= I && iabs(p2 - p1) <= I && iabs(p1 - p0) <= I && iabs(q1 - q0) <= I && iabs(q2 - q1) <= I &&
              iabs(q3 - q2) <= I && iabs(p0 - q0) * 2 + (iabs(p1 - q1) >> 1) <= E))
            continue;
        int flat_in = wd >= 8, flat_out = wd >= 16;
        for (int k = 1; k <= 3 && flat_in; k++)
            flat_in = iabs(px[7 - k] - p0) <= 1 && iabs(px[8 + k] - q0) <= 1;
        for (int k = 4; k <= 7 && flat_out; k++)
            flat_out = iabs(px[7 - k] - p0) <= 1 && iabs(px[8 + k] - q0) <= 1;
        if (flat_out && flat_in) {
            for (int c = 1; c <= 14; c++) {
                int s = px[c] + 8;
                for (int t = -7; t <= 7; t++)
                    s += px[c + t < 0 ? 0 : c + t > 15 ? 15 : c + t];
                dst[(c - 8) * across] = s >> 4;
            }
        } else if (flat_in) {
            for (int c = 5; c <= 10; c++) {
                int s = px[c] + 4;
                for (int t = -3; t <= 3; t++)
                    s += px[c + t < 4 ? 4 : c + t > 11 ? 11 : c + t];
                dst[(c - 8) * across] = s >> 3;
            }
        } else {
            const int hev = iabs(p1 - p0) > H || iabs(q1 - q0) > H;
            int f = clip_i8(3 * (q0 - p0) + (hev ? clip_i8(p1 - q1) : 0));
            const int f1 = (f + 4 > 127 ? 127 : f + 4) >> 3, f2 = (f + 3 > 127 ? 127 : f + 3) >> 3;
            dst[-across] = clip_px(p0 + f2);
            dst[0] = clip_px(q0 - f1);
            if (!hev) {
                f = (f1 + 1) >> 1;
                dst[-2 * across] = clip_px(p1 + f);
                dst[across] = clip_px(q1 - f);
            }
        }
    }
}

/*
 * VP9 intra prediction, 8 bits: VP9DSPContext.intra_pred[tx][mode](dst, stride, left, top)
 * (libavcodec/vp9dsp_template.c:33-1153; enum IntraPredMode, libavcodec/vp9.h:45-62).  left[] runs bottom to top (left[N-1] is
 * beside row 0), top[-1] is the corner.  The reference writes every size of every mode out; here each mode is its per-sample rule
 * over the "edge line" e[] = left[0..N-1], corner, top[0..]: e[k] walks up the left column, round the corner and along the top.
 */
static int A2(int a, int b) { return (a + b + 1) >> 1; }
static int A3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }

void ffo_vp9_intra_pred(int tx, int mode, uint8_t *dst, ptrdiff_t stride, const uint8_t *left, const uint8_t *top)
{
    const int n = 4 << tx, lg = 2 + tx;
    int e[32 + 1 + 64]; /* edge line; the 4x4 down-left / vert-left modes read 8 top samples, nobody reads more than 2n */
    int dc = 0;
    const int uses_top = mode == 0 || mode == 2 || mode == 3 || mode == 4 || mode == 5 || mode == 6 || mode == 7 || mode == 9 || mode == 11;
    const int uses_left = mode == 1 || mode == 2 || mode == 4 || mode == 5 || mode == 6 || mode == 8 || mode == 9 || mode == 10;
    const int ntop = (tx == 0 && (mode == 3 || mode == 7)) ? 8 : n;
    memset(e, 0, sizeof(e));
    if (uses_left)
        for (int k = 0; k < n; k++)
            e[k] = left[k];
    if (mode == 4 || mode == 5 || mode == 6 || mode == 9)
        e[n] = top[-1];
    if (uses_top)
        for (int k = 0; k < ntop; k++)
            e[n + 1 + k] = top[k];
    const int *T = e + n + 1; /* T[-1] = corner */
    if (mode == 2) {
        for (int k = 0; k < n; k++)
            dc += e[k] + T[k];
        dc = (dc + n) >> (lg + 1);
    } else if (mode == 10 || mode == 11) {
        for (int k = 0; k < n; k++)
            dc += mode == 10 ? e[k] : T[k];
        dc = (dc + n / 2) >> lg;
    } else if (mode >= 12) {
        dc = mode == 12 ? 128 : mode == 13 ? 127 : 129;
    }
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            int v;
            switch (mode) {
            case 0: v = T[x]; break;                                                     /* VERT */
            case 1: v = e[n - 1 - y]; break;                                             /* HOR */
            case 3: {                                                                    /* DIAG_DOWN_LEFT */
                const int i = x + y;
                if (tx == 0)
                    v = i < 6 ? A3(T[i], T[i + 1], T[i + 2]) : T[7];
                else
                    v = i < n - 2 ? A3(T[i], T[i + 1], T[i + 2]) : i == n - 2 ? (T[n - 2] + 3 * T[n - 1] + 2) >> 2 : T[n - 1];
                break;
            }
            case 4: {                                                                    /* DIAG_DOWN_RIGHT: along the edge line */
                const int i = n - 1 - y + x;
                v = A3(e[i], e[i + 1], e[i + 2]);
                break;
            }
            case 5: {                                                                    /* VERT_RIGHT */
                const int i = n / 2 - 1 - (y >> 1) + x;
                if (i >= n / 2 - 1) {
                    const int k = n + i - (n / 2 - 1);
                    v = (y & 1) ? A3(e[k - 1], e[k], e[k + 1]) : A2(e[k], e[k + 1]);
                } else {
                    v = (y & 1) ? A3(e[2 * i + 3], e[2 * i + 2], e[2 * i + 1]) : A3(e[2 * i + 4], e[2 * i + 3], e[2 * i + 2]);
                }
                break;
            }
            case 6: {                                                                    /* HOR_DOWN */
                const int i = 2 * n - 2 - 2 * y + x;
                if (i >= 2 * n)
                    v = A3(e[i - n], e[i - n + 1], e[i - n + 2]);
                else
                    v = (i & 1) ? A3(e[(i >> 1) + 2], e[(i >> 1) + 1], e[i >> 1]) : A2(e[(i >> 1) + 1], e[i >> 1]);
                break;
            }
            case 7: {                                                                    /* VERT_LEFT */
                const int i = (y >> 1) + x;
                if (tx == 0)
                    v = (y & 1) ? A3(T[i], T[i + 1], T[i + 2]) : A2(T[i], T[i + 1]);
                else if (i >= n - 1)
                    v = T[n - 1];
                else if (y & 1)
                    v = i < n - 2 ? A3(T[i], T[i + 1], T[i + 2]) : (T[n - 2] + 3 * T[n - 1] + 2) >> 2;
                else
                    v = A2(T[i], T[i + 1]);
                break;
            }
            case 8: {                                                                    /* HOR_UP: left[] counted from index 0 */
                const int i = 2 * y + x;
                if (i >= 2 * n - 2)
                    v = e[n - 1];
                else if (i == 2 * n - 3)
                    v = (e[n - 2] + 3 * e[n - 1] + 2) >> 2;
                else
                    v = (i & 1) ? A3(e[i >> 1], e[(i >> 1) + 1], e[(i >> 1) + 2]) : A2(e[i >> 1], e[(i >> 1) + 1]);
                break;
            }
            case 9: v = clip_px(T[x] + e[n - 1 - y] - T[-1]); break;                     /* TM */
            default: v = dc; break;                                                      /* DC, LEFT_DC, TOP_DC, DC_128/127/129 */
            }
            dst[y * stride + x] = (uint8_t)v;
        }
}

/*
 * VP9 scaled motion compensation, 8 bits: VP9DSPContext.smc[size][filter][avg](dst, dst_stride, ref, ref_stride, h, mx, my, dx, dy)
 * (libavcodec/vp9dsp_template.c:2362-2540): the reference picture has another size, so the sampling position advances by
 * dx / dy sixteenths per output sample: output x reads around column (mx + x dx) >> 4 with the taps of fraction (mx + x dx) & 15,
 * output y around row (my + y dy) >> 4; horizontally filtered 8-bit temporaries first, exactly as the unscaled 2-D form.
 * Fraction 0 is the tap set { 0, 0, 0, 128, ... }: (128 s + 64) >> 7 = s.
 */
static int vp9_tap8s(int filter, int m, const uint8_t *s, ptrdiff_t step)
{
    return m ? vp9_tap8(filter, m, s, step) : s[0];
}

void ffo_vp9_smc(int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width, int height,
                 int mx, int my, int dx, int dy)
{
    static uint8_t tmp[135 * 64];
    const int bil = filter == 3, before = bil ? 0 : 3;
    const int rows = (((height - 1) * dy + my) >> 4) + (bil ? 2 : 8);
    for (int r = 0; r < rows; r++)
        for (int x = 0; x < width; x++) {
            const int pos = mx + x * dx;
            const uint8_t *s = src + (r - before) * srcstride + (pos >> 4);
            tmp[r * 64 + x] = bil ? vp9_bilin(pos & 15, s, 1) : vp9_tap8s(filter, pos & 15, s, 1);
        }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int pos = my + y * dy;
            const uint8_t *t = tmp + ((pos >> 4) + before) * 64 + x;
            const int v = bil ? vp9_bilin(pos & 15, t, 64) : vp9_tap8s(filter, pos & 15, t, 64);
            dst[y * dststride + x] = avg ? (dst[y * dststride + x] + v + 1) >> 1 : v;
        }
}
