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

#define ST int32_t
#define UT uint32_t
#define FN(x) x
#include "ffo_vp9_tx.inc"
#undef ST
#undef UT
#undef FN
#define ST int64_t
#define UT uint64_t
#define FN(x) x##_w64
#include "ffo_vp9_tx.inc"
#undef ST
#undef UT
#undef FN
typedef uint32_t u32;
#define R14(x) ((int32_t)((u32)(x) + (1u << 13)) >> 14)

static uint8_t clip_px(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
/* pixel access by bit depth (bit_depth_template.c): uint8_t at 8 bits, uint16_t above; i counts SAMPLES from a byte pointer */
static inline int pget(const uint8_t *p, ptrdiff_t i, int bd) { return bd > 8 ? ((const uint16_t *)p)[i] : p[i]; }
static inline void pput(uint8_t *p, ptrdiff_t i, int v, int bd)
{
    if (bd > 8)
        ((uint16_t *)p)[i] = (uint16_t)v;
    else
        p[i] = (uint8_t)v;
}
static int clipp(int v, int bd) { const int m = (1 << bd) - 1; return v < 0 ? 0 : v > m ? m : v; }
static ptrdiff_t spx(ptrdiff_t stride_bytes, int bd) { return bd > 8 ? stride_bytes / 2 : stride_bytes; }

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

/* the same members above 8 bits: dctcoef = int32_t (the block argument points at int32 coefficients), dctint = int64_t, pixels
 * uint16_t clipped to (1 << bd) - 1, stride in bytes (vp9dsp_template.c:1155-1195 with BIT_DEPTH 10 / 12) */
void ffo_vp9_itxfm_add_bd(int bd, int tx, int txtp, uint8_t *dst, ptrdiff_t stride, int32_t *block, int eob)
{
    const int wht = tx == 4, n = wht ? 4 : 4 << tx, bits = wht ? 0 : tx == 0 ? 4 : tx == 1 ? 5 : 6;
    const int first = wht ? 2 : (tx == 3 ? 0 : (txtp == 1 || txtp == 3)), second = wht ? 2 : (tx == 3 ? 0 : (txtp == 2 || txtp == 3));
    int32_t tmp[32 * 32];
    stride = spx(stride, bd);
    if (!wht && !first && !second && eob == 1) {
        const int t = (int)((((((int64_t)block[0] * 11585 + (1 << 13)) >> 14) * 11585) + (1 << 13)) >> 14);
        block[0] = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                pput(dst, j * stride + i, clipp(pget(dst, j * stride + i, bd) + (bits ? (int)((unsigned)t + (1u << (bits - 1))) >> bits : t), bd), bd);
        return;
    }
    for (int i = 0; i < n; i++) {
        int64_t x[32], o[32];
        for (int k = 0; k < n; k++)
            x[k] = block[i + k * n];
        tx1d_w64(first, n, x, o, 0);
        for (int k = 0; k < n; k++)
            tmp[i * n + k] = (int32_t)o[k];
    }
    memset(block, 0, sizeof(int32_t) * n * n);
    for (int i = 0; i < n; i++) {
        int64_t x[32], o[32];
        for (int k = 0; k < n; k++)
            x[k] = tmp[i + k * n];
        tx1d_w64(second, n, x, o, 1);
        for (int j = 0; j < n; j++) {
            const int32_t v = (int32_t)o[j]; /* out[] is dctcoef too */
            pput(dst, j * stride + i, clipp(pget(dst, j * stride + i, bd) + (bits ? (int)((unsigned)v + (1u << (bits - 1))) >> bits : v), bd), bd);
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

/* s + i: sample i of a plane of depth bd (byte pointer, sample index) */
static int vp9_tap8(int bd, int filter, int m, const uint8_t *s, ptrdiff_t i, ptrdiff_t step)
{
    int sum = 64;
    for (int k = 0; k < 8; k++)
        sum += vp9_taps[filter][m][k] * pget(s, i + (k - 3) * step, bd);
    return clipp(sum >> 7, bd);
}
static int vp9_bilin(int bd, int m, const uint8_t *s, ptrdiff_t i, ptrdiff_t step)
{
    return pget(s, i, bd) + ((m * (pget(s, i + step, bd) - pget(s, i, bd)) + 8) >> 4);
}

/* width 4..64 (multiple of 4), height 1..64, filter 0..3, mx / my 0..15; above 8 bits the temporaries are pixels of that depth */
void ffo_vp9_mc_bd(int bd, int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width,
                   int height, int mx, int my)
{
    uint16_t tmp16[71 * 64];
    uint8_t tmp8[71 * 64];
    uint8_t *tmp = bd > 8 ? (uint8_t *)tmp16 : tmp8;
    const int bil = filter == 3;
    srcstride = spx(srcstride, bd);
    dststride = spx(dststride, bd);
    if (mx && my) { /* rows -3..h+3 (bilinear: 0..h) through the horizontal filter */
        const int r0 = bil ? 0 : -3, rows = bil ? height + 1 : height + 7;
        for (int r = 0; r < rows; r++)
            for (int x = 0; x < width; x++) {
                const ptrdiff_t at = (r + r0) * srcstride + x;
                pput(tmp, r * 64 + x, bil ? vp9_bilin(bd, mx, src, at, 1) : vp9_tap8(bd, filter, mx, src, at, 1), bd);
            }
    }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const ptrdiff_t at = y * srcstride + x;
            int v;
            if (mx && my)
                v = bil ? vp9_bilin(bd, my, tmp, y * 64 + x, 64) : vp9_tap8(bd, filter, my, tmp, (y + 3) * 64 + x, 64);
            else if (mx)
                v = bil ? vp9_bilin(bd, mx, src, at, 1) : vp9_tap8(bd, filter, mx, src, at, 1);
            else if (my)
                v = bil ? vp9_bilin(bd, my, src, at, srcstride) : vp9_tap8(bd, filter, my, src, at, srcstride);
            else
                v = pget(src, at, bd);
            pput(dst, y * dststride + x, avg ? (pget(dst, y * dststride + x, bd) + v + 1) >> 1 : v, bd);
        }
}
void ffo_vp9_mc(int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width, int height,
                int mx, int my)
{
    ffo_vp9_mc_bd(8, filter, avg, dst, dststride, src, srcstride, width, height, mx, my);
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

static int clip_sp(int v, int bits) { const int lo = -(1 << bits), hi = (1 << bits) - 1; return v < lo ? lo : v > hi ? hi : v; } /* av_clip_intp2 */

/* E, I, H arrive in 8-bit units and are scaled by << (bd - 8), the flatness threshold is 1 << (bd - 8) (:1784-1788) */
void ffo_vp9_loop_filter_bd(int bd, int wd, int dir, uint8_t *dst_, ptrdiff_t stride, int E, int I, int H)
{
    const ptrdiff_t st = spx(stride, bd), along = dir ? 1 : st, across = dir ? st : 1;
    const int F = 1 << (bd - 8), fmax = (1 << (bd - 1)) - 1;
    ptrdiff_t dst = 0;
    E <<= bd - 8;
    I <<= bd - 8;
    H <<= bd - 8;
    for (int i = 0; i < 8; i++, dst += along) {
        int px[16]; /* p7 .. p0, q0 .. q7 */
        const int r = wd >= 16 ? 8 : 4;
        for (int k = -r; k < r; k++)
            px[8 + k] = pget(dst_, dst + k * across, bd);
        const int p3 = px[4], p2 = px[5], p1 = px[6], p0 = px[7], q0 = px[8], q1 = px[9], q2 = px[10], q3 = px[11];
        if (!(iabs(p3 - p2) <= I && iabs(p2 - p1) <= I && iabs(p1 - p0) <= I && iabs(q1 - q0) <= I && iabs(q2 - q1) <= I &&
              iabs(q3 - q2) <= I && iabs(p0 - q0) * 2 + (iabs(p1 - q1) >> 1) <= E))
            continue;
        int flat_in = wd >= 8, flat_out = wd >= 16;
        for (int k = 1; k <= 3 && flat_in; k++)
            flat_in = iabs(px[7 - k] - p0) <= F && iabs(px[8 + k] - q0) <= F;
        for (int k = 4; k <= 7 && flat_out; k++)
            flat_out = iabs(px[7 - k] - p0) <= F && iabs(px[8 + k] - q0) <= F;
        if (flat_out && flat_in) {
            for (int c = 1; c <= 14; c++) {
                int s = px[c] + 8;
                for (int t = -7; t <= 7; t++)
                    s += px[c + t < 0 ? 0 : c + t > 15 ? 15 : c + t];
                pput(dst_, dst + (c - 8) * across, s >> 4, bd);
            }
        } else if (flat_in) {
            for (int c = 5; c <= 10; c++) {
                int s = px[c] + 4;
                for (int t = -3; t <= 3; t++)
                    s += px[c + t < 4 ? 4 : c + t > 11 ? 11 : c + t];
                pput(dst_, dst + (c - 8) * across, s >> 3, bd);
            }
        } else {
            const int hev = iabs(p1 - p0) > H || iabs(q1 - q0) > H;
            int f = clip_sp(3 * (q0 - p0) + (hev ? clip_sp(p1 - q1, bd - 1) : 0), bd - 1);
            const int f1 = (f + 4 > fmax ? fmax : f + 4) >> 3, f2 = (f + 3 > fmax ? fmax : f + 3) >> 3;
            pput(dst_, dst - across, clipp(p0 + f2, bd), bd);
            pput(dst_, dst, clipp(q0 - f1, bd), bd);
            if (!hev) {
                f = (f1 + 1) >> 1;
                pput(dst_, dst - 2 * across, clipp(p1 + f, bd), bd);
                pput(dst_, dst + across, clipp(q1 - f, bd), bd);
            }
        }
    }
}
void ffo_vp9_loop_filter(int wd, int dir, uint8_t *dst, ptrdiff_t stride, int E, int I, int H)
{
    ffo_vp9_loop_filter_bd(8, wd, dir, dst, stride, E, I, H);
}

/*
 * VP9 intra prediction, 8 bits: VP9DSPContext.intra_pred[tx][mode](dst, stride, left, top)
 * (libavcodec/vp9dsp_template.c:33-1153; enum IntraPredMode, libavcodec/vp9.h:45-62).  left[] runs bottom to top (left[N-1] is
 * beside row 0), top[-1] is the corner.  The reference writes every size of every mode out; here each mode is its per-sample rule
 * over the "edge line" e[] = left[0..N-1], corner, top[0..]: e[k] walks up the left column, round the corner and along the top.
 */
static int A2(int a, int b) { return (a + b + 1) >> 1; }
static int A3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }

void ffo_vp9_intra_pred_bd(int bd, int tx, int mode, uint8_t *dst, ptrdiff_t stride, const uint8_t *left, const uint8_t *top)
{
    stride = spx(stride, bd);
    const int n = 4 << tx, lg = 2 + tx;
    int e[32 + 1 + 64]; /* edge line; the 4x4 down-left / vert-left modes read 8 top samples, nobody reads more than 2n */
    int dc = 0;
    const int uses_top = mode == 0 || mode == 2 || mode == 3 || mode == 4 || mode == 5 || mode == 6 || mode == 7 || mode == 9 || mode == 11;
    const int uses_left = mode == 1 || mode == 2 || mode == 4 || mode == 5 || mode == 6 || mode == 8 || mode == 9 || mode == 10;
    const int ntop = (tx == 0 && (mode == 3 || mode == 7)) ? 8 : n;
    memset(e, 0, sizeof(e));
    if (uses_left)
        for (int k = 0; k < n; k++)
            e[k] = pget(left, k, bd);
    if (mode == 4 || mode == 5 || mode == 6 || mode == 9)
        e[n] = pget(top, -1, bd);
    if (uses_top)
        for (int k = 0; k < ntop; k++)
            e[n + 1 + k] = pget(top, k, bd);
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
        dc = (128 << (bd - 8)) + (mode == 12 ? 0 : mode == 13 ? -1 : 1);
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
            case 9: v = clipp(T[x] + e[n - 1 - y] - T[-1], bd); break;                     /* TM */
            default: v = dc; break;                                                      /* DC, LEFT_DC, TOP_DC, DC_128/127/129 */
            }
            pput(dst, y * stride + x, v, bd);
        }
}
void ffo_vp9_intra_pred(int tx, int mode, uint8_t *dst, ptrdiff_t stride, const uint8_t *left, const uint8_t *top)
{
    ffo_vp9_intra_pred_bd(8, tx, mode, dst, stride, left, top);
}

/*
 * VP9 scaled motion compensation, 8 bits: VP9DSPContext.smc[size][filter][avg](dst, dst_stride, ref, ref_stride, h, mx, my, dx, dy)
 * (libavcodec/vp9dsp_template.c:2362-2540): the reference picture has another size, so the sampling position advances by
 * dx / dy sixteenths per output sample: output x reads around column (mx + x dx) >> 4 with the taps of fraction (mx + x dx) & 15,
 * output y around row (my + y dy) >> 4; horizontally filtered 8-bit temporaries first, exactly as the unscaled 2-D form.
 * Fraction 0 is the tap set { 0, 0, 0, 128, ... }: (128 s + 64) >> 7 = s.
 */
static int vp9_tap8s(int bd, int filter, int m, const uint8_t *s, ptrdiff_t i, ptrdiff_t step)
{
    return m ? vp9_tap8(bd, filter, m, s, i, step) : pget(s, i, bd);
}

void ffo_vp9_smc_bd(int bd, int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width,
                    int height, int mx, int my, int dx, int dy)
{
    static uint16_t tmp16[135 * 64];
    static uint8_t tmp8[135 * 64];
    uint8_t *tmp = bd > 8 ? (uint8_t *)tmp16 : tmp8;
    const int bil = filter == 3, before = bil ? 0 : 3;
    const int rows = (((height - 1) * dy + my) >> 4) + (bil ? 2 : 8);
    srcstride = spx(srcstride, bd);
    dststride = spx(dststride, bd);
    for (int r = 0; r < rows; r++)
        for (int x = 0; x < width; x++) {
            const int pos = mx + x * dx;
            const ptrdiff_t at = (r - before) * srcstride + (pos >> 4);
            pput(tmp, r * 64 + x, bil ? vp9_bilin(bd, pos & 15, src, at, 1) : vp9_tap8s(bd, filter, pos & 15, src, at, 1), bd);
        }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int pos = my + y * dy;
            const ptrdiff_t at = ((pos >> 4) + before) * 64 + x;
            const int v = bil ? vp9_bilin(bd, pos & 15, tmp, at, 64) : vp9_tap8s(bd, filter, pos & 15, tmp, at, 64);
            pput(dst, y * dststride + x, avg ? (pget(dst, y * dststride + x, bd) + v + 1) >> 1 : v, bd);
        }
}
void ffo_vp9_smc(int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width, int height,
                 int mx, int my, int dx, int dy)
{
    ffo_vp9_smc_bd(8, filter, avg, dst, dststride, src, srcstride, width, height, mx, my, dx, dy);
}

/* ------------------------------------------------------------------------------------------
 * ff_vp9_loopfilter_sb() (libavcodec/vp9lpf.c:27-203): the loop filter of ONE 64x64 superblock in the decoder's order — per
 * plane all column edges (filter_plane_cols, :27-99), then all row edges (filter_plane_rows, :101-178) — restated call for call:
 * every dsp call of the reference (loop_filter_16 / loop_filter_8[wd] / loop_filter_mix2[wd1][wd2]) is the one or two 8-sample
 * segments it stands for, on ffo_vp9_loop_filter_bd, in the same order.  lflvl_level = VP9Filter.level[64], lflvl_mask =
 * VP9Filter.mask[2][2][8][4]; lim_lut / mblim_lut = VP9Context.filter_lut (vp9.c:683-697); planes and strides in bytes;
 * (row, col) in 8-sample units as the reference passes them (superblock (r, c) -> row = 8 r, col = 8 c).
 * ---------------------------------------------------------------------------------------- */
static void lf_seg(int bd, int wd, int dir, uint8_t *p, ptrdiff_t ls, int L, const uint8_t *lim_lut, const uint8_t *mblim_lut)
{
    ffo_vp9_loop_filter_bd(bd, wd, dir, p, ls, mblim_lut[L], lim_lut[L], L >> 4);
}

static void lf_plane_cols(int bd, int col, int ss_h, int ss_v, const uint8_t *lvl, const uint8_t (*mask)[4], uint8_t *dst, ptrdiff_t ls,
                          const uint8_t *lim_lut, const uint8_t *mblim_lut)
{
    const int bpp = bd > 8 ? 2 : 1;
    for (int y = 0; y < 8; y += 2 << ss_v, dst += 16 * ls, lvl += 16 << ss_v) {
        uint8_t *ptr = dst;
        const uint8_t *l = lvl, *hmask1 = mask[y], *hmask2 = mask[y + 1 + ss_v];
        const unsigned hm1 = hmask1[0] | hmask1[1] | hmask1[2], hm13 = hmask1[3];
        const unsigned hm2 = hmask2[1] | hmask2[2], hm23 = hmask2[3];
        const unsigned hm = hm1 | hm2 | hm13 | hm23;
        for (unsigned x = 1; hm & ~(x - 1); x <<= 1, ptr += 8 * bpp >> ss_h) {
            if (col || x > 1) {
                if (hm1 & x) {
                    const int L = *l;
                    if (hmask1[0] & x) {
                        lf_seg(bd, 16, 0, ptr, ls, L, lim_lut, mblim_lut);
                        if (hmask2[0] & x) /* loop_filter_16: both halves with the first one's level */
                            lf_seg(bd, 16, 0, ptr + 8 * ls, ls, L, lim_lut, mblim_lut);
                    } else if (hm2 & x) { /* loop_filter_mix2 */
                        lf_seg(bd, (hmask1[1] & x) ? 8 : 4, 0, ptr, ls, L, lim_lut, mblim_lut);
                        lf_seg(bd, (hmask2[1] & x) ? 8 : 4, 0, ptr + 8 * ls, ls, l[8 << ss_v], lim_lut, mblim_lut);
                    } else {
                        lf_seg(bd, (hmask1[1] & x) ? 8 : 4, 0, ptr, ls, L, lim_lut, mblim_lut);
                    }
                } else if (hm2 & x) {
                    lf_seg(bd, (hmask2[1] & x) ? 8 : 4, 0, ptr + 8 * ls, ls, l[8 << ss_v], lim_lut, mblim_lut);
                }
            }
            if (ss_h) {
                if (x & 0xAA)
                    l += 2;
            } else {
                if (hm13 & x) {
                    lf_seg(bd, 4, 0, ptr + 4 * bpp, ls, *l, lim_lut, mblim_lut);
                    if (hm23 & x)
                        lf_seg(bd, 4, 0, ptr + 4 * bpp + 8 * ls, ls, l[8 << ss_v], lim_lut, mblim_lut);
                } else if (hm23 & x) {
                    lf_seg(bd, 4, 0, ptr + 8 * ls + 4 * bpp, ls, l[8 << ss_v], lim_lut, mblim_lut);
                }
                l++;
            }
        }
    }
}

static void lf_plane_rows(int bd, int row, int ss_h, int ss_v, const uint8_t *lvl, const uint8_t (*mask)[4], uint8_t *dst, ptrdiff_t ls,
                          const uint8_t *lim_lut, const uint8_t *mblim_lut)
{
    const int bpp = bd > 8 ? 2 : 1;
    for (int y = 0; y < 8; y++, dst += 8 * ls >> ss_v) {
        uint8_t *ptr = dst;
        const uint8_t *l = lvl, *vmask = mask[y];
        const unsigned vm = vmask[0] | vmask[1] | vmask[2], vm3 = vmask[3];
        for (unsigned x = 1; vm & ~(x - 1); x <<= (2 << ss_h), ptr += 16 * bpp, l += 2 << ss_h) {
            const unsigned x2 = x << (1 + ss_h);
            if (row || y) {
                if (vm & x) {
                    const int L = *l;
                    if (vmask[0] & x) {
                        lf_seg(bd, 16, 1, ptr, ls, L, lim_lut, mblim_lut);
                        if (vmask[0] & x2)
                            lf_seg(bd, 16, 1, ptr + 8 * bpp, ls, L, lim_lut, mblim_lut);
                    } else if (vm & x2) {
                        lf_seg(bd, (vmask[1] & x) ? 8 : 4, 1, ptr, ls, L, lim_lut, mblim_lut);
                        lf_seg(bd, (vmask[1] & x2) ? 8 : 4, 1, ptr + 8 * bpp, ls, l[1 + ss_h], lim_lut, mblim_lut);
                    } else {
                        lf_seg(bd, (vmask[1] & x) ? 8 : 4, 1, ptr, ls, L, lim_lut, mblim_lut);
                    }
                } else if (vm & x2) {
                    lf_seg(bd, (vmask[1] & x2) ? 8 : 4, 1, ptr + 8 * bpp, ls, l[1 + ss_h], lim_lut, mblim_lut);
                }
            }
            if (!ss_v) {
                if (vm3 & x) {
                    lf_seg(bd, 4, 1, ptr + ls * 4, ls, *l, lim_lut, mblim_lut);
                    if (vm3 & x2)
                        lf_seg(bd, 4, 1, ptr + ls * 4 + 8 * bpp, ls, l[1 + ss_h], lim_lut, mblim_lut);
                } else if (vm3 & x2) {
                    lf_seg(bd, 4, 1, ptr + ls * 4 + 8 * bpp, ls, l[1 + ss_h], lim_lut, mblim_lut);
                }
            }
        }
        if (ss_v) {
            if (y & 1)
                lvl += 16;
        } else {
            lvl += 8;
        }
    }
}

void ffo_vp9_loopfilter_sb(int bd, int ss_h, int ss_v, const uint8_t *lflvl_level, const uint8_t *lflvl_mask, int row, int col, uint8_t *y,
                           uint8_t *u, uint8_t *v, ptrdiff_t ls_y, ptrdiff_t ls_uv, const uint8_t *lim_lut, const uint8_t *mblim_lut)
{
    const uint8_t(*m)[2][8][4] = (const uint8_t(*)[2][8][4])lflvl_mask; /* mask[2][2][8][4] */
    const uint8_t(*uv)[8][4] = m[ss_h | ss_v];
    lf_plane_cols(bd, col, 0, 0, lflvl_level, m[0][0], y, ls_y, lim_lut, mblim_lut);
    lf_plane_rows(bd, row, 0, 0, lflvl_level, m[0][1], y, ls_y, lim_lut, mblim_lut);
    lf_plane_cols(bd, col, ss_h, ss_v, lflvl_level, uv[0], u, ls_uv, lim_lut, mblim_lut);
    lf_plane_rows(bd, row, ss_h, ss_v, lflvl_level, uv[1], u, ls_uv, lim_lut, mblim_lut);
    lf_plane_cols(bd, col, ss_h, ss_v, lflvl_level, uv[0], v, ls_uv, lim_lut, mblim_lut);
    lf_plane_rows(bd, row, ss_h, ss_v, lflvl_level, uv[1], v, ls_uv, lim_lut, mblim_lut);
}
