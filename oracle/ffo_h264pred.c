/*
 * ffo_h264pred.c — CPU restatement of H264PredContext for the H.264 codec, 8 bits, chroma_format_idc <= 1
 * (libavcodec/h264pred.h:92-116; the member table is filled at libavcodec/h264pred.c:448-538; the bodies are
 * libavcodec/h264pred_template.c).  TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * are the only callers; the product path never links or loads it.  Pinned bit-exact to the reference built in place
 * (tests/test_oracle_vs_ref.py::test_h264_pred_*).
 *
 * Restated, not transcribed: every mode is a per-sample rule over the block's "edge line"
 *     e[0..n-1] = the left column bottom-up (e[n-1-y] is the sample left of row y), e[n] = the corner,
 *     e[n+1+i]  = the row above, i = 0..2n-1 (running on into the top-right block),
 * which is also how the HIP kernel (ffmpeg_amd/csrc/kernels/h264_pred.hip) evaluates them.  pred8x8l runs the same rules over
 * the low-pass filtered line (h264pred_template.c:822-856).
 */
#include <string.h>
#include "ffo.h"

static int a2(int a, int b) { return (a + b + 1) >> 1; }
static int a3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }
static int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

/* which neighbours a directional mode reads: bit0 left, bit1 top, bit2 corner, bit3 top-right (indexed by the pred4x4 enum,
 * h264pred.h:35-48) */
static const unsigned char need4[12] = { 2, 1, 3, 2 | 8, 7, 7, 7, 2 | 8, 1, 1, 2, 0 };

/* the nine directional rules + the DC family over an edge line of n = 4 or 8 (h264pred_template.c:34-330, :858-1102) */
static int dir_sample(int mode, const int *e, int n, int x, int y, int dc)
{
    const int *T = e + n + 1;
    switch (mode) {
    case 0: return T[x];
    case 1: return e[n - 1 - y];
    case 3: {
        const int i = x + y;
        return i < 2 * n - 2 ? a3(T[i], T[i + 1], T[i + 2]) : (T[2 * n - 2] + 3 * T[2 * n - 1] + 2) >> 2;
    }
    case 4: {
        const int i = n - 1 - y + x;
        return a3(e[i], e[i + 1], e[i + 2]);
    }
    case 5: {
        const int d = 2 * x - y; /* >= 0: along the top from the corner; < 0: down the left column */
        if (d < 0)
            return a3(e[n + d], e[n + d + 1], e[n + d + 2]);
        return (d & 1) ? a3(e[n + (d >> 1)], e[n + (d >> 1) + 1], e[n + (d >> 1) + 2]) : a2(e[n + (d >> 1)], e[n + (d >> 1) + 1]);
    }
    case 6: {
        const int d = 2 * y - x; /* >= 0: down the left column from the corner; < 0: along the top */
        if (d < 0)
            return a3(e[n - d - 2], e[n - d - 1], e[n - d]);
        return (d & 1) ? a3(e[n - (d >> 1)], e[n - (d >> 1) - 1], e[n - (d >> 1) - 2]) : a2(e[n - (d >> 1)], e[n - (d >> 1) - 1]);
    }
    case 7: {
        const int i = (y >> 1) + x;
        return (y & 1) ? a3(T[i], T[i + 1], T[i + 2]) : a2(T[i], T[i + 1]);
    }
    case 8: {
        const int i = 2 * y + x, j = n - 1 - (i >> 1); /* e[j] = left[i >> 1] */
        if (i >= 2 * n - 2)
            return e[0];
        if (i == 2 * n - 3)
            return (e[1] + 3 * e[0] + 2) >> 2;
        return (i & 1) ? a3(e[j], e[j - 1], e[j - 2]) : a2(e[j], e[j - 1]);
    }
    default: return dc;
    }
}

static int dir_dc(int mode, const int *e, int n)
{
    int sl = 0, st = 0;
    for (int i = 0; i < n; i++) {
        sl += e[i];
        st += e[n + 1 + i];
    }
    if (mode == 2)
        return (sl + st + n) >> (n == 4 ? 3 : 4);
    if (mode == 9)
        return (sl + n / 2) >> (n == 4 ? 2 : 3);
    if (mode == 10)
        return (st + n / 2) >> (n == 4 ? 2 : 3);
    return 128;
}

/* H264PredContext.pred4x4[mode] (h264pred.h:93; h264pred_template.c:34-330) */
void ffo_h264_pred4x4(int mode, uint8_t *src, const uint8_t *topright, ptrdiff_t stride)
{
    int e[4 + 1 + 8] = { 0 };
    const unsigned need = need4[mode];
    if (need & 1)
        for (int y = 0; y < 4; y++)
            e[3 - y] = src[y * stride - 1];
    if (need & 2)
        for (int x = 0; x < 4; x++)
            e[5 + x] = src[x - stride];
    if (need & 4)
        e[4] = src[-1 - stride];
    if (need & 8)
        for (int x = 0; x < 4; x++)
            e[9 + x] = topright[x];
    const int dc = dir_dc(mode, e, 4);
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++)
            src[y * stride + x] = (uint8_t)dir_sample(mode, e, 4, x, y, dc);
}

/* the filtered edge line of an 8x8 luma block (PREDICT_8x8_LOAD_*, h264pred_template.c:822-856) */
static void edge8x8l(int *f, unsigned need, const uint8_t *src, int has_topleft, int has_topright, ptrdiff_t stride)
{
    memset(f, 0, 25 * sizeof(*f));
    if (need & 1) {
        int L[8];
        for (int y = 0; y < 8; y++)
            L[y] = src[y * stride - 1];
        f[7] = a3(has_topleft ? src[-1 - stride] : L[0], L[0], L[1]);
        for (int y = 1; y < 7; y++)
            f[7 - y] = a3(L[y - 1], L[y], L[y + 1]);
        f[0] = (L[6] + 3 * L[7] + 2) >> 2;
    }
    if (need & 2) {
        const uint8_t *T = src - stride;
        f[9] = a3(has_topleft ? T[-1] : T[0], T[0], T[1]);
        for (int x = 1; x < 7; x++)
            f[9 + x] = a3(T[x - 1], T[x], T[x + 1]);
        f[16] = a3(has_topright ? T[8] : T[7], T[7], T[6]);
        if (need & 8) {
            if (has_topright) {
                for (int x = 8; x < 15; x++)
                    f[9 + x] = a3(T[x - 1], T[x], T[x + 1]);
                f[24] = (T[14] + 3 * T[15] + 2) >> 2;
            } else {
                for (int x = 8; x < 16; x++)
                    f[9 + x] = T[7];
            }
        }
    }
    if (need & 4)
        f[8] = a3(src[-1], src[-1 - stride], src[-stride]);
}

/* H264PredContext.pred8x8l[mode] (h264pred.h:95; h264pred_template.c:858-1102) */
void ffo_h264_pred8x8l(int mode, uint8_t *src, int has_topleft, int has_topright, ptrdiff_t stride)
{
    int f[25];
    edge8x8l(f, need4[mode], src, has_topleft, has_topright, stride);
    const int dc = dir_dc(mode, f, 8);
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            src[y * stride + x] = (uint8_t)dir_sample(mode, f, 8, x, y, dc);
}

/* the plane predictor (h264pred_template.c:410-455 for 16x16, :746-780 for 8x8): a gradient fitted to the border */
static void plane(uint8_t *src, ptrdiff_t stride, int n)
{
    const int h = n / 2, mul = n == 16 ? 5 : 17, rnd = n == 16 ? 32 : 16, sh = n == 16 ? 6 : 5;
    int H = 0, V = 0;
    for (int k = 1; k <= h; k++) {
        H += k * (src[h - 1 + k - stride] - src[h - 1 - k - stride]);
        V += k * (src[(h - 1 + k) * stride - 1] - src[(h - 1 - k) * stride - 1]);
    }
    H = (mul * H + rnd) >> sh;
    V = (mul * V + rnd) >> sh;
    const int a = 16 * (src[(n - 1) * stride - 1] + src[n - 1 - stride] + 1) - (h - 1) * (V + H);
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            src[y * stride + x] = (uint8_t)clip8((a + y * V + x * H) >> 5);
}

/* H264PredContext.pred8x8[mode], chroma_format_idc <= 1 (h264pred.h:97; h264pred_template.c:463-800): DC per 4x4 quadrant */
void ffo_h264_pred8x8(int mode, uint8_t *src, ptrdiff_t stride)
{
    if (mode == 3) {
        plane(src, stride, 8);
        return;
    }
    int q[4] = { 128, 128, 128, 128 }; /* quadrant DCs: [0] top-left, [1] top-right, [2] bottom-left, [3] bottom-right */
    int t0 = 0, t1 = 0, l0 = 0, l1 = 0;
    const int use_t = mode == 0 || mode == 2 || mode == 5 || mode == 7 || mode == 8;
    const int use_l = mode == 0 || mode == 1 || mode == 4 || mode >= 7;
    if (use_t)
        for (int i = 0; i < 4; i++) {
            t0 += src[i - stride];
            t1 += src[4 + i - stride];
        }
    if (use_l)
        for (int i = 0; i < 4; i++) {
            l0 += src[i * stride - 1];
            if (mode != 7)
                l1 += src[(i + 4) * stride - 1];
        }
    switch (mode) {
    case 0: q[0] = (t0 + l0 + 4) >> 3; q[1] = (t1 + 2) >> 2; q[2] = (l1 + 2) >> 2; q[3] = (t1 + l1 + 4) >> 3; break;
    case 4: q[0] = q[1] = (l0 + 2) >> 2; q[2] = q[3] = (l1 + 2) >> 2; break;
    case 5: q[0] = q[2] = (t0 + 2) >> 2; q[1] = q[3] = (t1 + 2) >> 2; break;
    case 7: q[0] = (t0 + l0 + 4) >> 3; q[2] = (t0 + 2) >> 2; q[1] = q[3] = (t1 + 2) >> 2; break;   /* mad cow: left 0..3 + top */
    case 8: q[0] = (t0 + 2) >> 2; q[1] = (t1 + 2) >> 2; q[2] = (l1 + 2) >> 2; q[3] = (t1 + l1 + 4) >> 3; break;
    case 9: q[0] = q[1] = (l0 + 2) >> 2; break;
    case 10: q[2] = q[3] = (l1 + 2) >> 2; break;
    default: break;
    }
    uint8_t top[8], left[8];
    if (mode == 2)
        memcpy(top, src - stride, 8);
    if (mode == 1)
        for (int y = 0; y < 8; y++)
            left[y] = src[y * stride - 1];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            src[y * stride + x] = mode == 1 ? left[y] : mode == 2 ? top[x] : (uint8_t)q[2 * (y >> 2) + (x >> 2)];
}

/* H264PredContext.pred8x8[mode] at chroma_format_idc 2 (h264pred.c:478-512): the 8 wide x 16 tall forms of 4:2:2 chroma
 * (h264pred_template.c:477-817).  One DC per 4x4 cell, o[2 * cell_row + cell_col]; the sums are the same four-sample groups:
 * t0 / t1 over the top row's halves, l[r] over rows 4r..4r+3 of the left column. */
void ffo_h264_pred8x16(int mode, uint8_t *src, ptrdiff_t stride)
{
    if (mode == 3) {                                                 /* pred8x16_plane (:781-817) */
        int H = 0, V = 0;
        for (int k = 1; k <= 4; k++)
            H += k * (src[3 + k - stride] - src[3 - k - stride]);
        for (int k = 1; k <= 8; k++)
            V += k * (src[(7 + k) * stride - 1] - src[(7 - k) * stride - 1]);
        H = (17 * H + 16) >> 5;
        V = (5 * V + 32) >> 6;
        const int a = 16 * (src[15 * stride - 1] + src[7 - stride] + 1) - 7 * V - 3 * H;
        for (int y = 0; y < 16; y++)
            for (int x = 0; x < 8; x++)
                src[y * stride + x] = (uint8_t)clip8((a + y * V + x * H) >> 5);
        return;
    }
    int o[8], t0 = 0, t1 = 0, l[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 8; i++)
        o[i] = 128;
    const int use_t = mode == 0 || mode == 2 || mode == 5 || mode == 7 || mode == 8;
    const int use_l = mode == 0 || mode == 1 || mode == 4 || mode >= 7;
    if (use_t)
        for (int i = 0; i < 4; i++) {
            t0 += src[i - stride];
            t1 += src[4 + i - stride];
        }
    if (use_l)
        for (int r = 0; r < 4; r++)
            for (int i = 0; i < 4; i++)
                if (mode != 7 || r == 0)
                    l[r] += src[(4 * r + i) * stride - 1];
    switch (mode) {
    case 0: case 8:                                                  /* pred8x16_dc (:650-695); 8 = _0lt: cell 0 from the top alone */
        o[0] = mode == 0 ? (t0 + l[0] + 4) >> 3 : (t0 + 2) >> 2;
        o[1] = (t1 + 2) >> 2;
        for (int r = 1; r < 4; r++) {
            o[2 * r] = (l[r] + 2) >> 2;
            o[2 * r + 1] = (t1 + l[r] + 4) >> 3;
        }
        break;
    case 4: case 9: case 10:                                         /* left_dc (:567-571); 9 = _l00: cells of rows 4..7 are 128;
                                                                        10 = _0l0: cells of rows 0..3 are 128 (:725-749) */
        for (int r = 0; r < 4; r++)
            if (!(mode == 9 && r == 1) && !(mode == 10 && r == 0))
                o[2 * r] = o[2 * r + 1] = (l[r] + 2) >> 2;
        break;
    case 5: case 7:                                                  /* top_dc (:599-603); 7 = _l0t: cell 0 is pred4x4_dc */
        for (int r = 0; r < 4; r++) {
            o[2 * r] = (t0 + 2) >> 2;
            o[2 * r + 1] = (t1 + 2) >> 2;
        }
        if (mode == 7)
            o[0] = (t0 + l[0] + 4) >> 3;
        break;
    default: break;                                                  /* 6: 128 everywhere */
    }
    uint8_t top[8], left[16];
    if (mode == 2)
        memcpy(top, src - stride, 8);
    if (mode == 1)
        for (int y = 0; y < 16; y++)
            left[y] = src[y * stride - 1];
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 8; x++)
            src[y * stride + x] = mode == 1 ? left[y] : mode == 2 ? top[x] : (uint8_t)o[2 * (y >> 2) + (x >> 2)];
}

/* H264PredContext.pred16x16[mode] (h264pred.h:98; h264pred_template.c:332-461) */
void ffo_h264_pred16x16(int mode, uint8_t *src, ptrdiff_t stride)
{
    if (mode == 3) {
        plane(src, stride, 16);
        return;
    }
    int st = 0, sl = 0, dc = 128;
    uint8_t top[16], left[16];
    if (mode == 0 || mode == 2 || mode == 5)
        for (int i = 0; i < 16; i++)
            st += top[i] = src[i - stride];
    if (mode == 0 || mode == 1 || mode == 4)
        for (int i = 0; i < 16; i++)
            sl += left[i] = src[i * stride - 1];
    if (mode == 0)
        dc = (st + sl + 16) >> 5;
    else if (mode == 4)
        dc = (sl + 8) >> 4;
    else if (mode == 5)
        dc = (st + 8) >> 4;
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++)
            src[y * stride + x] = mode == 1 ? left[y] : mode == 2 ? top[x] : (uint8_t)dc;
}

/* the lossless (transform-bypass) predictors: the residual is integrated along the prediction direction, in 8-bit wrapping
 * arithmetic, and the coefficient block is cleared (h264pred_template.c:1104-1330).  pred[] holds the n border samples. */
static void integrate(int mode, uint8_t *pix, int16_t *block, ptrdiff_t stride, int n, const int *pred)
{
    for (int i = 0; i < n; i++) {
        unsigned v = (unsigned)pred[i];
        for (int k = 0; k < n; k++) {
            v = (v + (unsigned)block[mode == 0 ? k * n + i : i * n + k]) & 255;
            pix[mode == 0 ? k * stride + i : i * stride + k] = (uint8_t)v;
        }
    }
    memset(block, 0, sizeof(*block) * n * n);
}
static void plain_add(int mode, uint8_t *pix, int16_t *block, ptrdiff_t stride, int n)
{
    int pred[8];
    for (int i = 0; i < n; i++)
        pred[i] = mode == 0 ? pix[i - stride] : pix[i * stride - 1];
    integrate(mode, pix, block, stride, n, pred);
}
/* mode: 0 = VERT_PRED, 1 = HOR_PRED (the only members the reference fills) */
void ffo_h264_pred4x4_add(int mode, uint8_t *pix, int16_t *block, ptrdiff_t stride) { plain_add(mode, pix, block, stride, 4); }
void ffo_h264_pred8x8l_add(int mode, uint8_t *pix, int16_t *block, ptrdiff_t stride) { plain_add(mode, pix, block, stride, 8); }
void ffo_h264_pred8x8l_filter_add(int mode, uint8_t *pix, int16_t *block, int has_topleft, int has_topright, ptrdiff_t stride)
{
    int f[25], pred[8];
    edge8x8l(f, mode == 0 ? 2 : 1, pix, has_topleft, has_topright, stride);
    for (int i = 0; i < 8; i++)
        pred[i] = mode == 0 ? f[9 + i] : f[7 - i];
    integrate(mode, pix, block, stride, 8, pred);
}
/* mode: 2 = VERT_PRED8x8, 1 = HOR_PRED8x8 (h264pred.h:73-76) — sequences of 4x4 blocks at block_offset[] */
void ffo_h264_pred8x8_add(int mode, uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    for (int i = 0; i < 4; i++)
        plain_add(mode == 2 ? 0 : 1, pix + block_offset[i], block + i * 16, stride, 4);
}
/* pred8x8_add[] at chroma_format_idc 2 (h264pred_template.c:1302-1330): blocks 0..3 at block_offset[0..3], 4..7 at [8..11] */
void ffo_h264_pred8x16_add(int mode, uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    for (int i = 0; i < 8; i++)
        plain_add(mode == 2 ? 0 : 1, pix + block_offset[i < 4 ? i : i + 4], block + i * 16, stride, 4);
}
void ffo_h264_pred16x16_add(int mode, uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    for (int i = 0; i < 16; i++)
        plain_add(mode == 2 ? 0 : 1, pix + block_offset[i], block + i * 16, stride, 4);
}
