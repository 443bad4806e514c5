/*
 * ffo_h264_hbd.c — CPU restatement of the reference's h264dsp / h264qpel / h264chroma templates at ANY bit depth and for the
 * 4:2:2 / MBAFF members.  TEST INFRASTRUCTURE ONLY (see ffo.h).  Pinned against oracle/_ref (the reference's own instantiations at
 * 8 / 9 / 10 / 12 / 14 bits: libavcodec/h264dsp.c:81-153, h264qpel.c:87-103, h264chroma.c:38-52) by tests/test_oracle_vs_ref_h264_hbd.py.
 *
 * The reference instantiates its templates per BIT_DEPTH (libavcodec/bit_depth_template.c: pixel = uint8_t / uint16_t, dctcoef =
 * int16_t / int32_t, av_clip_pixel = clip to (1 << BIT_DEPTH) - 1, strides in BYTES).  Here the depth is an argument; above 8 bits
 * samples are uint16_t and coefficients int32_t.  Each function names the template lines it follows.
 */
#include <stdint.h>
#include <string.h>

#include "ffo.h"

static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int clip_px(int v, int bd) { const int m = (1 << bd) - 1; return v < 0 ? 0 : v > m ? m : v; }
/* sample i of a line, strides in samples */
static inline int rd(const uint8_t *p, int bd, ptrdiff_t i) { return bd > 8 ? ((const uint16_t *)p)[i] : p[i]; }
static inline void wr(uint8_t *p, int bd, ptrdiff_t i, int v) { if (bd > 8) ((uint16_t *)p)[i] = (uint16_t)v; else p[i] = (uint8_t)v; }
/* coefficient i of a block */
static inline int32_t cf(const int16_t *b, int bd, int i) { return bd > 8 ? ((const int32_t *)b)[i] : b[i]; }
static inline void cfw(int16_t *b, int bd, int i, uint32_t v) { if (bd > 8) ((int32_t *)b)[i] = (int32_t)v; else b[i] = (int16_t)v; }

/* scan8[]: libavcodec/h264_parse.h:40-57 */
static const uint8_t scan8[16 * 3] = {
    4 + 1 * 8, 5 + 1 * 8, 4 + 2 * 8, 5 + 2 * 8, 6 + 1 * 8, 7 + 1 * 8, 6 + 2 * 8, 7 + 2 * 8,
    4 + 3 * 8, 5 + 3 * 8, 4 + 4 * 8, 5 + 4 * 8, 6 + 3 * 8, 7 + 3 * 8, 6 + 4 * 8, 7 + 4 * 8,
    4 + 6 * 8, 5 + 6 * 8, 4 + 7 * 8, 5 + 7 * 8, 6 + 6 * 8, 7 + 6 * 8, 6 + 7 * 8, 7 + 7 * 8,
    4 + 8 * 8, 5 + 8 * 8, 4 + 9 * 8, 5 + 9 * 8, 6 + 8 * 8, 7 + 8 * 8, 6 + 9 * 8, 7 + 9 * 8,
    4 + 11 * 8, 5 + 11 * 8, 4 + 12 * 8, 5 + 12 * 8, 6 + 11 * 8, 7 + 11 * 8, 6 + 12 * 8, 7 + 12 * 8,
    4 + 13 * 8, 5 + 13 * 8, 4 + 14 * 8, 5 + 14 * 8, 6 + 13 * 8, 7 + 13 * 8, 6 + 14 * 8, 7 + 14 * 8,
};

/* ---- IDCT: h264idct_template.c:33-175, h264addpx_template.c:30-74.  kind = FFHIP_H264_IDCT4 .. ADD_PIXELS8_CLEAR ---------------- */
static void idct8_1d(const int32_t in[8], uint32_t out[8])
{
    uint32_t a0 = (uint32_t)in[0] + (uint32_t)in[4], a2 = (uint32_t)in[0] - (uint32_t)in[4];
    uint32_t a4 = (uint32_t)(in[2] >> 1) - (uint32_t)in[6], a6 = (uint32_t)(in[6] >> 1) + (uint32_t)in[2];
    uint32_t b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    int32_t a1 = (int32_t)(-(uint32_t)in[3] + (uint32_t)in[5] - (uint32_t)in[7] - (uint32_t)(in[7] >> 1));
    int32_t a3 = (int32_t)((uint32_t)in[1] + (uint32_t)in[7] - (uint32_t)in[3] - (uint32_t)(in[3] >> 1));
    int32_t a5 = (int32_t)(-(uint32_t)in[1] + (uint32_t)in[7] + (uint32_t)in[5] + (uint32_t)(in[5] >> 1));
    int32_t a7 = (int32_t)((uint32_t)in[3] + (uint32_t)in[5] + (uint32_t)in[1] + (uint32_t)(in[1] >> 1));
    uint32_t b1 = (uint32_t)(a7 >> 2) + (uint32_t)a1, b3 = (uint32_t)a3 + (uint32_t)(a5 >> 2);
    uint32_t b5 = (uint32_t)(a3 >> 2) - (uint32_t)a5, b7 = (uint32_t)a7 - (uint32_t)(a1 >> 2);
    out[0] = b0 + b7; out[7] = b0 - b7; out[1] = b2 + b5; out[6] = b2 - b5;
    out[2] = b4 + b3; out[5] = b4 - b3; out[3] = b6 + b1; out[4] = b6 - b1;
}

void ffo_h264_idct_bd(int bd, int kind, uint8_t *dst, int16_t *block, ptrdiff_t stride)
{
    const ptrdiff_t s = bd > 8 ? stride / 2 : stride;
    const int csz = bd > 8 ? 4 : 2;
    if (kind == 0) { /* ff_h264_idct_add */
        cfw(block, bd, 0, (uint32_t)cf(block, bd, 0) + 32);
        for (int i = 0; i < 4; i++) {
            const int32_t c0 = cf(block, bd, i), c1 = cf(block, bd, i + 4), c2 = cf(block, bd, i + 8), c3 = cf(block, bd, i + 12);
            uint32_t z0 = (uint32_t)c0 + (uint32_t)c2, z1 = (uint32_t)c0 - (uint32_t)c2;
            uint32_t z2 = (uint32_t)(c1 >> 1) - (uint32_t)c3, z3 = (uint32_t)c1 + (uint32_t)(c3 >> 1);
            cfw(block, bd, i, z0 + z3); cfw(block, bd, i + 4, z1 + z2); cfw(block, bd, i + 8, z1 - z2); cfw(block, bd, i + 12, z0 - z3);
        }
        for (int i = 0; i < 4; i++) {
            const int32_t c0 = cf(block, bd, 4 * i), c1 = cf(block, bd, 4 * i + 1), c2 = cf(block, bd, 4 * i + 2), c3 = cf(block, bd, 4 * i + 3);
            uint32_t z0 = (uint32_t)c0 + (uint32_t)c2, z1 = (uint32_t)c0 - (uint32_t)c2;
            uint32_t z2 = (uint32_t)(c1 >> 1) - (uint32_t)c3, z3 = (uint32_t)c1 + (uint32_t)(c3 >> 1);
            wr(dst, bd, i, clip_px(rd(dst, bd, i) + ((int32_t)(z0 + z3) >> 6), bd));
            wr(dst, bd, i + s, clip_px(rd(dst, bd, i + s) + ((int32_t)(z1 + z2) >> 6), bd));
            wr(dst, bd, i + 2 * s, clip_px(rd(dst, bd, i + 2 * s) + ((int32_t)(z1 - z2) >> 6), bd));
            wr(dst, bd, i + 3 * s, clip_px(rd(dst, bd, i + 3 * s) + ((int32_t)(z0 - z3) >> 6), bd));
        }
        memset(block, 0, 16 * csz);
    } else if (kind == 1) { /* ff_h264_idct8_add */
        int32_t in[8];
        uint32_t out[8];
        cfw(block, bd, 0, (uint32_t)cf(block, bd, 0) + 32);
        for (int i = 0; i < 8; i++) {
            for (int k = 0; k < 8; k++) in[k] = cf(block, bd, i + 8 * k);
            idct8_1d(in, out);
            for (int k = 0; k < 8; k++) cfw(block, bd, i + 8 * k, out[k]);
        }
        for (int i = 0; i < 8; i++) {
            for (int k = 0; k < 8; k++) in[k] = cf(block, bd, 8 * i + k);
            idct8_1d(in, out);
            for (int k = 0; k < 8; k++) wr(dst, bd, i + k * s, clip_px(rd(dst, bd, i + k * s) + ((int32_t)out[k] >> 6), bd));
        }
        memset(block, 0, 64 * csz);
    } else if (kind == 2 || kind == 3) { /* ff_h264_idct_dc_add / idct8_dc_add */
        const int n = kind == 2 ? 4 : 8, dc = (cf(block, bd, 0) + 32) >> 6;
        cfw(block, bd, 0, 0);
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++)
                wr(dst, bd, x + y * s, clip_px(rd(dst, bd, x + y * s) + dc, bd));
    } else { /* ff_h264_add_pixels4 / 8: dst += coefficient, no clipping, the sample type's wrap-around */
        const int n = kind == 4 ? 4 : 8;
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++)
                wr(dst, bd, x + y * s, (int)((unsigned)rd(dst, bd, x + y * s) + (unsigned)cf(block, bd, y * n + x)));
        memset(block, 0, n * n * csz);
    }
}

/* dispatchers: h264idct_template.c:177-262.  `block + i*16*sizeof(pixel)` in int16_t units = coefficient block i at either depth */
#define BLK(i) (block + (i) * 16 * (bd > 8 ? 2 : 1))
void ffo_h264_idct_mb_bd(int bd, int which, uint8_t *dst, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc)
{
    if (which == 1) { /* idct8_add4 */
        for (int i = 0; i < 16; i += 4) {
            const int nnz = nnzc[scan8[i]];
            if (nnz)
                ffo_h264_idct_bd(bd, (nnz == 1 && cf(BLK(i), bd, 0)) ? 3 : 1, dst + bo[i], BLK(i), stride);
        }
        return;
    }
    for (int i = 0; i < 16; i++) {
        const int nnz = nnzc[scan8[i]];
        if (which == 0) { /* idct_add16 */
            if (nnz)
                ffo_h264_idct_bd(bd, (nnz == 1 && cf(BLK(i), bd, 0)) ? 2 : 0, dst + bo[i], BLK(i), stride);
        } else { /* idct_add16intra */
            if (nnz) ffo_h264_idct_bd(bd, 0, dst + bo[i], BLK(i), stride);
            else if (cf(BLK(i), bd, 0)) ffo_h264_idct_bd(bd, 2, dst + bo[i], BLK(i), stride);
        }
    }
}

void ffo_h264_idct_add8_bd(int bd, int is422, uint8_t **dest, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc)
{
    for (int j = 1; j < 3; j++)
        for (int i = j * 16; i < j * 16 + 4; i++) {
            if (nnzc[scan8[i]]) ffo_h264_idct_bd(bd, 0, dest[j - 1] + bo[i], BLK(i), stride);
            else if (cf(BLK(i), bd, 0)) ffo_h264_idct_bd(bd, 2, dest[j - 1] + bo[i], BLK(i), stride);
        }
    if (!is422)
        return;
    for (int j = 1; j < 3; j++)
        for (int i = j * 16 + 4; i < j * 16 + 8; i++) { /* ff_h264_idct_add8_422: the lower four blocks sit at i + 4 in the offset / nnz tables */
            if (nnzc[scan8[i + 4]]) ffo_h264_idct_bd(bd, 0, dest[j - 1] + bo[i + 4], BLK(i), stride);
            else if (cf(BLK(i), bd, 0)) ffo_h264_idct_bd(bd, 2, dest[j - 1] + bo[i + 4], BLK(i), stride);
        }
}
#undef BLK

/* ff_h264_luma_dc_dequant_idct: h264idct_template.c:264-302 */
void ffo_h264_luma_dc_dequant_bd(int bd, int16_t *output, int16_t *input, int qmul)
{
    static const uint8_t x_offset[4] = { 0, 2 * 16, 8 * 16, 10 * 16 };
    int temp[16];
    for (int i = 0; i < 4; i++) {
        const int z0 = cf(input, bd, 4 * i) + cf(input, bd, 4 * i + 1), z1 = cf(input, bd, 4 * i) - cf(input, bd, 4 * i + 1);
        const int z2 = cf(input, bd, 4 * i + 2) - cf(input, bd, 4 * i + 3), z3 = cf(input, bd, 4 * i + 2) + cf(input, bd, 4 * i + 3);
        temp[4 * i] = z0 + z3; temp[4 * i + 1] = z0 - z3; temp[4 * i + 2] = z1 - z2; temp[4 * i + 3] = z1 + z2;
    }
    for (int i = 0; i < 4; i++) {
        const int o = x_offset[i];
        const uint32_t z0 = (uint32_t)temp[i] + (uint32_t)temp[8 + i], z1 = (uint32_t)temp[i] - (uint32_t)temp[8 + i];
        const uint32_t z2 = (uint32_t)temp[4 + i] - (uint32_t)temp[12 + i], z3 = (uint32_t)temp[4 + i] + (uint32_t)temp[12 + i];
        cfw(output, bd, 16 * 0 + o, (uint32_t)((int32_t)((z0 + z3) * (uint32_t)qmul + 128) >> 8));
        cfw(output, bd, 16 * 1 + o, (uint32_t)((int32_t)((z1 + z2) * (uint32_t)qmul + 128) >> 8));
        cfw(output, bd, 16 * 4 + o, (uint32_t)((int32_t)((z1 - z2) * (uint32_t)qmul + 128) >> 8));
        cfw(output, bd, 16 * 5 + o, (uint32_t)((int32_t)((z0 - z3) * (uint32_t)qmul + 128) >> 8));
    }
}

/* ff_h264_chroma_dc_dequant_idct / ff_h264_chroma422_dc_dequant_idct: h264idct_template.c:304-352 */
void ffo_h264_chroma_dc_dequant_bd(int bd, int is422, int16_t *block, int qmul)
{
    const int stride = 32, xs = 16;
    if (!is422) {
        uint32_t a = (uint32_t)cf(block, bd, 0), b = (uint32_t)cf(block, bd, xs), c = (uint32_t)cf(block, bd, stride), d = (uint32_t)cf(block, bd, stride + xs);
        uint32_t e = a - b;
        a = a + b;
        b = c - d;
        c = c + d;
        cfw(block, bd, 0, (uint32_t)((int32_t)((a + c) * (uint32_t)qmul) >> 7));
        cfw(block, bd, xs, (uint32_t)((int32_t)((e + b) * (uint32_t)qmul) >> 7));
        cfw(block, bd, stride, (uint32_t)((int32_t)((a - c) * (uint32_t)qmul) >> 7));
        cfw(block, bd, stride + xs, (uint32_t)((int32_t)((e - b) * (uint32_t)qmul) >> 7));
        return;
    }
    uint32_t temp[8];
    static const uint8_t x_offset[2] = { 0, 16 };
    for (int i = 0; i < 4; i++) {
        temp[2 * i] = (uint32_t)cf(block, bd, stride * i) + (uint32_t)cf(block, bd, stride * i + xs);
        temp[2 * i + 1] = (uint32_t)cf(block, bd, stride * i) - (uint32_t)cf(block, bd, stride * i + xs);
    }
    for (int i = 0; i < 2; i++) {
        const int o = x_offset[i];
        const uint32_t z0 = temp[i] + temp[4 + i], z1 = temp[i] - temp[4 + i], z2 = temp[2 + i] - temp[6 + i], z3 = temp[2 + i] + temp[6 + i];
        cfw(block, bd, stride * 0 + o, (uint32_t)((int32_t)((z0 + z3) * (uint32_t)qmul + 128) >> 8));
        cfw(block, bd, stride * 1 + o, (uint32_t)((int32_t)((z1 + z2) * (uint32_t)qmul + 128) >> 8));
        cfw(block, bd, stride * 2 + o, (uint32_t)((int32_t)((z1 - z2) * (uint32_t)qmul + 128) >> 8));
        cfw(block, bd, stride * 3 + o, (uint32_t)((int32_t)((z0 - z3) * (uint32_t)qmul + 128) >> 8));
    }
}

/* ---- loop filters: h264dsp_template.c:104-330.  kind: bit 0 = the edge is vertical (h_ filter: samples of a line 1 apart),
 *      bit 1 = chroma, bit 2 = intra (bS 4); inner = lines per tc0 entry (luma 4 / MBAFF 2; chroma 2 / MBAFF 1 / 4:2:2 4 / 4:2:2 MBAFF 2) */
void ffo_h264_loop_filter_bd(int bd, int kind, int inner, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, const int8_t *tc0)
{
    const ptrdiff_t s = bd > 8 ? stride / 2 : stride;
    const ptrdiff_t xs = (kind & 1) ? 1 : s, ys = (kind & 1) ? s : 1;
    const int chroma = kind & 2, intra = kind & 4;
    alpha <<= bd - 8;
    beta <<= bd - 8;
    for (int d = 0; d < 4 * inner; d++, pix += ys * (bd > 8 ? 2 : 1)) {
        const int p0 = rd(pix, bd, -xs), p1 = rd(pix, bd, -2 * xs), q0 = rd(pix, bd, 0), q1 = rd(pix, bd, xs);
        if (intra) {
            if (!(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta))
                continue;
            if (chroma) {
                wr(pix, bd, -xs, (2 * p1 + p0 + q1 + 2) >> 2);
                wr(pix, bd, 0, (2 * q1 + q0 + p1 + 2) >> 2);
                continue;
            }
            const int p2 = rd(pix, bd, -3 * xs), q2 = rd(pix, bd, 2 * xs);
            if (iabs(p0 - q0) < ((alpha >> 2) + 2)) {
                if (iabs(p2 - p0) < beta) {
                    const int p3 = rd(pix, bd, -4 * xs);
                    wr(pix, bd, -xs, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                    wr(pix, bd, -2 * xs, (p2 + p1 + p0 + q0 + 2) >> 2);
                    wr(pix, bd, -3 * xs, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
                } else
                    wr(pix, bd, -xs, (2 * p1 + p0 + q1 + 2) >> 2);
                if (iabs(q2 - q0) < beta) {
                    const int q3 = rd(pix, bd, 3 * xs);
                    wr(pix, bd, 0, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                    wr(pix, bd, xs, (p0 + q0 + q1 + q2 + 2) >> 2);
                    wr(pix, bd, 2 * xs, (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
                } else
                    wr(pix, bd, 0, (2 * q1 + q0 + p1 + 2) >> 2);
            } else {
                wr(pix, bd, -xs, (2 * p1 + p0 + q1 + 2) >> 2);
                wr(pix, bd, 0, (2 * q1 + q0 + p1 + 2) >> 2);
            }
            continue;
        }
        const int t0 = tc0[d / inner];
        if (chroma) {
            const int tc = (int)(((unsigned)t0 - 1U) << (bd - 8)) + 1;
            if (tc <= 0)
                continue;
            if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta) {
                const int delta = clip3(((q0 - p0) * 4 + (p1 - q1) + 4) >> 3, -tc, tc);
                wr(pix, bd, -xs, clip_px(p0 + delta, bd));
                wr(pix, bd, 0, clip_px(q0 - delta, bd));
            }
            continue;
        }
        const int tc_orig = t0 * (1 << (bd - 8));
        if (tc_orig < 0)
            continue;
        const int p2 = rd(pix, bd, -3 * xs), q2 = rd(pix, bd, 2 * xs);
        if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta) {
            int tc = tc_orig;
            if (iabs(p2 - p0) < beta) {
                if (tc_orig)
                    wr(pix, bd, -2 * xs, p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc_orig, tc_orig));
                tc++;
            }
            if (iabs(q2 - q0) < beta) {
                if (tc_orig)
                    wr(pix, bd, xs, q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc_orig, tc_orig));
                tc++;
            }
            const int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
            wr(pix, bd, -xs, clip_px(p0 + delta, bd));
            wr(pix, bd, 0, clip_px(q0 - delta, bd));
        }
    }
}

/* ---- luma qpel: h264qpel_template.c:77-465 (lowpass filters), :375-465 (the 16 positions).  Exact integer arithmetic at every depth
 *      (the template's `pad` only re-centres the int16 temporaries of the 10-bit SIMD versions: it cancels) ------------------------ */
static int tap6(int a, int b, int c, int d, int e, int f) { return (c + d) * 20 - (b + e) * 5 + (a + f); }
static int hfilt(const uint8_t *p, int bd, ptrdiff_t i) { return tap6(rd(p, bd, i - 2), rd(p, bd, i - 1), rd(p, bd, i), rd(p, bd, i + 1), rd(p, bd, i + 2), rd(p, bd, i + 3)); }
static int vfilt(const uint8_t *p, int bd, ptrdiff_t i, ptrdiff_t s) { return tap6(rd(p, bd, i - 2 * s), rd(p, bd, i - s), rd(p, bd, i), rd(p, bd, i + s), rd(p, bd, i + 2 * s), rd(p, bd, i + 3 * s)); }

void ffo_h264_qpel_bd(int bd, int avg, int size_idx, int mcxy, uint8_t *dst, const uint8_t *src, ptrdiff_t stride)
{
    const ptrdiff_t s = bd > 8 ? stride / 2 : stride;
    const int n = 16 >> size_idx, mx = mcxy & 3, my = mcxy >> 2;
    int out[16][16];
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            const ptrdiff_t o = y * s + x;
#define H(at) clip_px((hfilt(src, bd, (at)) + 16) >> 5, bd)
#define V(at) clip_px((vfilt(src, bd, (at), s) + 16) >> 5, bd)
#define AVG2(a, b) (((a) + (b) + 1) >> 1)
            int hv = 0;
            if (mx == 2 || my == 2) { /* hv_lowpass: vertical 6-tap over the unrounded horizontal sums */
                int t[6];
                for (int k = 0; k < 6; k++)
                    t[k] = hfilt(src, bd, o + (k - 2) * s);
                hv = clip_px((tap6(t[0], t[1], t[2], t[3], t[4], t[5]) + 512) >> 10, bd);
            }
            int v;
            switch (mcxy) {
            case 0:  v = rd(src, bd, o); break;
            case 1:  v = AVG2(rd(src, bd, o), H(o)); break;
            case 2:  v = H(o); break;
            case 3:  v = AVG2(rd(src, bd, o + 1), H(o)); break;
            case 4:  v = AVG2(rd(src, bd, o), V(o)); break;
            case 8:  v = V(o); break;
            case 12: v = AVG2(rd(src, bd, o + s), V(o)); break;
            case 5:  v = AVG2(H(o), V(o)); break;
            case 7:  v = AVG2(H(o), V(o + 1)); break;
            case 13: v = AVG2(H(o + s), V(o)); break;
            case 15: v = AVG2(H(o + s), V(o + 1)); break;
            case 10: v = hv; break;
            case 6:  v = AVG2(H(o), hv); break;
            case 14: v = AVG2(H(o + s), hv); break;
            case 9:  v = AVG2(V(o), hv); break;
            default: v = AVG2(V(o + 1), hv); break; /* 11 */
            }
            out[y][x] = v;
            (void)mx; (void)my;
        }
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            wr(dst, bd, y * s + x, avg ? AVG2(rd(dst, bd, y * s + x), out[y][x]) : out[y][x]);
#undef H
#undef V
}

/* ---- chroma MC: h264chroma_template.c:28-190 (put/avg_h264_chroma_mc{8,4,2,1}): bilinear, no clipping needed at any depth ---------- */
void ffo_h264_chroma_mc_bd(int bd, int avg, int w, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h, int x, int y)
{
    const ptrdiff_t s = bd > 8 ? stride / 2 : stride;
    const int A = (8 - x) * (8 - y), B = x * (8 - y), Cc = (8 - x) * y, D = x * y;
    for (int r = 0; r < h; r++)
        for (int i = 0; i < w; i++) {
            const ptrdiff_t o = r * s + i;
            /* the template never reads a neighbour whose weight is zero (its D == 0 / E == 0 branches): neither does this */
            int v = A * rd(src, bd, o);
            if (B) v += B * rd(src, bd, o + 1);
            if (Cc) v += Cc * rd(src, bd, o + s);
            if (D) v += D * rd(src, bd, o + s + 1);
            v = (v + 32) >> 6;
            wr(dst, bd, o, avg ? (rd(dst, bd, o) + v + 1) >> 1 : v);
        }
}

/* ---- explicit weighted prediction: h264dsp_template.c:30-98 --------------------------------------------------------------------- */
void ffo_h264_weight_bd(int bd, int w, uint8_t *block, ptrdiff_t stride, int height, int log2_denom, int weight, int offset)
{
    const ptrdiff_t s = bd > 8 ? stride / 2 : stride;
    offset = (int)((unsigned)offset << (log2_denom + (bd - 8)));
    if (log2_denom)
        offset += 1 << (log2_denom - 1);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < w; x++)
            wr(block, bd, y * s + x, clip_px((rd(block, bd, y * s + x) * weight + offset) >> log2_denom, bd));
}

void ffo_h264_biweight_bd(int bd, int w, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd,
                          int weights, int offset)
{
    const ptrdiff_t s = bd > 8 ? stride / 2 : stride;
    offset = (int)((unsigned)offset << (bd - 8));
    offset = (int)((unsigned)((offset + 1) | 1) << log2_denom);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < w; x++)
            wr(dst, bd, y * s + x, clip_px((rd(src, bd, y * s + x) * weights + rd(dst, bd, y * s + x) * weightd + offset) >> (log2_denom + 1), bd));
}

/* Frame order at depth bd: ffo_h264_deblock_frame / _chroma (ffo_h264.c) with the depth's filters — the macroblocks in raster order,
 * each its vertical edges left to right, then its horizontal ones top to bottom (h264_loopfilter.c:716 ff_h264_filter_mb, the raster
 * walk of h264_slice.c loop_filter()).  Samples uint16_t above 8 bits, stride in bytes; the records as the 8-bit functions take them. */
void ffo_h264_deblock_frame_bd(int bd, int chroma, uint8_t *plane, ptrdiff_t stride, int mb_w, int mb_h, const FfoH264Edge *edges)
{
    const int n = chroma ? 8 : 16, ne = chroma ? 2 : 4, ps = bd > 8 ? 2 : 1;
    for (int my = 0; my < mb_h; my++)
        for (int mx = 0; mx < mb_w; mx++) {
            const FfoH264Edge *e = edges + (size_t)(my * mb_w + mx) * 2 * ne;
            uint8_t *mb = plane + (ptrdiff_t)my * n * stride + (ptrdiff_t)mx * n * ps;
            for (int dir = 0; dir < 2; dir++)
                for (int k = 0; k < ne; k++) {
                    const FfoH264Edge *ed = e + dir * ne + k;
                    uint8_t *pix = dir ? mb + (ptrdiff_t)4 * k * stride : mb + 4 * k * ps;
                    if (!ed->alpha || !ed->beta)
                        continue;
                    if (k == 0 && (dir ? my == 0 : mx == 0))
                        continue;
                    const int intra = ed->kind >= 4;
                    /* dir 0: a vertical edge -> the h_ filter (bit 0); chroma: bit 1; bS 4: bit 2; lines per tc0 entry: luma 4, chroma 2 */
                    ffo_h264_loop_filter_bd(bd, (dir ? 0 : 1) + (chroma ? 2 : 0) + (intra ? 4 : 0), chroma ? 2 : 4, pix, stride, ed->alpha, ed->beta, ed->tc0);
                }
        }
}

/* One 4:2:2 chroma plane in frame order (ff_h264_filter_mb at chroma_format_idc 2, h264_loopfilter.c:601-703): 8 x 16 macroblocks, six
 * edge records each — the vertical edges at x = 0, 4 (h_loop_filter_chroma422: 16 lines, tc0 per 4 lines), then the horizontal ones at
 * y = 0, 4, 8, 12 (v_loop_filter_chroma: 8 columns, tc0 per 2) — macroblocks in raster order, vertical before horizontal. */
void ffo_h264_deblock_frame_c422_bd(int bd, uint8_t *plane, ptrdiff_t stride, int mb_w, int mb_h, const FfoH264Edge *edges)
{
    const int ps = bd > 8 ? 2 : 1;
    for (int my = 0; my < mb_h; my++)
        for (int mx = 0; mx < mb_w; mx++) {
            const FfoH264Edge *e = edges + (size_t)(my * mb_w + mx) * 6;
            uint8_t *mb = plane + (ptrdiff_t)my * 16 * stride + (ptrdiff_t)mx * 8 * ps;
            for (int k = 0; k < 6; k++) {
                const FfoH264Edge *ed = e + k;
                const int dir = k >= 2, pos = dir ? k - 2 : k;
                uint8_t *pix = dir ? mb + (ptrdiff_t)4 * pos * stride : mb + 4 * pos * ps;
                if (!ed->alpha || !ed->beta)
                    continue;
                if (pos == 0 && (dir ? my == 0 : mx == 0))
                    continue;
                ffo_h264_loop_filter_bd(bd, (dir ? 0 : 1) + 2 + (ed->kind >= 4 ? 4 : 0), dir ? 2 : 4, pix, stride, ed->alpha, ed->beta, ed->tc0);
            }
        }
}
