/*
 * ffo_h264.c — CPU restatement of the reference's 8-bit h264dsp / h264qpel functions.
 * TEST INFRASTRUCTURE ONLY (see ffo.h).  Pinned against oracle/_ref and tests/golden.
 */
#include <stdint.h>
#include <string.h>

#include "ffo.h"

static inline int clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int iabs(int v) { return v < 0 ? -v : v; }

/* scan8[]: libavcodec/h264_parse.h:40-57 (luma part): position of 4x4 block i in the 8-wide nnz cache */
static const uint8_t scan8_luma[16] = {
    4 + 1 * 8, 5 + 1 * 8, 4 + 2 * 8, 5 + 2 * 8, 6 + 1 * 8, 7 + 1 * 8, 6 + 2 * 8, 7 + 2 * 8,
    4 + 3 * 8, 5 + 3 * 8, 4 + 4 * 8, 5 + 4 * 8, 6 + 3 * 8, 7 + 3 * 8, 6 + 4 * 8, 7 + 4 * 8,
};

/* ------------------------------------------------------------------------------------------
 * IDCT.  ff_h264_idct_add_8_c libavcodec/h264idct_template.c:33-67.
 * Coefficients are int16 and the first pass stores back into them (16-bit wrap); arithmetic is
 * modulo 2^32 with arithmetic right shifts.
 * ---------------------------------------------------------------------------------------- */
void ffo_h264_idct_add(uint8_t *dst, int16_t *block, ptrdiff_t stride)
{
    block[0] += 32;
    for (int i = 0; i < 4; i++) {
        uint32_t e0 = (uint32_t)block[i] + (uint32_t)block[i + 8];
        uint32_t e1 = (uint32_t)block[i] - (uint32_t)block[i + 8];
        uint32_t o0 = (uint32_t)(block[i + 4] >> 1) - (uint32_t)block[i + 12];
        uint32_t o1 = (uint32_t)block[i + 4] + (uint32_t)(block[i + 12] >> 1);
        block[i]      = (int16_t)(e0 + o1);
        block[i + 4]  = (int16_t)(e1 + o0);
        block[i + 8]  = (int16_t)(e1 - o0);
        block[i + 12] = (int16_t)(e0 - o1);
    }
    for (int i = 0; i < 4; i++) {
        const int16_t *r = block + 4 * i;
        uint32_t e0 = (uint32_t)r[0] + (uint32_t)r[2];
        uint32_t e1 = (uint32_t)r[0] - (uint32_t)r[2];
        uint32_t o0 = (uint32_t)(r[1] >> 1) - (uint32_t)r[3];
        uint32_t o1 = (uint32_t)r[1] + (uint32_t)(r[3] >> 1);
        dst[i]              = clip_u8(dst[i]              + ((int32_t)(e0 + o1) >> 6));
        dst[i + stride]     = clip_u8(dst[i + stride]     + ((int32_t)(e1 + o0) >> 6));
        dst[i + 2 * stride] = clip_u8(dst[i + 2 * stride] + ((int32_t)(e1 - o0) >> 6));
        dst[i + 3 * stride] = clip_u8(dst[i + 3 * stride] + ((int32_t)(e0 - o1) >> 6));
    }
    memset(block, 0, 16 * sizeof(*block));
}

/* one 8-point butterfly of ff_h264_idct8_add_8_c (h264idct_template.c:69-143) on in[0..7] */
static inline void idct8_1d(const int in[8], uint32_t out[8])
{
    uint32_t a0 = (uint32_t)in[0] + (uint32_t)in[4];
    uint32_t a2 = (uint32_t)in[0] - (uint32_t)in[4];
    uint32_t a4 = (uint32_t)(in[2] >> 1) - (uint32_t)in[6];
    uint32_t a6 = (uint32_t)(in[6] >> 1) + (uint32_t)in[2];
    uint32_t b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    int32_t a1 = (int32_t)(-(uint32_t)in[3] + (uint32_t)in[5] - (uint32_t)in[7] - (uint32_t)(in[7] >> 1));
    int32_t a3 = (int32_t)((uint32_t)in[1] + (uint32_t)in[7] - (uint32_t)in[3] - (uint32_t)(in[3] >> 1));
    int32_t a5 = (int32_t)(-(uint32_t)in[1] + (uint32_t)in[7] + (uint32_t)in[5] + (uint32_t)(in[5] >> 1));
    int32_t a7 = (int32_t)((uint32_t)in[3] + (uint32_t)in[5] + (uint32_t)in[1] + (uint32_t)(in[1] >> 1));
    uint32_t b1 = (uint32_t)(a7 >> 2) + (uint32_t)a1;
    uint32_t b3 = (uint32_t)a3 + (uint32_t)(a5 >> 2);
    uint32_t b5 = (uint32_t)(a3 >> 2) - (uint32_t)a5;
    uint32_t b7 = (uint32_t)a7 - (uint32_t)(a1 >> 2);
    out[0] = b0 + b7; out[7] = b0 - b7;
    out[1] = b2 + b5; out[6] = b2 - b5;
    out[2] = b4 + b3; out[5] = b4 - b3;
    out[3] = b6 + b1; out[4] = b6 - b1;
}

void ffo_h264_idct8_add(uint8_t *dst, int16_t *block, ptrdiff_t stride)
{
    int in[8];
    uint32_t out[8];
    block[0] += 32;
    for (int i = 0; i < 8; i++) {          /* pass 1: stride-8 samples, results wrap to int16 */
        for (int k = 0; k < 8; k++)
            in[k] = block[i + 8 * k];
        idct8_1d(in, out);
        for (int k = 0; k < 8; k++)
            block[i + 8 * k] = (int16_t)out[k];
    }
    for (int i = 0; i < 8; i++) {          /* pass 2: contiguous samples -> column i of dst */
        for (int k = 0; k < 8; k++)
            in[k] = block[8 * i + k];
        idct8_1d(in, out);
        for (int k = 0; k < 8; k++)
            dst[i + k * stride] = clip_u8(dst[i + k * stride] + ((int32_t)out[k] >> 6));
    }
    memset(block, 0, 64 * sizeof(*block));
}

/* ff_h264_idct_dc_add_8_c / ff_h264_idct8_dc_add_8_c: h264idct_template.c:145-175 */
static void dc_add(uint8_t *dst, int16_t *block, ptrdiff_t stride, int n)
{
    int dc = (block[0] + 32) >> 6;
    block[0] = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[x + y * stride] = clip_u8(dst[x + y * stride] + dc);
}
void ffo_h264_idct_dc_add(uint8_t *dst, int16_t *block, ptrdiff_t stride) { dc_add(dst, block, stride, 4); }
void ffo_h264_idct8_dc_add(uint8_t *dst, int16_t *block, ptrdiff_t stride) { dc_add(dst, block, stride, 8); }

/* dispatchers: h264idct_template.c:177-214 */
void ffo_h264_idct_add16(uint8_t *dst, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc)
{
    for (int i = 0; i < 16; i++) {
        int nnz = nnzc[scan8_luma[i]];
        if (!nnz)
            continue;
        if (nnz == 1 && block[i * 16])
            ffo_h264_idct_dc_add(dst + bo[i], block + i * 16, stride);
        else
            ffo_h264_idct_add(dst + bo[i], block + i * 16, stride);
    }
}
void ffo_h264_idct_add16intra(uint8_t *dst, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc)
{
    for (int i = 0; i < 16; i++) {
        if (nnzc[scan8_luma[i]])
            ffo_h264_idct_add(dst + bo[i], block + i * 16, stride);
        else if (block[i * 16])
            ffo_h264_idct_dc_add(dst + bo[i], block + i * 16, stride);
    }
}
/* position of chroma 4x4 block i (16..19, 32..35) in the 15-row nnz cache: the chroma rows of scan8[], h264_parse.h:45-52 */
static int scan8_chroma(int i)
{
    const int k = i & 15, row = (i >> 4) * 5 + 1; /* plane 1: cache rows 6,7; plane 2: rows 11,12 */
    return 4 + (k & 1) + ((k >> 2) & 1) * 2 + (row + ((k >> 1) & 1)) * 8;
}
/* ff_h264_idct_add8_8_c (4:2:0), h264idct_template.c:216-228: the four 4x4 blocks of each chroma plane; dest[0] Cb, dest[1] Cr */
void ffo_h264_idct_add8(uint8_t **dest, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc)
{
    for (int j = 1; j < 3; j++)
        for (int i = j * 16; i < j * 16 + 4; i++) {
            if (nnzc[scan8_chroma(i)])
                ffo_h264_idct_add(dest[j - 1] + bo[i], block + i * 16, stride);
            else if (block[i * 16])
                ffo_h264_idct_dc_add(dest[j - 1] + bo[i], block + i * 16, stride);
        }
}
/* ff_h264_luma_dc_dequant_idct_8_c, h264idct_template.c:259-293: 4x4 Hadamard of the 16 luma DC values, dequantised, scattered to
 * the DC positions of the macroblock's 16 blocks (unsigned wrap-around products, arithmetic >> 8) */
void ffo_h264_luma_dc_dequant_idct(int16_t *output, int16_t *input, int qmul)
{
    static const uint8_t x_offset[4] = { 0, 2 * 16, 8 * 16, 10 * 16 };
    int temp[16];
    for (int i = 0; i < 4; i++) {
        const int z0 = input[4 * i + 0] + input[4 * i + 1], z1 = input[4 * i + 0] - input[4 * i + 1];
        const int z2 = input[4 * i + 2] - input[4 * i + 3], z3 = input[4 * i + 2] + input[4 * i + 3];
        temp[4 * i + 0] = z0 + z3; temp[4 * i + 1] = z0 - z3; temp[4 * i + 2] = z1 - z2; temp[4 * i + 3] = z1 + z2;
    }
    for (int i = 0; i < 4; i++) {
        const int o = x_offset[i];
        const unsigned z0 = (unsigned)temp[i] + temp[8 + i], z1 = (unsigned)temp[i] - temp[8 + i];
        const unsigned z2 = (unsigned)temp[4 + i] - temp[12 + i], z3 = (unsigned)temp[4 + i] + temp[12 + i];
        output[16 * 0 + o] = (int16_t)((int)((z0 + z3) * (unsigned)qmul + 128) >> 8);
        output[16 * 1 + o] = (int16_t)((int)((z1 + z2) * (unsigned)qmul + 128) >> 8);
        output[16 * 4 + o] = (int16_t)((int)((z1 - z2) * (unsigned)qmul + 128) >> 8);
        output[16 * 5 + o] = (int16_t)((int)((z0 - z3) * (unsigned)qmul + 128) >> 8);
    }
}
/* ff_h264_chroma_dc_dequant_idct_8_c (4:2:0), h264idct_template.c:323-345: 2x2 Hadamard of the DCs at block[0,16,32,48] */
void ffo_h264_chroma_dc_dequant_idct(int16_t *block, int qmul)
{
    unsigned a = block[0], b = block[16], c = block[32], d = block[48], e;
    e = a - b; a = a + b; b = c - d; c = c + d;
    block[0]  = (int16_t)((int)((a + c) * (unsigned)qmul) >> 7);
    block[16] = (int16_t)((int)((e + b) * (unsigned)qmul) >> 7);
    block[32] = (int16_t)((int)((a - c) * (unsigned)qmul) >> 7);
    block[48] = (int16_t)((int)((e - b) * (unsigned)qmul) >> 7);
}
/* ff_h264_add_pixels4_8_c / ff_h264_add_pixels8_8_c (the lossless bypass), h264addpx_template.c:28-74: wrap-around add, then clear */
void ffo_h264_add_pixels_clear(int n, uint8_t *dst, int16_t *block, ptrdiff_t stride)
{
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[y * stride + x] = (uint8_t)(dst[y * stride + x] + (unsigned)block[y * n + x]);
    memset(block, 0, sizeof(int16_t) * n * n);
}
void ffo_h264_idct8_add4(uint8_t *dst, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc)
{
    for (int i = 0; i < 16; i += 4) {
        int nnz = nnzc[scan8_luma[i]];
        if (!nnz)
            continue;
        if (nnz == 1 && block[i * 16])
            ffo_h264_idct8_dc_add(dst + bo[i], block + i * 16, stride);
        else
            ffo_h264_idct8_add(dst + bo[i], block + i * 16, stride);
    }
}

/* ------------------------------------------------------------------------------------------
 * Deblocking: libavcodec/h264dsp_template.c:104-330.  xs = step across the edge, ys = step along it.
 * ---------------------------------------------------------------------------------------- */
static void lf_luma(uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, int alpha, int beta, const int8_t *tc0)
{
    for (int g = 0; g < 4; g++, pix += 4 * ys) {
        int t0 = tc0[g];
        if (t0 < 0)
            continue;
        for (int d = 0; d < 4; d++) {
            uint8_t *p = pix + d * ys;
            int p0 = p[-xs], p1 = p[-2 * xs], p2 = p[-3 * xs];
            int q0 = p[0], q1 = p[xs], q2 = p[2 * xs];
            if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta)
                continue;
            int tc = t0;
            if (iabs(p2 - p0) < beta) {
                if (t0)
                    p[-2 * xs] = p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -t0, t0);
                tc++;
            }
            if (iabs(q2 - q0) < beta) {
                if (t0)
                    p[xs] = q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -t0, t0);
                tc++;
            }
            int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
            p[-xs] = clip_u8(p0 + delta);
            p[0]   = clip_u8(q0 - delta);
        }
    }
}

static void lf_luma_intra(uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, int alpha, int beta)
{
    for (int d = 0; d < 16; d++, pix += ys) {
        int p2 = pix[-3 * xs], p1 = pix[-2 * xs], p0 = pix[-xs];
        int q0 = pix[0], q1 = pix[xs], q2 = pix[2 * xs];
        if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta)
            continue;
        if (iabs(p0 - q0) < ((alpha >> 2) + 2)) {
            if (iabs(p2 - p0) < beta) {
                int p3 = pix[-4 * xs];
                pix[-xs]     = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
                pix[-2 * xs] = (p2 + p1 + p0 + q0 + 2) >> 2;
                pix[-3 * xs] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
            } else {
                pix[-xs] = (2 * p1 + p0 + q1 + 2) >> 2;
            }
            if (iabs(q2 - q0) < beta) {
                int q3 = pix[3 * xs];
                pix[0]      = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
                pix[xs]     = (p0 + q0 + q1 + q2 + 2) >> 2;
                pix[2 * xs] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
            } else {
                pix[0] = (2 * q1 + q0 + p1 + 2) >> 2;
            }
        } else {
            pix[-xs] = (2 * p1 + p0 + q1 + 2) >> 2;
            pix[0]   = (2 * q1 + q0 + p1 + 2) >> 2;
        }
    }
}

static void lf_chroma(uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, int alpha, int beta, const int8_t *tc0)
{
    for (int g = 0; g < 4; g++, pix += 2 * ys) {
        int tc = tc0[g]; /* ((tc0 - 1U) << 0) + 1 at 8 bits */
        if (tc <= 0)
            continue;
        for (int d = 0; d < 2; d++) {
            uint8_t *p = pix + d * ys;
            int p0 = p[-xs], p1 = p[-2 * xs], q0 = p[0], q1 = p[xs];
            if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta) {
                int delta = clip3(((q0 - p0) * 4 + (p1 - q1) + 4) >> 3, -tc, tc);
                p[-xs] = clip_u8(p0 + delta);
                p[0]   = clip_u8(q0 - delta);
            }
        }
    }
}

static void lf_chroma_intra(uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, int alpha, int beta)
{
    for (int d = 0; d < 8; d++, pix += ys) {
        int p0 = pix[-xs], p1 = pix[-2 * xs], q0 = pix[0], q1 = pix[xs];
        if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta) {
            pix[-xs] = (2 * p1 + p0 + q1 + 2) >> 2;
            pix[0]   = (2 * q1 + q0 + p1 + 2) >> 2;
        }
    }
}

void ffo_h264_loop_filter(int which, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, const int8_t *tc0)
{
    /* v_ filters a horizontal edge: across = stride, along = 1; h_ the other way round */
    switch (which) {
    case 0: lf_luma(pix, stride, 1, alpha, beta, tc0); break;
    case 1: lf_luma(pix, 1, stride, alpha, beta, tc0); break;
    case 2: lf_chroma(pix, stride, 1, alpha, beta, tc0); break;
    case 3: lf_chroma(pix, 1, stride, alpha, beta, tc0); break;
    case 4: lf_luma_intra(pix, stride, 1, alpha, beta); break;
    case 5: lf_luma_intra(pix, 1, stride, alpha, beta); break;
    case 6: lf_chroma_intra(pix, stride, 1, alpha, beta); break;
    case 7: lf_chroma_intra(pix, 1, stride, alpha, beta); break;
    }
}

/*
 * Frame-order luma deblocking: the order ff_h264_filter_mb() issues h264dsp calls in
 * (libavcodec/h264_loopfilter.c:716-): MBs in raster order; per MB the 4 vertical edges left to
 * right, then the 4 horizontal edges top to bottom.  An edge with alpha == 0 or beta == 0 is
 * skipped (filter_mb_edgev/edgeh early return, h264_loopfilter.c:104,200); the picture's outer
 * left/top edges are never filtered.
 */
void ffo_h264_deblock_frame(uint8_t *luma, ptrdiff_t stride, int mb_w, int mb_h, const FfoH264Edge *edges)
{
    for (int my = 0; my < mb_h; my++)
        for (int mx = 0; mx < mb_w; mx++) {
            const FfoH264Edge *e = edges + (size_t)(my * mb_w + mx) * 8;
            uint8_t *mb = luma + (ptrdiff_t)my * 16 * stride + mx * 16;
            for (int dir = 0; dir < 2; dir++)
                for (int k = 0; k < 4; k++) {
                    const FfoH264Edge *ed = e + dir * 4 + k;
                    uint8_t *pix = dir ? mb + (ptrdiff_t)4 * k * stride : mb + 4 * k;
                    if (!ed->alpha || !ed->beta)
                        continue;
                    if (k == 0 && (dir ? my == 0 : mx == 0))
                        continue;
                    int intra = ed->kind >= 4;
                    /* dir 0: vertical edge -> h_loop_filter (1/5); dir 1: horizontal -> v_ (0/4) */
                    ffo_h264_loop_filter((dir ? 0 : 1) + (intra ? 4 : 0), pix, stride, ed->alpha, ed->beta, ed->tc0);
                }
        }
}

/*
 * The same for one 4:2:0 chroma plane: per macroblock (8x8 chroma samples) the vertical edges at x = 0 and 4, then the
 * horizontal ones at y = 0 and 4 — filter_mb_dir() filters chroma on the even luma edges (h264_loopfilter.c:644-700,
 * (edge & 1) == 0, dest_cb + 2 * edge).  edges[(mb * 2 + dir) * 2 + e]; kinds: the chroma ones of FFHIP_H264_LF_*.
 */
void ffo_h264_deblock_frame_chroma(uint8_t *plane, ptrdiff_t stride, int mb_w, int mb_h, const FfoH264Edge *edges)
{
    for (int my = 0; my < mb_h; my++)
        for (int mx = 0; mx < mb_w; mx++) {
            const FfoH264Edge *e = edges + (size_t)(my * mb_w + mx) * 4;
            uint8_t *mb = plane + (ptrdiff_t)my * 8 * stride + mx * 8;
            for (int dir = 0; dir < 2; dir++)
                for (int k = 0; k < 2; k++) {
                    const FfoH264Edge *ed = e + dir * 2 + k;
                    uint8_t *pix = dir ? mb + (ptrdiff_t)4 * k * stride : mb + 4 * k;
                    if (!ed->alpha || !ed->beta)
                        continue;
                    if (k == 0 && (dir ? my == 0 : mx == 0))
                        continue;
                    const int intra = ed->kind >= 4;
                    /* dir 0: vertical edge -> h_loop_filter_chroma (3/7); dir 1: horizontal -> v_ (2/6) */
                    ffo_h264_loop_filter((dir ? 2 : 3) + (intra ? 4 : 0), pix, stride, ed->alpha, ed->beta, ed->tc0);
                }
        }
}

/* ------------------------------------------------------------------------------------------
 * Luma quarter-pel MC: libavcodec/h264qpel_template.c:77-305 (6-tap lowpass), :313-459 (mcXY
 * compositions), :461-465 (rounding), hpel_template.c/pel_template.c (rnd_avg).
 * Stated per output sample instead of through temporaries.
 * ---------------------------------------------------------------------------------------- */
static inline int tap6(int a, int b, int c, int d, int e, int f) { return (c + d) * 20 - (b + e) * 5 + (a + f); }
static inline int Hraw(const uint8_t *s) { return tap6(s[-2], s[-1], s[0], s[1], s[2], s[3]); }
static inline int Fp(const uint8_t *s) { return s[0]; }
static inline int Hp(const uint8_t *s) { return clip_u8((Hraw(s) + 16) >> 5); }
static inline int Vp(const uint8_t *s, ptrdiff_t st)
{
    return clip_u8((tap6(s[-2 * st], s[-st], s[0], s[st], s[2 * st], s[3 * st]) + 16) >> 5);
}
static inline int Jp(const uint8_t *s, ptrdiff_t st)
{
    return clip_u8((tap6(Hraw(s - 2 * st), Hraw(s - st), Hraw(s), Hraw(s + st), Hraw(s + 2 * st), Hraw(s + 3 * st)) +
                    512) >> 10);
}
static inline int A2(int a, int b) { return (a + b + 1) >> 1; }

void ffo_h264_qpel(int avg, int size_idx, int mcxy, uint8_t *dst, const uint8_t *src, ptrdiff_t stride)
{
    const int n = 16 >> size_idx;
    uint8_t out[16 * 16];
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            const uint8_t *s = src + y * stride + x;
            int v;
            switch (mcxy) {
            case 0:  v = Fp(s); break;                                   /* mc00 */
            case 1:  v = A2(Fp(s), Hp(s)); break;                        /* mc10 */
            case 2:  v = Hp(s); break;                                   /* mc20 */
            case 3:  v = A2(Fp(s + 1), Hp(s)); break;                    /* mc30 */
            case 4:  v = A2(Fp(s), Vp(s, stride)); break;                /* mc01 */
            case 5:  v = A2(Hp(s), Vp(s, stride)); break;                /* mc11 */
            case 6:  v = A2(Hp(s), Jp(s, stride)); break;                /* mc21 */
            case 7:  v = A2(Hp(s), Vp(s + 1, stride)); break;            /* mc31 */
            case 8:  v = Vp(s, stride); break;                           /* mc02 */
            case 9:  v = A2(Vp(s, stride), Jp(s, stride)); break;        /* mc12 */
            case 10: v = Jp(s, stride); break;                           /* mc22 */
            case 11: v = A2(Vp(s + 1, stride), Jp(s, stride)); break;    /* mc32 */
            case 12: v = A2(Fp(s + stride), Vp(s, stride)); break;       /* mc03 */
            case 13: v = A2(Hp(s + stride), Vp(s, stride)); break;       /* mc13 */
            case 14: v = A2(Hp(s + stride), Jp(s, stride)); break;       /* mc23 */
            default: v = A2(Hp(s + stride), Vp(s + 1, stride)); break;   /* mc33 */
            }
            out[y * 16 + x] = (uint8_t)v;
        }
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[y * stride + x] = avg ? A2(dst[y * stride + x], out[y * 16 + x]) : out[y * 16 + x];
}

/* ------------------------------------------------------------------------------------------
 * Chroma 1/8-pel bilinear MC: libavcodec/h264chroma_template.c:28-172 (op_put/op_avg :169-170).
 * x, y in [0,8).  A=(8-x)(8-y) B=x(8-y) C=(8-x)y D=xy; the reference only reads the samples whose
 * weight is non-zero (three cases), and so do we.
 * ---------------------------------------------------------------------------------------- */
void ffo_h264_chroma_mc(int avg, int w, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h, int x, int y)
{
    const int A = (8 - x) * (8 - y), B = x * (8 - y), C = (8 - x) * y, D = x * y;
    for (int i = 0; i < h; i++, dst += stride, src += stride)
        for (int k = 0; k < w; k++) {
            int v;
            if (D)
                v = A * src[k] + B * src[k + 1] + C * src[stride + k] + D * src[stride + k + 1];
            else if (B + C)
                v = A * src[k] + (B + C) * src[(C ? stride : 1) + k];
            else
                v = A * src[k];
            v = (v + 32) >> 6;
            dst[k] = (uint8_t)(avg ? (dst[k] + v + 1) >> 1 : v);
        }
}

/* Explicit weighted prediction: libavcodec/h264dsp_template.c:30-100 (H264_WEIGHT), 8-bit. */
void ffo_h264_weight(int w, uint8_t *block, ptrdiff_t stride, int height, int log2_denom, int weight, int offset)
{
    offset = (int)((unsigned)offset << log2_denom);
    if (log2_denom)
        offset += 1 << (log2_denom - 1);
    for (int y = 0; y < height; y++, block += stride)
        for (int x = 0; x < w; x++)
            block[x] = clip_u8((block[x] * weight + offset) >> log2_denom);
}

void ffo_h264_biweight(int w, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd,
                       int weights, int offset)
{
    offset = (int)((unsigned)((offset + 1) | 1) << log2_denom);
    for (int y = 0; y < height; y++, dst += stride, src += stride)
        for (int x = 0; x < w; x++)
            dst[x] = clip_u8((src[x] * weights + dst[x] * weightd + offset) >> (log2_denom + 1));
}

/* ------------------------------------------------------------------------------------------
 * hl_decode_mb(), the IS_INTRA branch, SIMPLE / 8 bits / 4:2:0 / frame macroblock / no transform bypass
 * (libavcodec/h264_mb_template.c:137-262 with hl_decode_mb_predict_luma and hl_decode_mb_idct_luma,
 * libavcodec/h264_mb.c:612-760): the dsp calls one intra macroblock makes, in the reference's order, on the oracle's restatements
 * of those members.  Arguments are the H264SliceContext fields the reference reads: intra4x4_pred_mode[i] =
 * sl->intra4x4_pred_mode_cache[scan8[i]], nnzc = sl->non_zero_count_cache (15 x 8), mb = sl->mb (3 x 256), qmul[3] =
 * dequant4_coeff[0][qscale][0], [1][chroma_qp[0]][0], [2][chroma_qp[1]][0]; type as FFHIP_H264_INTRA_* (include/ffhip.h).
 * ---------------------------------------------------------------------------------------- */
void ffo_h264_hl_decode_intra_mb(uint8_t *dest_y, uint8_t *dest_cb, uint8_t *dest_cr, ptrdiff_t linesize, ptrdiff_t uvlinesize,
                                 int type, int intra16x16_pred_mode, int chroma_pred_mode, const uint8_t *intra4x4_pred_mode,
                                 unsigned topleft_samples_available, unsigned topright_samples_available, const uint8_t *nnzc,
                                 int cbp, int16_t *mb, int16_t *mb_luma_dc, const int *qmul, const uint8_t *intra_pcm_ptr)
{
    int block_offset[48];
    /* h264_slice.c init: block_offset[i] luma, [16 + i] / [32 + i] chroma (4:2:0: only the first four of each are used) */
    for (int i = 0; i < 16; i++) {
        const int x = 4 * ((i & 1) + ((i >> 2) & 1) * 2), y = 4 * (((i >> 1) & 1) + ((i >> 3) & 1) * 2);
        block_offset[i] = x + y * (int)linesize;
        block_offset[16 + i] = block_offset[32 + i] = x + y * (int)uvlinesize;
    }
    if (type == 3) { /* IS_INTRA_PCM, h264_mb_template.c:137-150 */
        for (int i = 0; i < 16; i++)
            memcpy(dest_y + i * linesize, intra_pcm_ptr + i * 16, 16);
        for (int i = 0; i < 8; i++) {
            memcpy(dest_cb + i * uvlinesize, intra_pcm_ptr + 256 + i * 8, 8);
            memcpy(dest_cr + i * uvlinesize, intra_pcm_ptr + 256 + 64 + i * 8, 8);
        }
        return;
    }
    ffo_h264_pred8x8(chroma_pred_mode, dest_cb, uvlinesize);
    ffo_h264_pred8x8(chroma_pred_mode, dest_cr, uvlinesize);
    /* hl_decode_mb_predict_luma */
    if (type == 2) {
        for (int i = 0; i < 16; i += 4) {
            uint8_t *ptr = dest_y + block_offset[i];
            const int dir = intra4x4_pred_mode[i], nnz = nnzc[scan8_luma[i]];
            ffo_h264_pred8x8l(dir, ptr, (topleft_samples_available << i) & 0x8000, (topright_samples_available << i) & 0x4000, linesize);
            if (nnz) {
                if (nnz == 1 && mb[i * 16])
                    ffo_h264_idct8_dc_add(ptr, mb + i * 16, linesize);
                else
                    ffo_h264_idct8_add(ptr, mb + i * 16, linesize);
            }
        }
    } else if (type == 1) {
        for (int i = 0; i < 16; i++) {
            uint8_t *ptr = dest_y + block_offset[i];
            const int dir = intra4x4_pred_mode[i], nnz = nnzc[scan8_luma[i]];
            uint32_t tr;
            const uint8_t *topright = NULL;
            if (dir == 3 || dir == 7) { /* DIAG_DOWN_LEFT_PRED, VERT_LEFT_PRED */
                if (!((topright_samples_available << i) & 0x8000)) {
                    tr = ptr[3 - linesize] * 0x01010101u;
                    topright = (const uint8_t *)&tr;
                } else {
                    topright = ptr + 4 - linesize;
                }
            }
            ffo_h264_pred4x4(dir, ptr, topright, linesize);
            if (nnz) {
                if (nnz == 1 && mb[i * 16])
                    ffo_h264_idct_dc_add(ptr, mb + i * 16, linesize);
                else
                    ffo_h264_idct_add(ptr, mb + i * 16, linesize);
            }
        }
    } else {
        ffo_h264_pred16x16(intra16x16_pred_mode, dest_y, linesize);
        if (nnzc[0]) /* scan8[LUMA_DC_BLOCK_INDEX] */
            ffo_h264_luma_dc_dequant_idct(mb, mb_luma_dc, qmul[0]);
        /* hl_decode_mb_idct_luma */
        ffo_h264_idct_add16intra(dest_y, block_offset, mb, linesize, nnzc);
    }
    if (cbp & 0x30) {
        uint8_t *dest[2] = { dest_cb, dest_cr };
        if (nnzc[5 * 8]) /* scan8[CHROMA_DC_BLOCK_INDEX + 0] */
            ffo_h264_chroma_dc_dequant_idct(mb + 256, qmul[1]);
        if (nnzc[10 * 8])
            ffo_h264_chroma_dc_dequant_idct(mb + 512, qmul[2]);
        ffo_h264_idct_add8(dest, block_offset, mb, uvlinesize, nnzc);
    }
}
