/*
 * h264_hbd.hip — the H.264 DSP tables above 8 bits (9 / 10 / 12 / 14: every depth the reference instantiates, libavcodec/h264dsp.c:
 * 135-147, h264qpel.c:87-103, h264chroma.c:38-52) and the members the 8-bit kernels do not cover at any depth: the MBAFF loop
 * filters, the 4:2:2 chroma forms (h_loop_filter_chroma422*, idct_add8_422, chroma422_dc_dequant_idct).
 *
 * Bit-exact restatement of the reference's templates with the depth as a kernel argument (libavcodec/bit_depth_template.c: pixel =
 * uint16_t, dctcoef = int32_t above 8 bits; strides and offsets stay in BYTES; av_clip_pixel clips to (1 << depth) - 1):
 *   h264idct_template.c:33-175, h264addpx_template.c:30-74           idct_add / idct8_add / *_dc_add / add_pixels{4,8}_clear
 *   h264idct_template.c:264-352                                      luma / chroma / chroma422 dc_dequant_idct
 *   h264dsp_template.c:104-330                                       the loop-filter family (one generic line filter)
 *   h264qpel_template.c:77-465                                       put / avg x 16 / 8 / 4 x 16 quarter-pel positions
 *   h264chroma_template.c:28-190, h264dsp_template.c:30-98           chroma MC, explicit weighted prediction
 * Templates on the sample type, so the 8-bit instantiation serves the MBAFF / 4:2:2 members of 8-bit streams.
 *
 * Shape: block lists as in the 8-bit faces.  IDCT: one lane per block, both passes in registers, 16-byte coefficient loads
 * (a 4x4 block of int32 is four of them) and 8-byte picture rows.  Loop filter: 16 lanes per edge, a lane per sample line.
 * Motion compensation / weighting: a lane per output sample row segment of 4, taps from global memory (the reference rows of a
 * block are re-read by its lanes through L1/L2; the 8-bit kernels' LDS staging is not repeated here: the high-depth path is
 * bandwidth-light next to its arithmetic).
 */
#include "common.h"
#include "h264_kernels.h"

namespace {

template <typename P> struct HbdCoef { typedef int16_t T; };
template <> struct HbdCoef<uint16_t> { typedef int32_t T; };

__device__ __forceinline__ int hclip(int v, int maxv) { return min(max(v, 0), maxv); }

/* ---- IDCT ------------------------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ void hbd_idct8_1d(const int (&in)[8], uint32_t (&out)[8])
{
    const uint32_t a0 = (uint32_t)in[0] + (uint32_t)in[4], a2 = (uint32_t)in[0] - (uint32_t)in[4];
    const uint32_t a4 = (uint32_t)(in[2] >> 1) - (uint32_t)in[6], a6 = (uint32_t)(in[6] >> 1) + (uint32_t)in[2];
    const uint32_t b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    const int a1 = (int)(-(uint32_t)in[3] + (uint32_t)in[5] - (uint32_t)in[7] - (uint32_t)(in[7] >> 1));
    const int a3 = (int)((uint32_t)in[1] + (uint32_t)in[7] - (uint32_t)in[3] - (uint32_t)(in[3] >> 1));
    const int a5 = (int)(-(uint32_t)in[1] + (uint32_t)in[7] + (uint32_t)in[5] + (uint32_t)(in[5] >> 1));
    const int a7 = (int)((uint32_t)in[3] + (uint32_t)in[5] + (uint32_t)in[1] + (uint32_t)(in[1] >> 1));
    const uint32_t b1 = (uint32_t)(a7 >> 2) + (uint32_t)a1, b3 = (uint32_t)a3 + (uint32_t)(a5 >> 2);
    const uint32_t b5 = (uint32_t)(a3 >> 2) - (uint32_t)a5, b7 = (uint32_t)a7 - (uint32_t)(a1 >> 2);
    out[0] = b0 + b7; out[7] = b0 - b7; out[1] = b2 + b5; out[6] = b2 - b5;
    out[2] = b4 + b3; out[5] = b4 - b3; out[3] = b6 + b1; out[4] = b6 - b1;
}

/* one block: kind FFHIP_H264_IDCT4 .. ADD_PIXELS8_CLEAR; c = its N*N coefficients (cleared as the reference clears them) */
template <typename P, int N>
__device__ __forceinline__ void hbd_block(int kind, P *dst, ptrdiff_t s, typename HbdCoef<P>::T *c, int maxv)
{
    typedef typename HbdCoef<P>::T CF;
    if (kind == FFHIP_H264_IDCT4_DC || kind == FFHIP_H264_IDCT8_DC) {
        const int dc = ((int)c[0] + 32) >> 6;
        c[0] = 0;
#pragma unroll
        for (int y = 0; y < N; y++)
#pragma unroll
            for (int x = 0; x < N; x++)
                dst[y * s + x] = (P)hclip((int)dst[y * s + x] + dc, maxv);
        return;
    }
    int v[N][N]; /* v[row][col] */
#pragma unroll
    for (int i = 0; i < N * N; i++)
        v[i / N][i % N] = (int)c[i];
#pragma unroll
    for (int i = 0; i < N * N; i++)
        c[i] = 0;
    if (kind == FFHIP_H264_ADD_PIXELS4_CLEAR || kind == FFHIP_H264_ADD_PIXELS8_CLEAR) {
#pragma unroll
        for (int y = 0; y < N; y++)
#pragma unroll
            for (int x = 0; x < N; x++)
                dst[y * s + x] = (P)((unsigned)dst[y * s + x] + (unsigned)v[y][x]); /* the sample type's wrap-around, no clip */
        return;
    }
    v[0][0] += 32;
    if (N == 4) {
        /* pass 1 over block[i + 4 k] (k = 0..3: down a column of the stored 4x4), results stored back through the coefficient type */
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t z0 = (uint32_t)v[0][i] + (uint32_t)v[2][i], z1 = (uint32_t)v[0][i] - (uint32_t)v[2][i];
            const uint32_t z2 = (uint32_t)(v[1][i] >> 1) - (uint32_t)v[3][i], z3 = (uint32_t)v[1][i] + (uint32_t)(v[3][i] >> 1);
            v[0][i] = (CF)(z0 + z3); v[1][i] = (CF)(z1 + z2); v[2][i] = (CF)(z1 - z2); v[3][i] = (CF)(z0 - z3);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { /* pass 2 over block[4 i + k]: its outputs go DOWN column i of dst */
            const uint32_t z0 = (uint32_t)v[i][0] + (uint32_t)v[i][2], z1 = (uint32_t)v[i][0] - (uint32_t)v[i][2];
            const uint32_t z2 = (uint32_t)(v[i][1] >> 1) - (uint32_t)v[i][3], z3 = (uint32_t)v[i][1] + (uint32_t)(v[i][3] >> 1);
            dst[i] = (P)hclip((int)dst[i] + ((int)(z0 + z3) >> 6), maxv);
            dst[i + s] = (P)hclip((int)dst[i + s] + ((int)(z1 + z2) >> 6), maxv);
            dst[i + 2 * s] = (P)hclip((int)dst[i + 2 * s] + ((int)(z1 - z2) >> 6), maxv);
            dst[i + 3 * s] = (P)hclip((int)dst[i + 3 * s] + ((int)(z0 - z3) >> 6), maxv);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int in[8];
            uint32_t out[8];
#pragma unroll
            for (int k = 0; k < 8; k++) in[k] = v[k % N][i % N];
            hbd_idct8_1d(in, out);
#pragma unroll
            for (int k = 0; k < 8; k++) v[k % N][i % N] = (CF)out[k];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int in[8];
            uint32_t out[8];
#pragma unroll
            for (int k = 0; k < 8; k++) in[k] = v[i % N][k % N];
            hbd_idct8_1d(in, out);
#pragma unroll
            for (int k = 0; k < 8; k++)
                dst[i + k * s] = (P)hclip((int)dst[i + k * s] + ((int)out[k] >> 6), maxv);
        }
    }
}

template <typename P>
__global__ __launch_bounds__(64) void k_h264_idct_hbd(int kind, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset, int16_t *blocks,
                                                     int n, int bd)
{
    typedef typename HbdCoef<P>::T CF;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n)
        return;
    P *dst = reinterpret_cast<P *>(dst_base + dst_offset[i]);
    const ptrdiff_t s = stride / (ptrdiff_t)sizeof(P);
    const int maxv = (1 << bd) - 1;
    const bool is8 = kind == FFHIP_H264_IDCT8 || kind == FFHIP_H264_IDCT8_DC || kind == FFHIP_H264_ADD_PIXELS8_CLEAR;
    CF *c = reinterpret_cast<CF *>(blocks) + (size_t)i * (is8 ? 64 : 16);
    if (is8)
        hbd_block<P, 8>(kind, dst, s, c, maxv);
    else
        hbd_block<P, 4>(kind, dst, s, c, maxv);
}

/* the macroblock dispatchers (h264idct_template.c:177-262) over nmb macroblocks: lane = (macroblock, block).
 * which 0 idct_add16, 1 idct8_add4, 2 idct_add16intra, 3 idct_add8 (4:2:0: blocks 16..19 / 32..35), 4 idct_add8_422 (+ 20..23 / 36..39
 * whose offsets / nnz entries sit four slots later).  For 3 / 4 `dst_base` is Cb and `dst2` Cr. */
__constant__ uint8_t hbd_scan8[48] = {
    4 + 1 * 8, 5 + 1 * 8, 4 + 2 * 8, 5 + 2 * 8, 6 + 1 * 8, 7 + 1 * 8, 6 + 2 * 8, 7 + 2 * 8,
    4 + 3 * 8, 5 + 3 * 8, 4 + 4 * 8, 5 + 4 * 8, 6 + 3 * 8, 7 + 3 * 8, 6 + 4 * 8, 7 + 4 * 8,
    4 + 6 * 8, 5 + 6 * 8, 4 + 7 * 8, 5 + 7 * 8, 6 + 6 * 8, 7 + 6 * 8, 6 + 7 * 8, 7 + 7 * 8,
    4 + 8 * 8, 5 + 8 * 8, 4 + 9 * 8, 5 + 9 * 8, 6 + 8 * 8, 7 + 8 * 8, 6 + 9 * 8, 7 + 9 * 8,
    4 + 11 * 8, 5 + 11 * 8, 4 + 12 * 8, 5 + 12 * 8, 6 + 11 * 8, 7 + 11 * 8, 6 + 12 * 8, 7 + 12 * 8,
    4 + 13 * 8, 5 + 13 * 8, 4 + 14 * 8, 5 + 14 * 8, 6 + 13 * 8, 7 + 13 * 8, 6 + 14 * 8, 7 + 14 * 8,
};

template <typename P>
__global__ __launch_bounds__(64) void k_h264_idct_mb_hbd(int which, uint8_t *dst_base, uint8_t *dst2, ptrdiff_t stride, const int32_t *mb_offset,
                                                        const int32_t *blockoffset, int16_t *blocks, const uint8_t *nnzc, int nmb, int bd)
{
    typedef typename HbdCoef<P>::T CF;
    const int per = which == 1 ? 4 : which == 3 ? 8 : 16; /* lanes per macroblock */
    const int t = blockIdx.x * 64 + threadIdx.x, m = t / per, k = t % per;
    if (m >= nmb)
        return;
    const int ncoef = which >= 3 ? 768 : 256, nnz_rows = which >= 3 ? 120 : 40;
    const uint8_t *nn = nnzc + (size_t)m * nnz_rows;
    CF *mbc = reinterpret_cast<CF *>(blocks) + (size_t)m * ncoef;
    const ptrdiff_t s = stride / (ptrdiff_t)sizeof(P);
    const int maxv = (1 << bd) - 1;
    uint8_t *base = dst_base;
    int blk, slot; /* coefficient block index, index into blockoffset / scan8 */
    if (which == 1) { blk = slot = 4 * k; }
    else if (which < 3) { blk = slot = k; }
    else {
        const int j = which == 3 ? k >> 2 : k >> 3, r = which == 3 ? k & 3 : k & 7; /* plane, block of the plane */
        base = j ? dst2 : dst_base;
        blk = 16 * (j + 1) + r;
        slot = blk + (r >= 4 ? 4 : 0);
    }
    const int nnz = nn[hbd_scan8[slot]];
    CF *c = mbc + blk * 16;
    P *dst = reinterpret_cast<P *>(base + mb_offset[m] + blockoffset[slot]);
    if (which == 1) {
        if (nnz) {
            if (nnz == 1 && c[0]) hbd_block<P, 8>(FFHIP_H264_IDCT8_DC, dst, s, c, maxv);
            else hbd_block<P, 8>(FFHIP_H264_IDCT8, dst, s, c, maxv);
        }
    } else if (which == 0) {
        if (nnz) {
            if (nnz == 1 && c[0]) hbd_block<P, 4>(FFHIP_H264_IDCT4_DC, dst, s, c, maxv);
            else hbd_block<P, 4>(FFHIP_H264_IDCT4, dst, s, c, maxv);
        }
    } else {
        if (nnz) hbd_block<P, 4>(FFHIP_H264_IDCT4, dst, s, c, maxv);
        else if (c[0]) hbd_block<P, 4>(FFHIP_H264_IDCT4_DC, dst, s, c, maxv);
    }
}

/* DC transforms: which 0 luma (16 values of input + m*in_pitch -> the DC positions of output + m*out_pitch), 1 chroma 4:2:0,
 * 2 chroma 4:2:2 (in place on blocks + block_offset[m]).  Pitches / offsets in coefficients. */
template <typename CF>
__global__ __launch_bounds__(64) void k_h264_dc_dequant_hbd(int which, CF *output, size_t out_pitch, const CF *input, size_t in_pitch,
                                                           const int32_t *block_offset, const int32_t *qmul, int n)
{
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= n)
        return;
    const uint32_t q = (uint32_t)qmul[m];
    if (which == 0) {
        const CF *in = input + (size_t)m * in_pitch;
        CF *out = output + (size_t)m * out_pitch;
        int temp[16];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int z0 = (int)in[4 * i] + (int)in[4 * i + 1], z1 = (int)in[4 * i] - (int)in[4 * i + 1];
            const int z2 = (int)in[4 * i + 2] - (int)in[4 * i + 3], z3 = (int)in[4 * i + 2] + (int)in[4 * i + 3];
            temp[4 * i] = z0 + z3; temp[4 * i + 1] = z0 - z3; temp[4 * i + 2] = z1 - z2; temp[4 * i + 3] = z1 + z2;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int o = i == 0 ? 0 : i == 1 ? 32 : i == 2 ? 128 : 160;
            const uint32_t z0 = (uint32_t)temp[i] + (uint32_t)temp[8 + i], z1 = (uint32_t)temp[i] - (uint32_t)temp[8 + i];
            const uint32_t z2 = (uint32_t)temp[4 + i] - (uint32_t)temp[12 + i], z3 = (uint32_t)temp[4 + i] + (uint32_t)temp[12 + i];
            out[o] = (CF)((int)((z0 + z3) * q + 128) >> 8);
            out[16 + o] = (CF)((int)((z1 + z2) * q + 128) >> 8);
            out[64 + o] = (CF)((int)((z1 - z2) * q + 128) >> 8);
            out[80 + o] = (CF)((int)((z0 - z3) * q + 128) >> 8);
        }
        return;
    }
    CF *b = output + block_offset[m];
    if (which == 1) {
        uint32_t a = (uint32_t)(int)b[0], bb = (uint32_t)(int)b[16], c = (uint32_t)(int)b[32], d = (uint32_t)(int)b[48];
        const uint32_t e = a - bb;
        a = a + bb;
        bb = c - d;
        c = c + d;
        b[0] = (CF)((int)((a + c) * q) >> 7);
        b[16] = (CF)((int)((e + bb) * q) >> 7);
        b[32] = (CF)((int)((a - c) * q) >> 7);
        b[48] = (CF)((int)((e - bb) * q) >> 7);
        return;
    }
    uint32_t temp[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        temp[2 * i] = (uint32_t)(int)b[32 * i] + (uint32_t)(int)b[32 * i + 16];
        temp[2 * i + 1] = (uint32_t)(int)b[32 * i] - (uint32_t)(int)b[32 * i + 16];
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int o = 16 * i;
        const uint32_t z0 = temp[i] + temp[4 + i], z1 = temp[i] - temp[4 + i], z2 = temp[2 + i] - temp[6 + i], z3 = temp[2 + i] + temp[6 + i];
        b[o] = (CF)((int)((z0 + z3) * q + 128) >> 8);
        b[32 + o] = (CF)((int)((z1 + z2) * q + 128) >> 8);
        b[64 + o] = (CF)((int)((z1 - z2) * q + 128) >> 8);
        b[96 + o] = (CF)((int)((z0 - z3) * q + 128) >> 8);
    }
}

/* ---- loop filter: 16 lanes per edge, lane = sample line.  FFHipH264Edge.pad = lines per tc0 entry (0: luma 4, chroma 2) ------------ */
template <typename P>
__global__ __launch_bounds__(256) void k_h264_loop_filter_hbd(uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n, int bd,
                                                             const int32_t *ab)
{
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 4, d = threadIdx.x & 15;
    if (e >= n)
        return;
    const FFHipH264Edge ed = edges[e];
    const int kind = ed.kind & 7;
    const bool chroma = kind & 2, intra = kind & 4, vert_edge = kind & 1;
    const int inner = ed.pad ? ed.pad : chroma ? 2 : 4;
    if (d >= 4 * inner)
        return;
    const ptrdiff_t s = stride / (ptrdiff_t)sizeof(P);
    const ptrdiff_t xs = vert_edge ? 1 : s, ys = vert_edge ? s : 1;
    P *pix = reinterpret_cast<P *>(base + ed.offset) + d * ys;
    const int maxv = (1 << bd) - 1, sh = bd - 8;
    /* the record holds alpha / beta as the decoder's tables do (bytes); the host faces pass the caller's ints through `ab`, whatever
     * they are (checkasm hands the functions values far outside the tables' range) */
    const int alpha = (int)((unsigned)(ab ? ab[2 * e] : (int)ed.alpha) << sh), beta = (int)((unsigned)(ab ? ab[2 * e + 1] : (int)ed.beta) << sh);
    const int p0 = pix[-xs], p1 = pix[-2 * xs], q0 = pix[0], q1 = pix[xs];
    if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta))
        return;
    if (intra) {
        if (chroma) {
            pix[-xs] = (P)((2 * p1 + p0 + q1 + 2) >> 2);
            pix[0] = (P)((2 * q1 + q0 + p1 + 2) >> 2);
            return;
        }
        const int p2 = pix[-3 * xs], q2 = pix[2 * xs];
        if (abs(p0 - q0) < ((alpha >> 2) + 2)) {
            if (abs(p2 - p0) < beta) {
                const int p3 = pix[-4 * xs];
                pix[-xs] = (P)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                pix[-2 * xs] = (P)((p2 + p1 + p0 + q0 + 2) >> 2);
                pix[-3 * xs] = (P)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            } else
                pix[-xs] = (P)((2 * p1 + p0 + q1 + 2) >> 2);
            if (abs(q2 - q0) < beta) {
                const int q3 = pix[3 * xs];
                pix[0] = (P)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                pix[xs] = (P)((p0 + q0 + q1 + q2 + 2) >> 2);
                pix[2 * xs] = (P)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
            } else
                pix[0] = (P)((2 * q1 + q0 + p1 + 2) >> 2);
        } else {
            pix[-xs] = (P)((2 * p1 + p0 + q1 + 2) >> 2);
            pix[0] = (P)((2 * q1 + q0 + p1 + 2) >> 2);
        }
        return;
    }
    const int t0 = ed.tc0[d / inner];
    if (chroma) {
        const int tc = (int)(((unsigned)t0 - 1U) << sh) + 1;
        if (tc <= 0)
            return;
        const int delta = clip3(((q0 - p0) * 4 + (p1 - q1) + 4) >> 3, -tc, tc);
        pix[-xs] = (P)hclip(p0 + delta, maxv);
        pix[0] = (P)hclip(q0 - delta, maxv);
        return;
    }
    const int tc_orig = t0 * (1 << sh);
    if (tc_orig < 0)
        return;
    const int p2 = pix[-3 * xs], q2 = pix[2 * xs];
    int tc = tc_orig;
    if (abs(p2 - p0) < beta) {
        if (tc_orig)
            pix[-2 * xs] = (P)(p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc_orig, tc_orig));
        tc++;
    }
    if (abs(q2 - q0) < beta) {
        if (tc_orig)
            pix[xs] = (P)(q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc_orig, tc_orig));
        tc++;
    }
    const int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    pix[-xs] = (P)hclip(p0 + delta, maxv);
    pix[0] = (P)hclip(q0 - delta, maxv);
}

/* ---- luma qpel: a lane per output sample (16 lanes per row of a 16-wide block) --------------------------------------------- */
__device__ __forceinline__ int hbd_tap6(int a, int b, int c, int d, int e, int f) { return (c + d) * 20 - (b + e) * 5 + (a + f); }

/* The reference samples around one output sample: at(dx, dy).  A record flagged FFHIP_MC_EMU (include/ffhip.h; h264_mb.c:229-247,
 * 297-317 -> videodsp_template.c:24-100) reads them at clamped coordinates of the reference picture whose (0, 0) is `p`. */
template <typename P>
struct HbdSrc {
    const P *p;      /* plain: the sample under the output sample; emu: the picture's (0, 0) */
    ptrdiff_t s;     /* row pitch in samples */
    int x, y, pw, ph;
    bool emu;
    __device__ __forceinline__ int at(int dx, int dy) const
    {
        if (emu)
            return (int)p[(ptrdiff_t)min(max(y + dy, 0), ph - 1) * s + min(max(x + dx, 0), pw - 1)];
        return (int)p[dy * s + dx];
    }
    __device__ __forceinline__ int h(int dx, int dy) const { return hbd_tap6(at(dx - 2, dy), at(dx - 1, dy), at(dx, dy), at(dx + 1, dy), at(dx + 2, dy), at(dx + 3, dy)); }
    __device__ __forceinline__ int v(int dx, int dy) const { return hbd_tap6(at(dx, dy - 2), at(dx, dy - 1), at(dx, dy), at(dx, dy + 1), at(dx, dy + 2), at(dx, dy + 3)); }
};

template <typename P>
__global__ __launch_bounds__(256) void k_h264_qpel_hbd(uint8_t *dst_base, const uint8_t *src_base, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                                                      int bd, int pic_w, int pic_h)
{
    const int b = blockIdx.x, t = threadIdx.x; /* one workgroup per block, 256 lanes = 16 x 16 samples */
    if (b >= n)
        return;
    const FFHipQpelBlock bl = blocks[b];
    const int sz = 16 >> bl.size_idx, x = t & 15, y = t >> 4;
    if (x >= sz || y >= sz)
        return;
    const ptrdiff_t s = stride / (ptrdiff_t)sizeof(P);
    HbdSrc<P> S;
    S.s = s; S.pw = pic_w; S.ph = pic_h;
    S.emu = pic_w > 0 && (bl.flags & FFHIP_MC_EMU);
    S.x = bl.src_x + x; S.y = bl.src_y + y;
    S.p = reinterpret_cast<const P *>(src_base + bl.src_offset) + (S.emu ? 0 : y * s + x);
    P *dst = reinterpret_cast<P *>(dst_base + bl.dst_offset) + y * s + x;
    const int maxv = (1 << bd) - 1, mc = bl.mcxy & 15, mx = mc & 3, my = mc >> 2;
#define QH(dx, dy) hclip((S.h(dx, dy) + 16) >> 5, maxv)
#define QV(dx, dy) hclip((S.v(dx, dy) + 16) >> 5, maxv)
#define QA(a, b) (((a) + (b) + 1) >> 1)
    int hv = 0;
    if (mx == 2 || my == 2) {
        if ((mx == 2 && my == 2) || (mx != 0 && my != 0)) {
            int tt[6];
#pragma unroll
            for (int k = 0; k < 6; k++)
                tt[k] = S.h(0, k - 2);
            hv = hclip((hbd_tap6(tt[0], tt[1], tt[2], tt[3], tt[4], tt[5]) + 512) >> 10, maxv);
        }
    }
    int v;
    switch (mc) {
    case 0:  v = S.at(0, 0); break;
    case 1:  v = QA(S.at(0, 0), QH(0, 0)); break;
    case 2:  v = QH(0, 0); break;
    case 3:  v = QA(S.at(1, 0), QH(0, 0)); break;
    case 4:  v = QA(S.at(0, 0), QV(0, 0)); break;
    case 8:  v = QV(0, 0); break;
    case 12: v = QA(S.at(0, 1), QV(0, 0)); break;
    case 5:  v = QA(QH(0, 0), QV(0, 0)); break;
    case 7:  v = QA(QH(0, 0), QV(1, 0)); break;
    case 13: v = QA(QH(0, 1), QV(0, 0)); break;
    case 15: v = QA(QH(0, 1), QV(1, 0)); break;
    case 10: v = hv; break;
    case 6:  v = QA(QH(0, 0), hv); break;
    case 14: v = QA(QH(0, 1), hv); break;
    case 9:  v = QA(QV(0, 0), hv); break;
    default: v = QA(QV(1, 0), hv); break;
    }
    dst[0] = (P)(bl.avg ? QA((int)dst[0], v) : v);
#undef QH
#undef QV
}

/* ---- chroma MC and explicit weighting: a lane per output sample ---------------------------------------------------------------- */
template <typename P>
__global__ __launch_bounds__(128) void k_h264_chroma_mc_hbd(uint8_t *dst_base, const uint8_t *src_base, ptrdiff_t stride, const FFHipChromaBlock *blocks,
                                                           int n, int pic_w, int pic_h)
{
    const int b = blockIdx.x, t = threadIdx.x; /* up to 8 wide x 16 rows */
    if (b >= n)
        return;
    const FFHipChromaBlock bl = blocks[b];
    const int w = 8 >> bl.w_idx, x = t & 7, y = t >> 3;
    if (x >= w || y >= bl.h)
        return;
    const ptrdiff_t s = stride / (ptrdiff_t)sizeof(P);
    HbdSrc<P> S;
    S.s = s; S.pw = pic_w; S.ph = pic_h;
    S.emu = pic_w > 0 && (bl.flags & FFHIP_MC_EMU);
    S.x = bl.src_x + x; S.y = bl.src_y + y;
    S.p = reinterpret_cast<const P *>(src_base + bl.src_offset) + (S.emu ? 0 : y * s + x);
    P *dst = reinterpret_cast<P *>(dst_base + bl.dst_offset) + y * s + x;
    const int fx = bl.x, fy = bl.y, A = (8 - fx) * (8 - fy), B = fx * (8 - fy), Cc = (8 - fx) * fy, D = fx * fy;
    int v = A * S.at(0, 0);
    if (B) v += B * S.at(1, 0);        /* the template never reads a neighbour whose weight is zero */
    if (Cc) v += Cc * S.at(0, 1);
    if (D) v += D * S.at(1, 1);
    v = (v + 32) >> 6;
    dst[0] = (P)(bl.avg ? ((int)dst[0] + v + 1) >> 1 : v);
}

template <typename P>
__global__ __launch_bounds__(256) void k_h264_weight_hbd(uint8_t *dst_base, const uint8_t *src_base, ptrdiff_t stride, const FFHipWeightBlock *blocks, int n,
                                                        int bd)
{
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= n)
        return;
    const FFHipWeightBlock bl = blocks[b];
    const int w = 16 >> bl.w_idx, x = t & 15, y = t >> 4;
    if (x >= w || y >= bl.height)
        return;
    const ptrdiff_t s = stride / (ptrdiff_t)sizeof(P);
    P *dst = reinterpret_cast<P *>(dst_base + bl.dst_offset) + y * s + x;
    const int maxv = (1 << bd) - 1, ld = bl.log2_denom;
    if (!bl.bi) {
        int offset = (int)((unsigned)(int)bl.offset << (ld + (bd - 8)));
        if (ld)
            offset += 1 << (ld - 1);
        dst[0] = (P)hclip(((int)dst[0] * (int)bl.weightd + offset) >> ld, maxv);
    } else {
        const P *src = reinterpret_cast<const P *>(src_base + bl.src_offset) + y * s + x;
        int offset = (int)((unsigned)(int)bl.offset << (bd - 8));
        offset = (int)((unsigned)((offset + 1) | 1) << ld);
        dst[0] = (P)hclip(((int)src[0] * (int)bl.weights + (int)dst[0] * (int)bl.weightd + offset) >> (ld + 1), maxv);
    }
}

bool hbd_depth_ok(int bd) { return bd == 8 || bd == 9 || bd == 10 || bd == 12 || bd == 14; }

} // namespace

#define HBD_CHECK(bd)                                                                                            \
    do {                                                                                                         \
        if (!hbd_depth_ok(bd)) {                                                                                 \
            ffhip_set_error("ffhip_h264: bit depth %d (8, 9, 10, 12 and 14 are the depths H.264 defines)", bd);  \
            return FFHIP_EINVAL;                                                                                 \
        }                                                                                                        \
        if (bd > 8 && (stride & 1)) {                                                                            \
            ffhip_set_error("ffhip_h264: odd byte stride %td with 16-bit samples", (ptrdiff_t)stride);           \
            return FFHIP_EINVAL;                                                                                 \
        }                                                                                                        \
    } while (0)

int ffhip_launch_h264_idct_add_bd(int bd, int kind, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset, int16_t *blocks, int n,
                                  hipStream_t stream)
{
    if (n <= 0)
        return 0;
    HBD_CHECK(bd);
    if (kind < FFHIP_H264_IDCT4 || kind > FFHIP_H264_ADD_PIXELS8_CLEAR)
        return FFHIP_EINVAL;
    if (bd > 8)
        hipLaunchKernelGGL(k_h264_idct_hbd<uint16_t>, dim3(cdiv(n, 64)), dim3(64), 0, stream, kind, dst_base, stride, dst_offset, blocks, n, bd);
    else
        hipLaunchKernelGGL(k_h264_idct_hbd<uint8_t>, dim3(cdiv(n, 64)), dim3(64), 0, stream, kind, dst_base, stride, dst_offset, blocks, n, bd);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_idct_mb_bd(int bd, int which, uint8_t *dst_base, uint8_t *dst2, ptrdiff_t stride, const int32_t *mb_offset,
                                 const int32_t *blockoffset, int16_t *blocks, const uint8_t *nnzc, int nmb, hipStream_t stream)
{
    if (nmb <= 0)
        return 0;
    HBD_CHECK(bd);
    if (which < 0 || which > 4)
        return FFHIP_EINVAL;
    const int per = which == 1 ? 4 : which == 3 ? 8 : 16;
    const dim3 g(cdiv(nmb * per, 64)), t(64);
    if (bd > 8)
        hipLaunchKernelGGL(k_h264_idct_mb_hbd<uint16_t>, g, t, 0, stream, which, dst_base, dst2, stride, mb_offset, blockoffset, blocks, nnzc, nmb, bd);
    else
        hipLaunchKernelGGL(k_h264_idct_mb_hbd<uint8_t>, g, t, 0, stream, which, dst_base, dst2, stride, mb_offset, blockoffset, blocks, nnzc, nmb, bd);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_dc_dequant_bd(int bd, int which, int16_t *output, size_t out_pitch, const int16_t *input, size_t in_pitch,
                                    const int32_t *block_offset, const int32_t *qmul, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    const ptrdiff_t stride = 0;
    HBD_CHECK(bd);
    if (which < 0 || which > 2)
        return FFHIP_EINVAL;
    if (bd > 8)
        hipLaunchKernelGGL(k_h264_dc_dequant_hbd<int32_t>, dim3(cdiv(n, 64)), dim3(64), 0, stream, which, reinterpret_cast<int32_t *>(output), out_pitch,
                           reinterpret_cast<const int32_t *>(input), in_pitch, block_offset, qmul, n);
    else
        hipLaunchKernelGGL(k_h264_dc_dequant_hbd<int16_t>, dim3(cdiv(n, 64)), dim3(64), 0, stream, which, output, out_pitch, input, in_pitch,
                           block_offset, qmul, n);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_loop_filter_bd(int bd, uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n, hipStream_t stream,
                                     const int32_t *alpha_beta)
{
    if (n <= 0)
        return 0;
    HBD_CHECK(bd);
    if (bd > 8)
        hipLaunchKernelGGL(k_h264_loop_filter_hbd<uint16_t>, dim3(cdiv(n, 16)), dim3(256), 0, stream, base, stride, edges, n, bd, alpha_beta);
    else
        hipLaunchKernelGGL(k_h264_loop_filter_hbd<uint8_t>, dim3(cdiv(n, 16)), dim3(256), 0, stream, base, stride, edges, n, bd, alpha_beta);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_qpel_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n, hipStream_t stream,
                              int pic_w, int pic_h)
{
    if (n <= 0)
        return 0;
    HBD_CHECK(bd);
    if (bd > 8)
        hipLaunchKernelGGL(k_h264_qpel_hbd<uint16_t>, dim3(n), dim3(256), 0, stream, dst, src, stride, blocks, n, bd, pic_w, pic_h);
    else
        hipLaunchKernelGGL(k_h264_qpel_hbd<uint8_t>, dim3(n), dim3(256), 0, stream, dst, src, stride, blocks, n, bd, pic_w, pic_h);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_chroma_mc_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks, int n,
                                   hipStream_t stream, int pic_w, int pic_h)
{
    if (n <= 0)
        return 0;
    HBD_CHECK(bd);
    if (bd > 8)
        hipLaunchKernelGGL(k_h264_chroma_mc_hbd<uint16_t>, dim3(n), dim3(128), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h);
    else
        hipLaunchKernelGGL(k_h264_chroma_mc_hbd<uint8_t>, dim3(n), dim3(128), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_weight_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks, int n,
                                hipStream_t stream)
{
    if (n <= 0)
        return 0;
    HBD_CHECK(bd);
    if (bd > 8)
        hipLaunchKernelGGL(k_h264_weight_hbd<uint16_t>, dim3(n), dim3(256), 0, stream, dst, src, stride, blocks, n, bd);
    else
        hipLaunchKernelGGL(k_h264_weight_hbd<uint8_t>, dim3(n), dim3(256), 0, stream, dst, src, stride, blocks, n, bd);
    LAUNCH_CHECK();
    return 0;
}
