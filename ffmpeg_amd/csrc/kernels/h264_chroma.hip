/*
 * h264_chroma.hip — H.264 8-bit chroma 1/8-pel bilinear MC and explicit weighted prediction, batched.
 *
 * Bit-exact restatement of
 *   put/avg_h264_chroma_mc{8,4,2}_8_c          libavcodec/h264chroma_template.c:28-172
 *       A=(8-x)(8-y) B=x(8-y) C=(8-x)y D=xy;  v = (A*s[0] + B*s[1] + C*s[stride] + D*s[stride+1] + 32) >> 6,
 *       avg: (dst + v + 1) >> 1; only samples with a non-zero weight are read (three cases, as the reference)
 *   weight_h264_pixels{16,8,4,2}_8_c, biweight_  libavcodec/h264dsp_template.c:30-100
 *
 * GPU design: 16 lanes per block, one lane per row (h <= 16) — these are the reference's per-call operands, one record per call;
 * throughput comes from the batch size.  Chroma MC moves its rows as dwords (below); weight / biweight still byte by byte.
 * Algorithmic traffic 2 B per sample (chroma MC), 2-3 B per sample (weight / biweight).
 */
#include "common.h"
#include "h264_kernels.h"

/* bytes p[0 .. nb-1] (nb <= 9, any alignment) as three dwords in stream order: only the aligned dwords that hold one of those
 * bytes are read (a block at the picture's edge reads nothing the reference does not read, rounded to its dwords) */
__device__ __forceinline__ void cm_row(const uint8_t *p, int nb, uint32_t (&w)[3])
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3), last = (sh + (uint32_t)nb - 1) >> 2; /* index of the last dword needed: 0..2 */
    const uint32_t d0 = q[0], d1 = last >= 1 ? q[1] : 0, d2 = last >= 2 ? q[2] : 0;
    w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
    w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
    w[2] = d2 >> (8 * sh);
}

/* FFHIP_MC_EMU (include/ffhip.h; h264_mb.c:297-317 -> videodsp_template.c:24-100): the same stream from clamped coordinates of the
 * reference picture's plane whose (0, 0) is org */
__device__ __forceinline__ void cm_row_emu(const uint8_t *org, ptrdiff_t stride, int x, int y, int nb, int pw, int ph, uint32_t (&w)[3])
{
    const uint8_t *row = org + (ptrdiff_t)min(max(y, 0), ph - 1) * stride;
    w[0] = w[1] = w[2] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
        if (i < nb)
            w[i >> 2] |= (uint32_t)row[min(max(x + i, 0), pw - 1)] << (8 * (i & 3));
}

/* 16 lanes per block, one lane per row: a row's (and the next row's) w + 1 source bytes arrive as aligned dwords, a sample is one
 * v_perm (s[k], s[k+1], t[k], t[k+1]) and one v_dot4_u32_u8 against (A, B, C, D), the row leaves as one or two dwords when dst
 * is aligned.  (Round 1 read and wrote every byte by itself: 26 memory instructions per lane for an 8-wide row, now 8.) */
/* workgroup `wg` of a list: 16 blocks, a thread per row */
__device__ __forceinline__ void chroma_mc_group(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks, int n, int wg,
                                                int pic_w, int pic_h)
{
    const int b = (wg * 256 + (int)threadIdx.x) >> 4;
    if (b >= n)
        return;
    const FFHipChromaBlock blk = blocks[b];
    /* 8 wide and at most 8 rows (the chroma of a 16 x 16 macroblock at 4:2:0: nearly every block of a picture): the block's sixteen lanes
     * take half a row each instead of leaving eight idle (round 6: 0.127 -> see docs/KERNELS.md R6.7) */
    const bool halves = blk.w_idx == 0 && blk.h <= 8;
    const int row = halves ? threadIdx.x & 7 : threadIdx.x & 15, xo = halves ? 4 * ((threadIdx.x >> 3) & 1) : 0;
    if (row >= blk.h)
        return;
    const int w = halves ? 4 : 8 >> blk.w_idx, x = blk.x & 7, y = blk.y & 7;
    const uint32_t A = (8 - x) * (8 - y), B = x * (8 - y), C = (8 - x) * y, D = x * y;
    const uint8_t *s = src + blk.src_offset + (ptrdiff_t)row * stride + xo;
    uint8_t *d = dst + blk.dst_offset + (ptrdiff_t)row * stride + xo;
    /* the reference's three cases read only samples with a non-zero weight: the right neighbour if x, the row below if y */
    uint32_t sw[3], tw[3] = { 0, 0, 0 };
    if (pic_w > 0 && (blk.flags & FFHIP_MC_EMU)) {
        const uint8_t *org = src + blk.src_offset;
        cm_row_emu(org, stride, blk.src_x + xo, blk.src_y + row, w + (x ? 1 : 0), pic_w, pic_h, sw);
        if (y)
            cm_row_emu(org, stride, blk.src_x + xo, blk.src_y + row + 1, w + (x ? 1 : 0), pic_w, pic_h, tw);
    } else {
        cm_row(s, w + (x ? 1 : 0), sw);
        if (y)
            cm_row(s + stride, w + (x ? 1 : 0), tw);
    }
    const uint32_t coef = A | B << 8 | C << 16 | D << 24;
    uint32_t out[2] = { 0, 0 };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < w) {
            const uint32_t s4 = __builtin_amdgcn_alignbyte(sw[(k >> 2) + 1], sw[k >> 2], k & 3);
            const uint32_t t4 = __builtin_amdgcn_alignbyte(tw[(k >> 2) + 1], tw[k >> 2], k & 3);
            const uint32_t q = __builtin_amdgcn_perm(t4, s4, 0x05040100u); /* s[k], s[k+1], t[k], t[k+1] */
            const uint32_t v = (__builtin_amdgcn_udot4(q, coef, 32u, false)) >> 6; /* <= 255: the weights sum to 64 */
            out[k >> 2] |= v << (8 * (k & 3));
        }
    }
    if (!(reinterpret_cast<uintptr_t>(d) & 3) && w >= 4) {
        uint32_t *dw = reinterpret_cast<uint32_t *>(d);
#pragma unroll
        for (int q = 0; q < 2; q++)
            if (4 * q < w) {
                uint32_t o = out[q];
                if (blk.avg) {
                    const uint32_t p = dw[q];
                    o = (p | o) - (((p ^ o) & 0xFEFEFEFEu) >> 1); /* four (a + b + 1) >> 1 at once (libavcodec/rnd_avg.h) */
                }
                dw[q] = o;
            }
    } else {
        for (int k = 0; k < w; k++) {
            const uint32_t v = (out[k >> 2] >> (8 * (k & 3))) & 0xFF;
            d[k] = (uint8_t)(blk.avg ? (d[k] + v + 1) >> 1 : v);
        }
    }
}

__global__ __launch_bounds__(256) void k_h264_chroma_mc(uint8_t *dst, const uint8_t *src, ptrdiff_t stride,
                                                        const FFHipChromaBlock *blocks, int n, int pic_w, int pic_h)
{
    chroma_mc_group(dst, src, stride, blocks, n, (int)blockIdx.x, pic_w, pic_h);
}

/* the lists of several planes in one launch (the picture layer: Cb and Cr of a stage) */
__global__ __launch_bounds__(256) void k_h264_chroma_mc_multi(FFHipPlaneMulti M)
{
    int si = 0;
    for (int i = 1; i < M.nseg; i++)
        if ((int)blockIdx.x >= M.seg[i].first)
            si = i;
    const FFHipPlaneSeg &S = M.seg[si];
    chroma_mc_group(S.dst, S.src, S.stride, static_cast<const FFHipChromaBlock *>(S.blocks), S.n, (int)blockIdx.x - S.first, M.pic_w, M.pic_h);
}

static int plane_multi_pack(FFHipPlaneMulti &M)
{
    int wg = 0, k = 0;
    for (int i = 0; i < M.nseg; i++) {
        if (M.seg[i].n <= 0)
            continue;
        M.seg[k] = M.seg[i];
        M.seg[k].first = wg;
        wg += cdiv(M.seg[k].n, 16);
        k++;
    }
    M.nseg = k;
    return wg;
}

int ffhip_launch_h264_chroma_mc_multi(FFHipPlaneMulti &M, hipStream_t stream)
{
    const int wg = plane_multi_pack(M);
    if (!wg)
        return 0;
    hipLaunchKernelGGL(k_h264_chroma_mc_multi, dim3(wg), dim3(256), 0, stream, M);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_chroma_mc(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks, int n,
                                hipStream_t stream, int pic_w, int pic_h)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_h264_chroma_mc, dim3(cdiv(n, 16)), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h);
    LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ void weight_group(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks, int n, int wg)
{
    const int b = (wg * 256 + (int)threadIdx.x) >> 4, row = threadIdx.x & 15;
    if (b >= n)
        return;
    const FFHipWeightBlock blk = blocks[b];
    if (row >= blk.height)
        return;
    const int w = 16 >> blk.w_idx, ld = blk.log2_denom;
    uint8_t *d = dst + blk.dst_offset + (ptrdiff_t)row * stride;
    if (!blk.bi) {
        int off = (int)((unsigned)(int)blk.offset << ld);
        if (ld)
            off += 1 << (ld - 1);
        for (int k = 0; k < w; k++)
            d[k] = (uint8_t)min(max((d[k] * blk.weightd + off) >> ld, 0), 255);
    } else {
        const uint8_t *s = src + blk.src_offset + (ptrdiff_t)row * stride;
        const int off = (int)((unsigned)(((int)blk.offset + 1) | 1) << ld);
        for (int k = 0; k < w; k++)
            d[k] = (uint8_t)min(max((s[k] * blk.weights + d[k] * blk.weightd + off) >> (ld + 1), 0), 255);
    }
}

__global__ __launch_bounds__(256) void k_h264_weight(uint8_t *dst, const uint8_t *src, ptrdiff_t stride,
                                                     const FFHipWeightBlock *blocks, int n)
{
    weight_group(dst, src, stride, blocks, n, (int)blockIdx.x);
}

__global__ __launch_bounds__(256) void k_h264_weight_multi(FFHipPlaneMulti M)
{
    int si = 0;
    for (int i = 1; i < M.nseg; i++)
        if ((int)blockIdx.x >= M.seg[i].first)
            si = i;
    const FFHipPlaneSeg &S = M.seg[si];
    weight_group(S.dst, S.src, S.stride, static_cast<const FFHipWeightBlock *>(S.blocks), S.n, (int)blockIdx.x - S.first);
}

int ffhip_launch_h264_weight_multi(FFHipPlaneMulti &M, hipStream_t stream)
{
    const int wg = plane_multi_pack(M);
    if (!wg)
        return 0;
    hipLaunchKernelGGL(k_h264_weight_multi, dim3(wg), dim3(256), 0, stream, M);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_weight(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks, int n,
                             hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_h264_weight, dim3(cdiv(n, 16)), dim3(256), 0, stream, dst, src, stride, blocks, n);
    LAUNCH_CHECK();
    return 0;
}
