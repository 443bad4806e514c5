/*
 * h264_chroma.hip — H.264 8-bit chroma 1/8-pel bilinear MC and explicit weighted prediction, batched.
 *
 * Bit-exact restatement of
 *   put/avg_h264_chroma_mc{8,4,2}_8_c          libavcodec/h264chroma_template.c:28-172
 *       A=(8-x)(8-y) B=x(8-y) C=(8-x)y D=xy;  v = (A*s[0] + B*s[1] + C*s[stride] + D*s[stride+1] + 32) >> 6,
 *       avg: (dst + v + 1) >> 1; only samples with a non-zero weight are read (three cases, as the reference)
 *   weight_h264_pixels{16,8,4,2}_8_c, biweight_  libavcodec/h264dsp_template.c:30-100
 *
 * GPU design: 16 lanes per block, one lane per row (h <= 16), rows of <= 16 bytes handled with byte accesses —
 * these are the reference's per-call operands, one record per call; throughput comes from the batch size.
 * Algorithmic traffic 2 B per sample (chroma MC), 2-3 B per sample (weight / biweight).
 */
#include "common.h"
#include "h264_kernels.h"

__global__ __launch_bounds__(256) void k_h264_chroma_mc(uint8_t *dst, const uint8_t *src, ptrdiff_t stride,
                                                        const FFHipChromaBlock *blocks, int n)
{
    const int b = (blockIdx.x * 256 + threadIdx.x) >> 4, row = threadIdx.x & 15;
    if (b >= n)
        return;
    const FFHipChromaBlock blk = blocks[b];
    if (row >= blk.h)
        return;
    const int w = 8 >> blk.w_idx, x = blk.x & 7, y = blk.y & 7;
    const int A = (8 - x) * (8 - y), B = x * (8 - y), C = (8 - x) * y, D = x * y;
    const uint8_t *s = src + blk.src_offset + (ptrdiff_t)row * stride;
    uint8_t *d = dst + blk.dst_offset + (ptrdiff_t)row * stride;
    const ptrdiff_t step = C ? stride : 1;
    for (int k = 0; k < w; k++) {
        int v;
        if (D)
            v = A * s[k] + B * s[k + 1] + C * s[stride + k] + D * s[stride + k + 1];
        else if (B + C)
            v = A * s[k] + (B + C) * s[step + k];
        else
            v = A * s[k];
        v = (v + 32) >> 6;
        d[k] = (uint8_t)(blk.avg ? (d[k] + v + 1) >> 1 : v);
    }
}

int ffhip_launch_h264_chroma_mc(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks, int n,
                                hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_h264_chroma_mc, dim3(cdiv(n, 16)), dim3(256), 0, stream, dst, src, stride, blocks, n);
    LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void k_h264_weight(uint8_t *dst, const uint8_t *src, ptrdiff_t stride,
                                                     const FFHipWeightBlock *blocks, int n)
{
    const int b = (blockIdx.x * 256 + threadIdx.x) >> 4, row = threadIdx.x & 15;
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

int ffhip_launch_h264_weight(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks, int n,
                             hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_h264_weight, dim3(cdiv(n, 16)), dim3(256), 0, stream, dst, src, stride, blocks, n);
    LAUNCH_CHECK();
    return 0;
}
