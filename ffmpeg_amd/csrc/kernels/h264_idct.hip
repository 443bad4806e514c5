/*
 * h264_idct.hip — H.264 8-bit inverse transforms + add, batched.
 *
 * Bit-exact restatement of ff_h264_idct_add_8_c, ff_h264_idct8_add_8_c, ff_h264_idct_dc_add_8_c,
 * ff_h264_idct8_dc_add_8_c (libavcodec/h264idct_template.c:33-175) and the macroblock dispatchers
 * ff_h264_idct_add16 / _add16intra / idct8_add4 (:177-214):
 *   - coefficients are int16, stored transposed; pass 1 walks stride-N samples and writes its
 *     results back as int16 (16-bit wrap), pass 2 walks contiguous samples and adds
 *     (x >> 6) into column i of dst with clipping; block[0] += 32 first; coefficients zeroed after.
 *   - all sums are modulo 2^32, shifts arithmetic.
 *
 * GPU design (HBM-bound: 384 B per 8x8 block = 128 read + 128 clear + 64 + 64 dst):
 *   one thread owns one block end to end, so both passes run in registers with no cross-lane
 *   traffic.  For 8x8 the 128-byte coefficient records of a workgroup's 256 blocks are first
 *   copied HBM -> LDS with fully coalesced 16-B loads (4 KiB contiguous per wave instruction)
 *   into records padded to 144 B, which makes the per-thread ds_read_b128 of "my block"
 *   bank-conflict free; the clears go back as coalesced 16-B stores over the same range.
 *   dst rows move as one 8-B (4-B for 4x4) access per row; consecutive blocks of a raster-ordered
 *   batch make those accesses contiguous across lanes.
 */
#include "common.h"
#include "h264_kernels.h"

#define NT 256

__device__ __forceinline__ int16_t lo16(uint32_t v) { return (int16_t)(v & 0xFFFF); }
__device__ __forceinline__ int16_t hi16(uint32_t v) { return (int16_t)(v >> 16); }

/* one 8-point pass of the 8x8 transform; in/out as 32-bit wrap-around values */
__device__ __forceinline__ void idct8_1d(const int in[8], uint32_t out[8])
{
    const uint32_t a0 = (uint32_t)in[0] + (uint32_t)in[4];
    const uint32_t a2 = (uint32_t)in[0] - (uint32_t)in[4];
    const uint32_t a4 = (uint32_t)(in[2] >> 1) - (uint32_t)in[6];
    const uint32_t a6 = (uint32_t)(in[6] >> 1) + (uint32_t)in[2];
    const uint32_t b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    const int a1 = (int)(-(uint32_t)in[3] + (uint32_t)in[5] - (uint32_t)in[7] - (uint32_t)(in[7] >> 1));
    const int a3 = (int)((uint32_t)in[1] + (uint32_t)in[7] - (uint32_t)in[3] - (uint32_t)(in[3] >> 1));
    const int a5 = (int)(-(uint32_t)in[1] + (uint32_t)in[7] + (uint32_t)in[5] + (uint32_t)(in[5] >> 1));
    const int a7 = (int)((uint32_t)in[3] + (uint32_t)in[5] + (uint32_t)in[1] + (uint32_t)(in[1] >> 1));
    const uint32_t b1 = (uint32_t)(a7 >> 2) + (uint32_t)a1;
    const uint32_t b3 = (uint32_t)a3 + (uint32_t)(a5 >> 2);
    const uint32_t b5 = (uint32_t)(a3 >> 2) - (uint32_t)a5;
    const uint32_t b7 = (uint32_t)a7 - (uint32_t)(a1 >> 2);
    out[0] = b0 + b7; out[7] = b0 - b7;
    out[1] = b2 + b5; out[6] = b2 - b5;
    out[2] = b4 + b3; out[5] = b4 - b3;
    out[3] = b6 + b1; out[4] = b6 - b1;
}

/* full 8x8 on coefficients held as 8 rows of 4 packed dwords (row k = block[8k..8k+7]) */
__device__ __forceinline__ void idct8_add_regs(const uint32_t rows[8][4], uint8_t *dst, ptrdiff_t stride, bool vec)
{
    int16_t t[8][8]; /* t[k][i] = intermediate block[i + 8k] after pass 1 */
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int in[8];
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t v = rows[k][i >> 1];
            in[k] = (i & 1) ? hi16(v) : lo16(v);
        }
        if (i == 0)
            in[0] = (int16_t)(in[0] + 32); /* block[0] += 32 in int16 */
        idct8_1d(in, o);
#pragma unroll
        for (int k = 0; k < 8; k++)
            t[k][i] = (int16_t)o[k];
    }
    /* pass 2: row i of t (contiguous block[8i..8i+7]) -> column i of dst */
    uint8_t res[8][8]; /* res[k][i] = delta-applied pixel at row k, column i */
    uint32_t drow[8][2];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint8_t *p = dst + k * stride;
        if (vec) {
            const uint2 w = *reinterpret_cast<const uint2 *>(p);
            drow[k][0] = w.x;
            drow[k][1] = w.y;
        } else {
            drow[k][0] = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
            drow[k][1] = p[4] | (p[5] << 8) | (p[6] << 16) | ((uint32_t)p[7] << 24);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int in[8];
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            in[k] = t[i][k];
        idct8_1d(in, o);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int px = (drow[k][i >> 2] >> (8 * (i & 3))) & 0xFF;
            res[k][i] = (uint8_t)clip_u8(px + ((int)o[k] >> 6));
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint8_t *p = dst + k * stride;
        const uint32_t w0 = pack4(res[k][0], res[k][1], res[k][2], res[k][3]);
        const uint32_t w1 = pack4(res[k][4], res[k][5], res[k][6], res[k][7]);
        if (vec) {
            *reinterpret_cast<uint2 *>(p) = make_uint2(w0, w1);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++)
                p[i] = res[k][i];
        }
    }
}

#define REC8 144 /* padded LDS record of one 8x8 block: 128 B + 16 B => conflict-free ds_read_b128 */

/* one workgroup's NT blocks, b0 = the first one's index; lds: NT * REC8 bytes */
__device__ __forceinline__ void idct8_add_group(uint8_t *lds, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset, int16_t *blocks,
                                                int n, int dst_vec, long long b0)
{
    const int tid = threadIdx.x;
    const int nb = (int)min((long long)NT, (long long)n - b0);
    uint4 *gsrc = reinterpret_cast<uint4 *>(blocks + b0 * 64);
    /* coalesced copy in: 8 x 16-B pieces per block */
    for (int it = tid; it < nb * 8; it += NT) {
        const uint4 v = gsrc[it];
        *reinterpret_cast<uint4 *>(lds + (it >> 3) * REC8 + (it & 7) * 16) = v;
    }
    __syncthreads();
    /* coalesced clear: memset(block, 0, 64 * sizeof(dctcoef)) */
    for (int it = tid; it < nb * 8; it += NT)
        gsrc[it] = make_uint4(0, 0, 0, 0);
    if (tid >= nb)
        return;
    uint32_t r2[8][4];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(lds + tid * REC8 + k * 16);
        r2[k][0] = v.x; r2[k][1] = v.y; r2[k][2] = v.z; r2[k][3] = v.w;
    }
    uint8_t *dst = dst_base + dst_offset[b0 + tid];
    const bool vec = dst_vec && !((uintptr_t)dst & 7);
    idct8_add_regs(r2, dst, stride, vec);
}

__global__ __launch_bounds__(NT) void k_h264_idct8_add(uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset,
                                                       int16_t *blocks, int n, int dst_vec)
{
    __shared__ __align__(16) uint8_t lds[NT * REC8];
    idct8_add_group(lds, dst_base, stride, dst_offset, blocks, n, dst_vec, (long long)blockIdx.x * NT);
}

/* 4x4: direct 2 x 16-B loads per thread */
__device__ __forceinline__ void idct4_add_regs(const uint4 &c0, const uint4 &c1, uint8_t *dst, ptrdiff_t stride, bool vec)
{
    const uint32_t w[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
    int16_t b[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
        b[i] = (i & 1) ? hi16(w[i >> 1]) : lo16(w[i >> 1]);
    b[0] = (int16_t)(b[0] + 32);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t z0 = (uint32_t)b[i] + (uint32_t)b[i + 8];
        const uint32_t z1 = (uint32_t)b[i] - (uint32_t)b[i + 8];
        const uint32_t z2 = (uint32_t)(b[i + 4] >> 1) - (uint32_t)b[i + 12];
        const uint32_t z3 = (uint32_t)b[i + 4] + (uint32_t)(b[i + 12] >> 1);
        b[i]      = (int16_t)(z0 + z3);
        b[i + 4]  = (int16_t)(z1 + z2);
        b[i + 8]  = (int16_t)(z1 - z2);
        b[i + 12] = (int16_t)(z0 - z3);
    }
    uint32_t drow[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint8_t *p = dst + k * stride;
        drow[k] = vec ? *reinterpret_cast<const uint32_t *>(p)
                      : (uint32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24));
    }
    int res[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int16_t *r = b + 4 * i;
        const uint32_t z0 = (uint32_t)r[0] + (uint32_t)r[2];
        const uint32_t z1 = (uint32_t)r[0] - (uint32_t)r[2];
        const uint32_t z2 = (uint32_t)(r[1] >> 1) - (uint32_t)r[3];
        const uint32_t z3 = (uint32_t)r[1] + (uint32_t)(r[3] >> 1);
        const uint32_t o[4] = { z0 + z3, z1 + z2, z1 - z2, z0 - z3 };
#pragma unroll
        for (int k = 0; k < 4; k++)
            res[k][i] = clip_u8((int)((drow[k] >> (8 * i)) & 0xFF) + ((int)o[k] >> 6));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint8_t *p = dst + k * stride;
        if (vec) {
            *reinterpret_cast<uint32_t *>(p) = pack4(res[k][0], res[k][1], res[k][2], res[k][3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                p[i] = (uint8_t)res[k][i];
        }
    }
}

template <int N>
__device__ __forceinline__ void dc_add_regs(int dc, uint8_t *dst, ptrdiff_t stride)
{
    for (int y = 0; y < N; y++)
#pragma unroll
        for (int x = 0; x < N; x++)
            dst[y * stride + x] = (uint8_t)clip_u8(dst[y * stride + x] + dc);
}

__global__ __launch_bounds__(NT) void k_h264_idct4_add(uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset,
                                                       int16_t *blocks, int n, int dst_vec)
{
    const long long i = (long long)blockIdx.x * NT + threadIdx.x;
    if (i >= n)
        return;
    uint4 *g = reinterpret_cast<uint4 *>(blocks + i * 16);
    const uint4 c0 = g[0], c1 = g[1];
    g[0] = make_uint4(0, 0, 0, 0);
    g[1] = make_uint4(0, 0, 0, 0);
    uint8_t *dst = dst_base + dst_offset[i];
    idct4_add_regs(c0, c1, dst, stride, dst_vec && !((uintptr_t)dst & 3));
}

template <int N>
__global__ __launch_bounds__(NT) void k_h264_idct_dc_add(uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset,
                                                         int16_t *blocks, int n)
{
    const long long i = (long long)blockIdx.x * NT + threadIdx.x;
    if (i >= n)
        return;
    int16_t *b = blocks + i * (N * N);
    const int dc = (b[0] + 32) >> 6;
    b[0] = 0;
    dc_add_regs<N>(dc, dst_base + dst_offset[i], stride);
}

/* add_pixels4_clear / add_pixels8_clear (the lossless transform bypass), h264addpx_template.c:28-74: dst += residual with 8-bit
 * wrap-around (no clip), then the residual is cleared.  One thread per block. */
template <int N>
__global__ __launch_bounds__(NT) void k_h264_add_pixels_clear(uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset,
                                                              int16_t *blocks, int n)
{
    const long long i = (long long)blockIdx.x * NT + threadIdx.x;
    if (i >= n)
        return;
    int16_t *b = blocks + i * (N * N);
    uint8_t *dst = dst_base + dst_offset[i];
    for (int y = 0; y < N; y++)
#pragma unroll
        for (int x = 0; x < N; x++) {
            dst[y * stride + x] = (uint8_t)(dst[y * stride + x] + (unsigned)b[y * N + x]);
            b[y * N + x] = 0;
        }
}

int ffhip_launch_h264_idct_add(int kind, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset,
                               int16_t *blocks, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    const dim3 grid(cdiv(n, NT)), block(NT);
    switch (kind) {
    case FFHIP_H264_IDCT8:
        hipLaunchKernelGGL(k_h264_idct8_add, grid, block, 0, stream, dst_base, stride, dst_offset, blocks, n,
                           !(((uintptr_t)dst_base | (size_t)stride) & 7));
        break;
    case FFHIP_H264_IDCT4:
        hipLaunchKernelGGL(k_h264_idct4_add, grid, block, 0, stream, dst_base, stride, dst_offset, blocks, n,
                           !(((uintptr_t)dst_base | (size_t)stride) & 3));
        break;
    case FFHIP_H264_IDCT4_DC:
        hipLaunchKernelGGL((k_h264_idct_dc_add<4>), grid, block, 0, stream, dst_base, stride, dst_offset, blocks, n);
        break;
    case FFHIP_H264_IDCT8_DC:
        hipLaunchKernelGGL((k_h264_idct_dc_add<8>), grid, block, 0, stream, dst_base, stride, dst_offset, blocks, n);
        break;
    case FFHIP_H264_ADD_PIXELS4_CLEAR:
        hipLaunchKernelGGL((k_h264_add_pixels_clear<4>), grid, block, 0, stream, dst_base, stride, dst_offset, blocks, n);
        break;
    case FFHIP_H264_ADD_PIXELS8_CLEAR:
        hipLaunchKernelGGL((k_h264_add_pixels_clear<8>), grid, block, 0, stream, dst_base, stride, dst_offset, blocks, n);
        break;
    default:
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}

/*
 * Several block lists in ONE launch (the picture layer's residual stage: up to three planes x four kinds; a launch costs the
 * host more than these small kernels cost the GPU): workgroup b serves the segment whose range of workgroups holds b.  The
 * lists never name the same destination block twice (each block has one kind), so the segments are independent.
 */
__global__ __launch_bounds__(NT) void k_h264_idct_multi(FFHipIdctMulti M)
{
    __shared__ __align__(16) uint8_t lds[NT * REC8];
    int si = 0;
    for (int i = 1; i < M.nseg; i++)
        if ((int)blockIdx.x >= M.seg[i].first)
            si = i;
    const FFHipIdctSeg &S = M.seg[si];
    const long long b0 = (long long)((int)blockIdx.x - S.first) * NT, i = b0 + threadIdx.x;
    const ptrdiff_t stride = S.stride;
    if (S.kind == FFHIP_H264_IDCT8) {
        idct8_add_group(lds, S.dst, stride, S.offs, S.coef, S.n, !(((uintptr_t)S.dst | (size_t)stride) & 7), b0);
        return;
    }
    if (i >= S.n)
        return;
    uint8_t *dst = S.dst + S.offs[i];
    if (S.kind == FFHIP_H264_IDCT4) {
        uint4 *g = reinterpret_cast<uint4 *>(S.coef + i * 16);
        const uint4 c0 = g[0], c1 = g[1];
        g[0] = make_uint4(0, 0, 0, 0);
        g[1] = make_uint4(0, 0, 0, 0);
        idct4_add_regs(c0, c1, dst, stride, !(((uintptr_t)S.dst | (size_t)stride | (uintptr_t)dst) & 3));
    } else {
        const int N = S.kind == FFHIP_H264_IDCT8_DC ? 8 : 4;
        int16_t *b = S.coef + i * (N * N);
        const int dc = (b[0] + 32) >> 6;
        b[0] = 0;
        if (N == 8)
            dc_add_regs<8>(dc, dst, stride);
        else
            dc_add_regs<4>(dc, dst, stride);
    }
}

int ffhip_launch_h264_idct_multi(FFHipIdctMulti &M, hipStream_t stream)
{
    int wg = 0, k = 0;
    for (int i = 0; i < M.nseg; i++) {
        if (M.seg[i].n <= 0)
            continue;
        if (M.seg[i].kind < FFHIP_H264_IDCT4 || M.seg[i].kind > FFHIP_H264_IDCT8_DC)
            return FFHIP_EINVAL;
        M.seg[k] = M.seg[i];
        M.seg[k].first = wg;
        wg += cdiv(M.seg[k].n, NT);
        k++;
    }
    M.nseg = k;
    if (!k)
        return 0;
    hipLaunchKernelGGL(k_h264_idct_multi, dim3(wg), dim3(NT), 0, stream, M);
    LAUNCH_CHECK();
    return 0;
}

/* ---- macroblock dispatchers --------------------------------------------------------------- */
/* scan8[] luma part, libavcodec/h264_parse.h:40-57 */
__constant__ uint8_t c_scan8[16] = {
    4 + 1 * 8, 5 + 1 * 8, 4 + 2 * 8, 5 + 2 * 8, 6 + 1 * 8, 7 + 1 * 8, 6 + 2 * 8, 7 + 2 * 8,
    4 + 3 * 8, 5 + 3 * 8, 4 + 4 * 8, 5 + 4 * 8, 6 + 3 * 8, 7 + 3 * 8, 6 + 4 * 8, 7 + 4 * 8,
};

/* which: 0 idct_add16, 2 idct_add16intra — one thread per 4x4 block */
__global__ __launch_bounds__(NT) void k_h264_idct_add16(int which, uint8_t *dst_base, ptrdiff_t stride,
                                                        const int32_t *mb_offset, const int32_t *blockoffset,
                                                        int16_t *blocks, const uint8_t *nnzc, int nmb, int dst_vec)
{
    const long long id = (long long)blockIdx.x * NT + threadIdx.x;
    if (id >= (long long)nmb * 16)
        return;
    const int mb = (int)(id >> 4), i = (int)(id & 15);
    const int nnz = nnzc[(size_t)mb * 40 + c_scan8[i]];
    int16_t *b = blocks + ((size_t)mb * 16 + i) * 16;
    uint8_t *dst = dst_base + mb_offset[mb] + blockoffset[i];
    bool full, dc;
    if (which == 0) {
        full = nnz && !(nnz == 1 && b[0]);
        dc = nnz == 1 && b[0];
    } else {
        full = nnz != 0;
        dc = !nnz && b[0];
    }
    if (full) {
        uint4 *g = reinterpret_cast<uint4 *>(b);
        const uint4 c0 = g[0], c1 = g[1];
        g[0] = make_uint4(0, 0, 0, 0);
        g[1] = make_uint4(0, 0, 0, 0);
        idct4_add_regs(c0, c1, dst, stride, dst_vec && !((uintptr_t)dst & 3));
    } else if (dc) {
        const int v = (b[0] + 32) >> 6;
        b[0] = 0;
        dc_add_regs<4>(v, dst, stride);
    }
}

/* idct8_add4 — one thread per 8x8 block, direct loads (skipped blocks cost no traffic) */
__global__ __launch_bounds__(NT) void k_h264_idct8_add4(uint8_t *dst_base, ptrdiff_t stride, const int32_t *mb_offset,
                                                        const int32_t *blockoffset, int16_t *blocks,
                                                        const uint8_t *nnzc, int nmb, int dst_vec)
{
    const long long id = (long long)blockIdx.x * NT + threadIdx.x;
    if (id >= (long long)nmb * 4)
        return;
    const int mb = (int)(id >> 2), i = (int)(id & 3) * 4;
    const int nnz = nnzc[(size_t)mb * 40 + c_scan8[i]];
    if (!nnz)
        return;
    int16_t *b = blocks + ((size_t)mb * 16 + i) * 16;
    uint8_t *dst = dst_base + mb_offset[mb] + blockoffset[i];
    if (nnz == 1 && b[0]) {
        const int v = (b[0] + 32) >> 6;
        b[0] = 0;
        dc_add_regs<8>(v, dst, stride);
        return;
    }
    uint4 *g = reinterpret_cast<uint4 *>(b);
    uint32_t r2[8][4];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint4 v = g[k];
        r2[k][0] = v.x; r2[k][1] = v.y; r2[k][2] = v.z; r2[k][3] = v.w;
        g[k] = make_uint4(0, 0, 0, 0);
    }
    idct8_add_regs(r2, dst, stride, dst_vec && !((uintptr_t)dst & 7));
}

int ffhip_launch_h264_idct_add_mb(int which, uint8_t *dst_base, ptrdiff_t stride, const int32_t *mb_offset,
                                  const int32_t *blockoffset16, int16_t *blocks, const uint8_t *nnzc, int nmb,
                                  hipStream_t stream)
{
    if (nmb <= 0)
        return 0;
    if (which == 1) {
        hipLaunchKernelGGL(k_h264_idct8_add4, dim3(cdiv(nmb * 4, NT)), dim3(NT), 0, stream, dst_base, stride, mb_offset,
                           blockoffset16, blocks, nnzc, nmb, !(((uintptr_t)dst_base | (size_t)stride) & 7));
    } else if (which == 0 || which == 2) {
        hipLaunchKernelGGL(k_h264_idct_add16, dim3(cdiv(nmb * 16, NT)), dim3(NT), 0, stream, which, dst_base, stride,
                           mb_offset, blockoffset16, blocks, nnzc, nmb, !(((uintptr_t)dst_base | (size_t)stride) & 3));
    } else {
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}

/* ---- idct_add8 (4:2:0): the four 4x4 blocks of Cb and of Cr of every macroblock (h264idct_template.c:216-228) ------------- */
/* one thread per chroma block: q = 0..7 -> plane j = 1 + q / 4, block i = 16 j + q % 4; nnzc is the decoder's 15 x 8 cache */
__global__ __launch_bounds__(NT) void k_h264_idct_add8(uint8_t *cb_base, uint8_t *cr_base, ptrdiff_t stride, const int32_t *mb_offset,
                                                       const int32_t *blockoffset48, int16_t *blocks, const uint8_t *nnzc, int nmb, int dst_vec)
{
    const long long id = (long long)blockIdx.x * NT + threadIdx.x;
    if (id >= (long long)nmb * 8)
        return;
    const int mb = (int)(id >> 3), q = (int)(id & 7), j = 1 + (q >> 2), i = 16 * j + (q & 3), k = q & 3;
    const int sc = 4 + (k & 1) + (5 * j + 1 + (k >> 1)) * 8;    /* scan8[i], libavcodec/h264_parse.h:45-50 */
    int16_t *b = blocks + ((size_t)mb * 48 + i) * 16;
    uint8_t *dst = (j == 1 ? cb_base : cr_base) + mb_offset[mb] + blockoffset48[i];
    if (nnzc[(size_t)mb * 120 + sc]) {
        uint4 *g = reinterpret_cast<uint4 *>(b);
        const uint4 c0 = g[0], c1 = g[1];
        g[0] = make_uint4(0, 0, 0, 0);
        g[1] = make_uint4(0, 0, 0, 0);
        idct4_add_regs(c0, c1, dst, stride, dst_vec && !((uintptr_t)dst & 3));
    } else if (b[0]) {
        const int v = (b[0] + 32) >> 6;
        b[0] = 0;
        dc_add_regs<4>(v, dst, stride);
    }
}

int ffhip_launch_h264_idct_add8(uint8_t *cb_base, uint8_t *cr_base, ptrdiff_t stride, const int32_t *mb_offset, const int32_t *blockoffset48,
                                int16_t *blocks, const uint8_t *nnzc, int nmb, hipStream_t stream)
{
    if (nmb <= 0)
        return 0;
    hipLaunchKernelGGL(k_h264_idct_add8, dim3(cdiv(nmb * 8, NT)), dim3(NT), 0, stream, cb_base, cr_base, stride, mb_offset, blockoffset48,
                       blocks, nnzc, nmb, !(((uintptr_t)cb_base | (uintptr_t)cr_base | (size_t)stride) & 3));
    LAUNCH_CHECK();
    return 0;
}

/* ---- DC transforms with dequantisation (h264idct_template.c:259-293, 323-345) ----------------------------------------------- */
/* luma: one thread per macroblock: 4x4 Hadamard of input[16], results scattered to the DC positions of output[256] */
__global__ __launch_bounds__(NT) void k_h264_luma_dc_dequant(int16_t *output, size_t out_pitch, const int16_t *input, size_t in_pitch,
                                                             const int32_t *qmul, int n)
{
    const long long m = (long long)blockIdx.x * NT + threadIdx.x;
    if (m >= n)
        return;
    const int16_t *in = input + m * in_pitch;
    int16_t *out = output + m * out_pitch;
    const uint32_t q = (uint32_t)qmul[m];
    int t[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int z0 = in[4 * i] + in[4 * i + 1], z1 = in[4 * i] - in[4 * i + 1], z2 = in[4 * i + 2] - in[4 * i + 3], z3 = in[4 * i + 2] + in[4 * i + 3];
        t[4 * i] = z0 + z3; t[4 * i + 1] = z0 - z3; t[4 * i + 2] = z1 - z2; t[4 * i + 3] = z1 + z2;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int o = (i & 1) * 32 + (i >> 1) * 128;  /* x_offset[] = { 0, 2*16, 8*16, 10*16 } */
        const uint32_t z0 = (uint32_t)t[i] + (uint32_t)t[8 + i], z1 = (uint32_t)t[i] - (uint32_t)t[8 + i];
        const uint32_t z2 = (uint32_t)t[4 + i] - (uint32_t)t[12 + i], z3 = (uint32_t)t[4 + i] + (uint32_t)t[12 + i];
        out[o]      = (int16_t)((int)((z0 + z3) * q + 128) >> 8);
        out[16 + o] = (int16_t)((int)((z1 + z2) * q + 128) >> 8);
        out[64 + o] = (int16_t)((int)((z1 - z2) * q + 128) >> 8);
        out[80 + o] = (int16_t)((int)((z0 - z3) * q + 128) >> 8);
    }
}

/* chroma (4:2:0): one thread per plane of a macroblock, in place on block[0], [16], [32], [48] */
__global__ __launch_bounds__(NT) void k_h264_chroma_dc_dequant(int16_t *blocks, const int32_t *block_offset, const int32_t *qmul, int n)
{
    const long long m = (long long)blockIdx.x * NT + threadIdx.x;
    if (m >= n)
        return;
    int16_t *b = blocks + block_offset[m];
    const uint32_t q = (uint32_t)qmul[m];
    uint32_t a = (uint32_t)(int)b[0], bb = (uint32_t)(int)b[16], c = (uint32_t)(int)b[32], d = (uint32_t)(int)b[48];
    const uint32_t e = a - bb;
    a = a + bb; bb = c - d; c = c + d;
    b[0]  = (int16_t)((int)((a + c) * q) >> 7);
    b[16] = (int16_t)((int)((e + bb) * q) >> 7);
    b[32] = (int16_t)((int)((a - c) * q) >> 7);
    b[48] = (int16_t)((int)((e - bb) * q) >> 7);
}

int ffhip_launch_h264_luma_dc_dequant(int16_t *output, size_t out_pitch, const int16_t *input, size_t in_pitch, const int32_t *qmul, int n,
                                      hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_h264_luma_dc_dequant, dim3(cdiv(n, NT)), dim3(NT), 0, stream, output, out_pitch, input, in_pitch, qmul, n);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_chroma_dc_dequant(int16_t *blocks, const int32_t *block_offset, const int32_t *qmul, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_h264_chroma_dc_dequant, dim3(cdiv(n, NT)), dim3(NT), 0, stream, blocks, block_offset, qmul, n);
    LAUNCH_CHECK();
    return 0;
}
