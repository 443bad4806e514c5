/*
 * h264_qpel.hip — H.264 8-bit luma quarter-pel motion compensation, batched.
 *
 * Bit-exact restatement of put/avg_h264_qpel{16,8,4}_mcXY_8_c (libavcodec/h264qpel_template.c:313-459)
 * stated per output sample:
 *   tap6(a..f) = (c+d)*20 - (b+e)*5 + (a+f)                      (:77-305, the 6-tap lowpass)
 *   H = clip_u8((tap6 along x + 16) >> 5), V likewise along y,
 *   J = clip_u8((tap6 along y of the UNCLIPPED horizontal sums + 512) >> 10)     (the "hv" centre)
 *   quarter positions = rnd_avg (a+b+1)>>1 of two of {F, H, V, J} per the 16-entry table,
 *   avg_ variants rnd_avg the result with dst (op_avg, :461).
 *
 * GPU design: one wave per block; lane (y, xg) owns the 4 horizontally adjacent samples
 * x = 4*xg..4*xg+3 of row y (64 lanes = 16x16; 8x8 and 4x4 blocks use 16 / 4 lanes of their wave).
 * A lane pulls the (up to) 6 source rows x 12 bytes it needs as aligned dwords + v_alignbyte — the
 * rows of neighbouring lanes overlap, so the 21x21 reference footprint is fetched from HBM once and
 * re-served by L1 — keeps everything in registers, averages four samples at a time with the
 * packed rnd_avg32 identity (a|b) - (((a^b) & 0xfefefefe) >> 1) (libavcodec/rnd_avg.h), and writes
 * one dword.  mcXY is wave-uniform, so only the taps a position needs are computed.
 * Algorithmic traffic 2 B per sample (reference read once + destination write).
 */
#include "common.h"
#include "h264_kernels.h"

/* 12 source bytes starting at p (any alignment): aligned dword loads, funnel-shifted into place */
struct Row12 { uint32_t w[3]; };

__device__ __forceinline__ Row12 load_row12(const uint8_t *p)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    /* bytes sh..sh+9 are used (10 of the 12): the 4th dword is touched only when it holds one */
    const uint32_t d3 = sh == 3 ? q[3] : 0;
    Row12 r;
    r.w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
    r.w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
    r.w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
    return r;
}

__device__ __forceinline__ int rbyte(const Row12 &r, int i) { return (int)((r.w[i >> 2] >> (8 * (i & 3))) & 0xFF); }
__device__ __forceinline__ int tap6(int a, int b, int c, int d, int e, int f) { return (c + d) * 20 - (b + e) * 5 + (a + f); }
/* unclipped horizontal sum at sample i (0..4) of the lane; stream byte 0 is x-2 */
__device__ __forceinline__ int hraw(const Row12 &r, int i)
{
    return tap6(rbyte(r, i), rbyte(r, i + 1), rbyte(r, i + 2), rbyte(r, i + 3), rbyte(r, i + 4), rbyte(r, i + 5));
}
__device__ __forceinline__ uint32_t rnd_avg4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) & 0xFEFEFEFEu) >> 1); }

__global__ __launch_bounds__(256) void k_h264_qpel(uint8_t *dst, const uint8_t *src, ptrdiff_t stride,
                                                   const FFHipQpelBlock *blocks, int n)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= n)
        return;
    const FFHipQpelBlock blk = blocks[b];
    const int size = 16 >> __builtin_amdgcn_readfirstlane((int)blk.size_idx);
    const int mc = __builtin_amdgcn_readfirstlane((int)blk.mcxy) & 15;
    const bool avg = __builtin_amdgcn_readfirstlane((int)blk.avg) != 0;
    const int per_row = size >> 2;
    const int y = lane / per_row, xg = lane - y * per_row;
    if (y >= size)
        return;
    const int mx = mc & 3, my = mc >> 2;
    const uint8_t *s = src + __builtin_amdgcn_readfirstlane(blk.src_offset) + (ptrdiff_t)y * stride + 4 * xg - 2;
    uint8_t *d = dst + __builtin_amdgcn_readfirstlane(blk.dst_offset) + (ptrdiff_t)y * stride + 4 * xg;

    /* which of F/H/V/J the position combines (appendix A.6 table), all wave-uniform */
    const bool useJ = (mx == 2 && my != 0) || (my == 2 && mx != 0);
    const bool useV = (mx != 2 && my != 0) || (mc == 8);       /* V at x (mx 0,1) or x+1 (mx 3) */
    const bool useH = (my != 2 && mx != 0) || (mc == 2);       /* H at y (my 0,1) or y+1 (my 3) */
    const bool vcol1 = mx == 3, hrow1 = my == 3;

    Row12 r[6]; /* rows y-2 .. y+3 */
    if (useV || useJ) {
#pragma unroll
        for (int k = 0; k < 6; k++)
            r[k] = load_row12(s + (ptrdiff_t)(k - 2) * stride);
    } else {
        r[2] = load_row12(s);
        r[3] = hrow1 ? load_row12(s + stride) : r[2];
    }

    uint32_t out = 0;
    uint32_t pj = 0, ph = 0, pv = 0, pf = 0;
    if (useJ) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = tap6(hraw(r[0], i), hraw(r[1], i), hraw(r[2], i), hraw(r[3], i), hraw(r[4], i), hraw(r[5], i));
            pj |= (uint32_t)clip_u8((v + 512) >> 10) << (8 * i);
        }
    }
    if (useH) {
        const Row12 &hr = hrow1 ? r[3] : r[2];
#pragma unroll
        for (int i = 0; i < 4; i++)
            ph |= (uint32_t)clip_u8((hraw(hr, i) + 16) >> 5) << (8 * i);
    }
    if (useV) {
        const int c0 = vcol1 ? 3 : 2; /* stream byte of sample 0's column */
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = tap6(rbyte(r[0], c0 + i), rbyte(r[1], c0 + i), rbyte(r[2], c0 + i), rbyte(r[3], c0 + i),
                               rbyte(r[4], c0 + i), rbyte(r[5], c0 + i));
            pv |= (uint32_t)clip_u8((v + 16) >> 5) << (8 * i);
        }
    }
    /* full-pel samples: F, F(+1,0) for mc30, F(0,+1) for mc03 */
    {
        const Row12 &fr = (mc == 12) ? r[3] : r[2];
        const uint32_t sh = (mc == 3) ? 3 : 2;
        pf = __builtin_amdgcn_alignbyte(fr.w[1], fr.w[0], sh);
    }
    switch (mc) {
    case 0:  out = pf; break;
    case 1: case 3:  out = rnd_avg4(pf, ph); break;
    case 2:  out = ph; break;
    case 4: case 12: out = rnd_avg4(pf, pv); break;
    case 5: case 7: case 13: case 15: out = rnd_avg4(ph, pv); break;
    case 6: case 14: out = rnd_avg4(ph, pj); break;
    case 8:  out = pv; break;
    case 9: case 11: out = rnd_avg4(pv, pj); break;
    default: out = pj; break; /* 10 */
    }
    if (!((reinterpret_cast<uintptr_t>(d)) & 3)) {
        uint32_t *dw = reinterpret_cast<uint32_t *>(d);
        if (avg)
            out = rnd_avg4(*dw, out);
        *dw = out;
    } else {
        for (int i = 0; i < 4; i++) {
            const uint32_t v = (out >> (8 * i)) & 0xFF;
            d[i] = (uint8_t)(avg ? (d[i] + v + 1) >> 1 : v);
        }
    }
}

int ffhip_launch_h264_qpel(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                           hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_h264_qpel, dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, src, stride, blocks, n);
    LAUNCH_CHECK();
    return 0;
}
