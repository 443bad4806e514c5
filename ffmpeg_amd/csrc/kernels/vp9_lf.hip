/*
 * vp9_lf.hip — VP9 loop filter, 8 / 10 / 12 bits, batched: loop_filter() of libavcodec/vp9dsp_template.c:1780-1889,
 * the body of loop_filter_8[wd][dir], loop_filter_16[dir] and loop_filter_mix2[wd1][wd2][dir] (:1891-1966).
 * A record is one 8-sample segment of an edge; 8 lanes per segment, one per sample line; every read of a line happens before
 * any write of it.  The flat filters are evaluated as windows: radius 3 over p3..q3 (radius 7 over p7..q7) around the sample with
 * the ends repeated and the centre counted twice, as running sums.  Segments of one launch must not share samples (all edges of
 * one direction that are at least 16 apart, or checkasm-style tiles): VP9 orders overlapping edges inside a superblock.
 */
#include "common.h"
#include "h264_kernels.h"

static_assert(sizeof(FFHipVp9Edge) == 12, "FFHipVp9Edge is a 12-byte record");

__device__ __forceinline__ int vl_abs(int v) { return v < 0 ? -v : v; }

/* PIX = uint8_t (bd 8) / uint16_t; E, I, H arrive in 8-bit units and are scaled by << (bd - 8), the flatness threshold is
 * 1 << (bd - 8), the filter value clips to bd - 1 signed bits (vp9dsp_template.c:1784-1788,1866-1878); stride and offsets in bytes */
template <typename PIX>
__global__ __launch_bounds__(256) void k_vp9_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, int bd)
{
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 3, line = threadIdx.x & 7;
    if (e >= n)
        return;
    const FFHipVp9Edge ed = edges[e];
    const int wd = ed.wd_idx == 0 ? 4 : ed.wd_idx == 1 ? 8 : 16;
    const ptrdiff_t st = stride / (ptrdiff_t)sizeof(PIX), along = ed.dir ? 1 : st, across = ed.dir ? st : 1;
    PIX *pix = reinterpret_cast<PIX *>(base + ed.offset) + line * along;
    const int sh = bd - 8, E = ed.E << sh, I = ed.I << sh, H = ed.H << sh, F = 1 << sh, fmax = (1 << (bd - 1)) - 1, maxv = (1 << bd) - 1;
    auto clipf = [&](int v) { return min(max(v, -fmax - 1), fmax); };
    auto clipp = [&](int v) { return min(max(v, 0), maxv); };
    int px[16]; /* p7 .. p0, q0 .. q7 */
#pragma unroll
    for (int k = 0; k < 16; k++)
        px[k] = (wd >= 16 || (k >= 4 && k < 12)) ? pix[(k - 8) * across] : 0;
    const int p3 = px[4], p2 = px[5], p1 = px[6], p0 = px[7], q0 = px[8], q1 = px[9], q2 = px[10], q3 = px[11];
    if (!(vl_abs(p3 - p2) <= I && vl_abs(p2 - p1) <= I && vl_abs(p1 - p0) <= I && vl_abs(q1 - q0) <= I && vl_abs(q2 - q1) <= I &&
          vl_abs(q3 - q2) <= I && vl_abs(p0 - q0) * 2 + (vl_abs(p1 - q1) >> 1) <= E))
        return;
    bool flat_in = wd >= 8, flat_out = wd >= 16;
#pragma unroll
    for (int k = 1; k <= 3; k++)
        flat_in = flat_in && vl_abs(px[7 - k] - p0) <= F && vl_abs(px[8 + k] - q0) <= F;
#pragma unroll
    for (int k = 4; k <= 7; k++)
        flat_out = flat_out && vl_abs(px[7 - k] - p0) <= F && vl_abs(px[8 + k] - q0) <= F;
    if (flat_out && flat_in) {
        /* window sums of radius 7 with clamped ends: s(c+1) = s(c) + px[min(c+8, 15)] - px[max(c-7, 0)] */
        int s = 8 * px[0];
#pragma unroll
        for (int t = 1; t <= 8; t++)
            s += px[t]; /* window of c = 1: indices -6..8 -> seven times px[0] (one of them is index 0 itself) + px[1..8] */
        s -= px[0];
#pragma unroll
        for (int c = 1; c <= 14; c++) {
            pix[(c - 8) * across] = (PIX)((s + px[c] + 8) >> 4);
            s += px[c + 8 > 15 ? 15 : c + 8] - px[c - 7 < 0 ? 0 : c - 7];
        }
    } else if (flat_in) {
        /* radius 3 over px[4..11] */
        int s = 3 * px[4] + px[5] + px[6] + px[7] + px[8]; /* window of c = 5: indices 2..8 clamped to 4..11 */
#pragma unroll
        for (int c = 5; c <= 10; c++) {
            pix[(c - 8) * across] = (PIX)((s + px[c] + 4) >> 3);
            s += px[c + 4 > 11 ? 11 : c + 4] - px[c - 3 < 4 ? 4 : c - 3];
        }
    } else {
        const bool hev = vl_abs(p1 - p0) > H || vl_abs(q1 - q0) > H;
        int f = clipf(3 * (q0 - p0) + (hev ? clipf(p1 - q1) : 0));
        const int f1 = min(f + 4, fmax) >> 3, f2 = min(f + 3, fmax) >> 3;
        pix[-across] = (PIX)clipp(p0 + f2);
        pix[0] = (PIX)clipp(q0 - f1);
        if (!hev) {
            f = (f1 + 1) >> 1;
            pix[-2 * across] = (PIX)clipp(p1 + f);
            pix[across] = (PIX)clipp(q1 - f);
        }
    }
}

int ffhip_launch_vp9_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, hipStream_t stream)
{
    return ffhip_launch_vp9_loop_filter_bd(8, base, stride, edges, n, stream);
}

int ffhip_launch_vp9_loop_filter_bd(int bd, uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (bd == 8)
        hipLaunchKernelGGL(k_vp9_loop_filter<uint8_t>, dim3(cdiv(n, 32)), dim3(256), 0, stream, base, stride, edges, n, 8);
    else if ((bd == 10 || bd == 12) && !(((uintptr_t)base | (size_t)stride) & 1))
        hipLaunchKernelGGL(k_vp9_loop_filter<uint16_t>, dim3(cdiv(n, 32)), dim3(256), 0, stream, base, stride, edges, n, bd);
    else {
        ffhip_set_error("ffhip_vp9_loop_filter: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}
