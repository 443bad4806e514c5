/*
 * vp9_intra.hip — VP9 intra prediction, 8 / 10 / 12 bits (PIX = uint8_t / uint16_t; offsets and strides in bytes), batched: VP9DSPContext.intra_pred[tx][mode]
 * (libavcodec/vp9dsp_template.c:33-1153; enum IntraPredMode, libavcodec/vp9.h:45-62).
 * A block's neighbours arrive as its "edge line" e[] = left[0..N-1] (bottom to top, as the reference's left[]), the corner, then
 * top[0..] — the samples met walking up the left column, round the corner and along the top; each mode is its per-sample rule
 * over that line.  One thread per 4 samples of a row, no block-level state: prediction is a gather.  Blocks of a launch are
 * independent (their edges are inputs); a decoder orders launches by its reconstruction wavefront.
 */
#include "common.h"
#include "h264_kernels.h"

static_assert(sizeof(FFHipVp9Intra) == 12, "FFHipVp9Intra is a 12-byte record");

__device__ __forceinline__ int vi_a2(int a, int b) { return (a + b + 1) >> 1; }
__device__ __forceinline__ int vi_a3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }

template <int LOG2, typename PIX>
__device__ __forceinline__ int vi_sample(int mode, const PIX *e, int x, int y, int dc, int maxv)
{
    constexpr int n = 1 << LOG2;
    const PIX *T = e + n + 1; /* T[-1] = the corner */
    switch (mode) {
    case 0: return T[x];                                                       /* VERT */
    case 1: return e[n - 1 - y];                                               /* HOR */
    case 3: {                                                                  /* DIAG_DOWN_LEFT */
        const int i = x + y;
        if (LOG2 == 2)
            return i < 6 ? vi_a3(T[i], T[i + 1], T[i + 2]) : T[7];
        return i < n - 2 ? vi_a3(T[i], T[i + 1], T[i + 2]) : i == n - 2 ? (T[n - 2] + 3 * T[n - 1] + 2) >> 2 : T[n - 1];
    }
    case 4: {                                                                  /* DIAG_DOWN_RIGHT */
        const int i = n - 1 - y + x;
        return vi_a3(e[i], e[i + 1], e[i + 2]);
    }
    case 5: {                                                                  /* VERT_RIGHT */
        const int i = n / 2 - 1 - (y >> 1) + x;
        if (i >= n / 2 - 1) {
            const int k = n + i - (n / 2 - 1);
            return (y & 1) ? vi_a3(e[k - 1], e[k], e[k + 1]) : vi_a2(e[k], e[k + 1]);
        }
        return (y & 1) ? vi_a3(e[2 * i + 3], e[2 * i + 2], e[2 * i + 1]) : vi_a3(e[2 * i + 4], e[2 * i + 3], e[2 * i + 2]);
    }
    case 6: {                                                                  /* HOR_DOWN */
        const int i = 2 * n - 2 - 2 * y + x;
        if (i >= 2 * n)
            return vi_a3(e[i - n], e[i - n + 1], e[i - n + 2]);
        return (i & 1) ? vi_a3(e[(i >> 1) + 2], e[(i >> 1) + 1], e[i >> 1]) : vi_a2(e[(i >> 1) + 1], e[i >> 1]);
    }
    case 7: {                                                                  /* VERT_LEFT */
        const int i = (y >> 1) + x;
        if (LOG2 == 2)
            return (y & 1) ? vi_a3(T[i], T[i + 1], T[i + 2]) : vi_a2(T[i], T[i + 1]);
        if (i >= n - 1)
            return T[n - 1];
        if (y & 1)
            return i < n - 2 ? vi_a3(T[i], T[i + 1], T[i + 2]) : (T[n - 2] + 3 * T[n - 1] + 2) >> 2;
        return vi_a2(T[i], T[i + 1]);
    }
    case 8: {                                                                  /* HOR_UP */
        const int i = 2 * y + x;
        if (i >= 2 * n - 2)
            return e[n - 1];
        if (i == 2 * n - 3)
            return (e[n - 2] + 3 * e[n - 1] + 2) >> 2;
        return (i & 1) ? vi_a3(e[i >> 1], e[(i >> 1) + 1], e[(i >> 1) + 2]) : vi_a2(e[i >> 1], e[(i >> 1) + 1]);
    }
    case 9: return min(max(T[x] + e[n - 1 - y] - T[-1], 0), maxv);              /* TM */
    default: return dc;                                                        /* the DC family */
    }
}

template <int LOG2, typename PIX>
__global__ __launch_bounds__(256) void k_vp9_intra(uint8_t *dst, ptrdiff_t stride, const uint8_t *edges, const FFHipVp9Intra *blocks, int n, int bd)
{
    constexpr int N = 1 << LOG2, ITEMS = N * N / 4, QW = N / 4;
    const int gid = blockIdx.x * 256 + threadIdx.x, b = gid / ITEMS, it = gid % ITEMS;
    if (b >= n)
        return;
    const FFHipVp9Intra k = blocks[b];
    const PIX *e = reinterpret_cast<const PIX *>(edges + k.edge_offset);
    const int maxv = (1 << bd) - 1;
    const int mode = k.mode, y = it / QW, x0 = 4 * (it % QW);
    int dc = 0;
    if (mode == 2 || mode == 10 || mode == 11) {
        int sl = 0, st = 0;
        for (int i = 0; i < N; i++) {
            sl += e[i];
            st += e[N + 1 + i];
        }
        dc = mode == 2 ? (sl + st + N) >> (LOG2 + 1) : ((mode == 10 ? sl : st) + N / 2) >> LOG2;
    } else if (mode >= 12) {
        dc = (128 << (bd - 8)) + (mode == 12 ? 0 : mode == 13 ? -1 : 1);
    }
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        v[j] = vi_sample<LOG2, PIX>(mode, e, x0 + j, y, dc, maxv);
    if (sizeof(PIX) == 2) {
        uint16_t *d16 = reinterpret_cast<uint16_t *>(dst + k.dst_offset + (ptrdiff_t)y * stride) + x0;
        if (!(reinterpret_cast<uintptr_t>(d16) & 7)) {
            *reinterpret_cast<uint2 *>(d16) = make_uint2((uint32_t)v[0] | (uint32_t)v[1] << 16, (uint32_t)v[2] | (uint32_t)v[3] << 16);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                d16[j] = (uint16_t)v[j];
        }
        return;
    }
    uint8_t *d = dst + k.dst_offset + (ptrdiff_t)y * stride + x0;
    if (!(reinterpret_cast<uintptr_t>(d) & 3)) {
        *reinterpret_cast<uint32_t *>(d) = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            d[j] = (uint8_t)v[j];
    }
}

int ffhip_launch_vp9_intra(int tx, uint8_t *dst, ptrdiff_t stride, const uint8_t *edges, const FFHipVp9Intra *blocks, int n, hipStream_t stream)
{
    return ffhip_launch_vp9_intra_bd(8, tx, dst, stride, edges, blocks, n, stream);
}

int ffhip_launch_vp9_intra_bd(int bd, int tx, uint8_t *dst, ptrdiff_t stride, const uint8_t *edges, const FFHipVp9Intra *blocks, int n,
                              hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (tx < 0 || tx > 3) {
        ffhip_set_error("ffhip_vp9_intra: tx %d outside 0..3", tx);
        return FFHIP_EINVAL;
    }
    if (bd != 8 && !((bd == 10 || bd == 12) && !(((uintptr_t)dst | (uintptr_t)edges | (size_t)stride) & 1))) {
        ffhip_set_error("ffhip_vp9_intra: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    const int items = (16 << (2 * tx)) / 4;
    const dim3 grid(cdiv((int)(((long long)n * items + 255) / 256), 1)), block(256);
#define VI_CASE(T, LG) case T: if (bd == 8) hipLaunchKernelGGL((k_vp9_intra<LG, uint8_t>), grid, block, 0, stream, dst, stride, edges, blocks, n, 8); \
                               else hipLaunchKernelGGL((k_vp9_intra<LG, uint16_t>), grid, block, 0, stream, dst, stride, edges, blocks, n, bd); break;
    switch (tx) {
    VI_CASE(0, 2) VI_CASE(1, 3) VI_CASE(2, 4)
    default:
    VI_CASE(3, 5)
    }
#undef VI_CASE
    LAUNCH_CHECK();
    return 0;
}
