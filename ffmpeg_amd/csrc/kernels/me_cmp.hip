/*
 * me_cmp.hip — block comparison metrics and the exhaustive motion search.
 *
 * Bit-exact restatement of
 *   pix_abs16_c / pix_abs8_c                      libavcodec/me_cmp.c:117-143,272-290   (SAD)
 *   hadamard8_diff8x8_c / hadamard8_diff16_c      libavcodec/me_cmp.c:514-562,933-950   (SATD: sum of |H8 (a-b) H8^T|)
 *   ff_me_search_esa driven as vf_mestimate does  libavfilter/motion_estimation.c:32-40,60-95,
 *                                                 libavfilter/vf_mestimate.c:101,119-129
 *
 * Search semantics: window [x_mb±R]∩[0,(b_w-1)*mb] x [y_mb±R]∩[0,(b_h-1)*mb]; the
 * zero-MV cost is evaluated first and kept unless a candidate is STRICTLY cheaper; candidates are
 * visited in raster order, so the first minimum wins.  Equivalent closed form used here: take the
 * minimum cost with the smallest raster index; it replaces the zero MV only if it is < cost(zero MV).
 *
 * GPU design: one wave per macroblock.  The reference window ((mb+2R)^2 bytes) and the current block
 * are staged once in LDS; lane l evaluates candidates l, l+64, ... (raster order per lane), SAD with
 * v_sad_u8 on dwords funnel-shifted (v_alignbyte) out of the LDS rows, SATD with the butterflies in
 * registers; the wave then reduces (cost, raster index) with a 64-bit min.  The search is VALU-bound
 * (111 abs-diff per byte of traffic at R=7), not HBM-bound: each frame byte is read from HBM once.
 */
#include "common.h"
#include "me_kernels.h"

/* 16 bytes at an arbitrary LDS byte address as 4 dwords */
__device__ __forceinline__ void lds16(const uint8_t *p, uint32_t o[4])
{
    const uint32_t a = (uint32_t)(uintptr_t)p;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p - (a & 3));
    const uint32_t sh = a & 3;
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4]; /* tile rows are padded by 8 bytes */
    o[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
    o[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
    o[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
    o[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
}

template <typename PA, typename PB>
__device__ __forceinline__ int sad_bytes(PA a, ptrdiff_t sa, PB b, ptrdiff_t sb, int w, int h)
{
    int s = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            s += abs((int)a[y * sa + x] - (int)b[y * sb + x]);
    return s;
}

/* sum |H8 d H8^T| of one 8x8 block, d = a - b (the sign does not matter) */
template <typename PA, typename PB>
__device__ __forceinline__ int satd8x8(PA a, ptrdiff_t sa, PB b, ptrdiff_t sb)
{
    int t[64];
#pragma unroll
    for (int y = 0; y < 8; y++)
#pragma unroll
        for (int x = 0; x < 8; x++)
            t[8 * y + x] = (int)a[y * sa + x] - (int)b[y * sb + x];
#pragma unroll
    for (int y = 0; y < 8; y++)
#pragma unroll
        for (int span = 1; span < 8; span <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (!(i & span)) {
                    const int p = t[8 * y + i], q = t[8 * y + i + span];
                    t[8 * y + i] = p + q;
                    t[8 * y + i + span] = p - q;
                }
    int sum = 0;
#pragma unroll
    for (int x = 0; x < 8; x++) {
#pragma unroll
        for (int span = 1; span < 8; span <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (!(i & span)) {
                    const int p = t[8 * i + x], q = t[8 * (i + span) + x];
                    t[8 * i + x] = p + q;
                    t[8 * (i + span) + x] = p - q;
                }
#pragma unroll
        for (int i = 0; i < 8; i++)
            sum += abs(t[8 * i + x]);
    }
    return sum;
}

template <typename PA, typename PB>
__device__ __forceinline__ int satd_block(PA a, ptrdiff_t sa, PB b, ptrdiff_t sb, int w, int h)
{
    int s = satd8x8(a, sa, b, sb);
    if (w == 16) {
        s += satd8x8(a + 8, sa, b + 8, sb);
        if (h == 16)
            s += satd8x8(a + 8 * sa, sa, b + 8 * sb, sb) + satd8x8(a + 8 * sa + 8, sa, b + 8 * sb + 8, sb);
    }
    return s;
}

/* ---- function-level batch: one thread per comparison -------------------------------------------- */
__global__ __launch_bounds__(64) void k_me_cmp(int kind, int width, int h, const uint8_t *blk1, const int32_t *off1,
                                               const uint8_t *blk2, const int32_t *off2, ptrdiff_t stride, int32_t *out, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n)
        return;
    const uint8_t *a = blk1 + off1[i], *b = blk2 + off2[i];
    out[i] = kind == FFHIP_ME_SAD ? sad_bytes(a, stride, b, stride, width, h) : satd_block(a, stride, b, stride, width, h);
}

int ffhip_launch_me_cmp(int kind, int width, int h, const uint8_t *blk1, const int32_t *off1, const uint8_t *blk2,
                        const int32_t *off2, ptrdiff_t stride, int32_t *out, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_me_cmp, dim3(cdiv(n, 64)), dim3(64), 0, stream, kind, width, h, blk1, off1, blk2, off2, stride,
                       out, n);
    LAUNCH_CHECK();
    return 0;
}

/* ---- exhaustive search: one wave per macroblock --------------------------------------------------- */
template <int KIND, int MB>
__global__ __launch_bounds__(64) void k_me_esa(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride,
                                               size_t frame_pitch, int R, int16_t *mv_out, uint32_t *cost_out)
{
    extern __shared__ __align__(16) uint8_t lds[];
    constexpr int LOG2 = MB == 16 ? 4 : 3;
    const int bw = width >> LOG2, bh = height >> LOG2;
    const int bx = blockIdx.x, by = blockIdx.y, f = blockIdx.z;
    const int lane = threadIdx.x;
    const int x_mb = bx << LOG2, y_mb = by << LOG2;
    const int lim_x = (bw - 1) << LOG2, lim_y = (bh - 1) << LOG2;
    const int x0 = max(x_mb - R, 0), y0 = max(y_mb - R, 0);
    const int x1 = min(x_mb + R, lim_x), y1 = min(y_mb + R, lim_y);
    const int ncx = x1 - x0 + 1, ncy = y1 - y0 + 1;
    const int wcols = ncx + MB - 1, wrows = ncy + MB - 1;
    const int pitch = ((2 * R + MB + 3) & ~3) + 8; /* dword-aligned rows + room for lds16's over-read */
    uint8_t *cblk = lds;                            /* MB x MB, pitch MB (16-byte aligned rows for MB 16) */
    uint8_t *win = lds + MB * MB;
    const uint8_t *cf = cur + (size_t)f * frame_pitch, *rf = ref + (size_t)f * frame_pitch;

    for (int i = lane; i < MB * MB; i += 64)
        cblk[i] = cf[(ptrdiff_t)(y_mb + i / MB) * stride + x_mb + i % MB];
    for (int i = lane; i < wrows * wcols; i += 64) {
        const int r = i / wcols, c = i - r * wcols;
        win[r * pitch + c] = rf[(ptrdiff_t)(y0 + r) * stride + x0 + c];
    }
    __syncthreads();

    uint32_t best = 0xFFFFFFFFu, best_ci = 0xFFFFFFFFu, cost0 = 0;
    const int ci0 = (y_mb - y0) * ncx + (x_mb - x0);
    for (int ci = lane; ci < ncx * ncy; ci += 64) {
        const int cy = ci / ncx, cx = ci - cy * ncx;
        const uint8_t *cand = win + cy * pitch + cx;
        uint32_t cost;
        if (KIND == FFHIP_ME_SAD) {
            cost = 0;
            if (MB == 16) {
#pragma unroll 4
                for (int y = 0; y < 16; y++) {
                    uint32_t rr[4];
                    lds16(cand + y * pitch, rr);
                    const uint4 cc = *reinterpret_cast<const uint4 *>(cblk + 16 * y);
                    cost = __builtin_amdgcn_sad_u8(cc.x, rr[0], cost);
                    cost = __builtin_amdgcn_sad_u8(cc.y, rr[1], cost);
                    cost = __builtin_amdgcn_sad_u8(cc.z, rr[2], cost);
                    cost = __builtin_amdgcn_sad_u8(cc.w, rr[3], cost);
                }
            } else {
                cost = (uint32_t)sad_bytes(cblk, MB, cand, pitch, MB, MB);
            }
        } else {
            cost = (uint32_t)satd_block(cblk, MB, cand, pitch, MB, MB);
        }
        if (ci == ci0)
            cost0 = cost;
        if (cost < best) {
            best = cost;
            best_ci = (uint32_t)ci;
        }
    }
    /* wave reduction of (cost, raster index): smaller cost, then smaller index */
    unsigned long long key = ((unsigned long long)best << 32) | best_ci;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(key, s, 64);
        key = o < key ? o : key;
    }
    /* the zero-MV cost lives in exactly one lane */
    const int l0 = ci0 & 63;
    cost0 = (uint32_t)__shfl((int)cost0, l0, 64);
    if (lane == 0) {
        const uint32_t mc = (uint32_t)(key >> 32), mi = (uint32_t)key;
        int mvx = x_mb, mvy = y_mb;
        uint32_t c = cost0;
        if (mc < cost0) {
            mvx = x0 + (int)(mi % (uint32_t)ncx);
            mvy = y0 + (int)(mi / (uint32_t)ncx);
            c = mc;
        }
        const size_t b = ((size_t)f * bh + by) * bw + bx;
        mv_out[2 * b] = (int16_t)mvx;
        mv_out[2 * b + 1] = (int16_t)mvy;
        cost_out[b] = c;
    }
}

int ffhip_launch_me_esa(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride, size_t frame_pitch,
                        int nframes, int mb_size, int R, int cost_kind, int16_t *mv_out, uint32_t *cost_out, hipStream_t stream)
{
    const int lg = mb_size == 16 ? 4 : 3;
    const int bw = width >> lg, bh = height >> lg;
    if (nframes <= 0 || bw <= 0 || bh <= 0)
        return 0;
    const int pitch = ((2 * R + mb_size + 3) & ~3) + 8;
    const size_t lds = (size_t)mb_size * mb_size + (size_t)(2 * R + mb_size) * pitch + 32;
    if (lds > 60 * 1024 || nframes > 65535 || bh > 65535) {
        ffhip_set_error("ffhip_me_esa: search_param %d / batch %d outside the supported range", R, nframes);
        return FFHIP_EINVAL;
    }
    const dim3 grid(bw, bh, nframes), block(64);
#define ESA(K, M) hipLaunchKernelGGL((k_me_esa<K, M>), grid, block, lds, stream, cur, ref, width, height, stride, frame_pitch, R, mv_out, cost_out)
    if (cost_kind == FFHIP_ME_SAD) {
        if (mb_size == 16) ESA(FFHIP_ME_SAD, 16); else ESA(FFHIP_ME_SAD, 8);
    } else {
        if (mb_size == 16) ESA(FFHIP_ME_SATD, 16); else ESA(FFHIP_ME_SATD, 8);
    }
#undef ESA
    LAUNCH_CHECK();
    return 0;
}
