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
#include <stdlib.h>

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

/* the half-pel SADs (pix_abs{16,8}_{x2,y2,xy2}_c, me_cmp.c:184-370: blk2 averaged with its right / lower / both neighbours), the sum
 * of squared differences (sse{16,8}_c, :53-104) and the noise-preserving variant (nsse{16,8}_c, :387-440, weight 8: the value the
 * reference uses without an encoder context) */
template <typename PA, typename PB>
__device__ __forceinline__ int cmp_other(int kind, PA a, PB b, ptrdiff_t s, int w, int h)
{
    int r = 0;
    if (kind == FFHIP_ME_NSSE) {
        int score2 = 0;
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                const int d = (int)a[y * s + x] - (int)b[y * s + x];
                r += d * d;
            }
            if (y + 1 < h)
                for (int x = 0; x < w - 1; x++)
                    score2 += abs((int)a[y * s + x] - (int)a[(y + 1) * s + x] - (int)a[y * s + x + 1] + (int)a[(y + 1) * s + x + 1]) -
                              abs((int)b[y * s + x] - (int)b[(y + 1) * s + x] - (int)b[y * s + x + 1] + (int)b[(y + 1) * s + x + 1]);
        }
        return r + abs(score2) * 8;
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int p = a[y * s + x], q0 = b[y * s + x];
            if (kind == FFHIP_ME_SSE) {
                r += (p - q0) * (p - q0);
                continue;
            }
            int q;
            if (kind == FFHIP_ME_SAD_X2) q = (q0 + (int)b[y * s + x + 1] + 1) >> 1;
            else if (kind == FFHIP_ME_SAD_Y2) q = (q0 + (int)b[(y + 1) * s + x] + 1) >> 1;
            else q = (q0 + (int)b[y * s + x + 1] + (int)b[(y + 1) * s + x] + (int)b[(y + 1) * s + x + 1] + 2) >> 2;
            r += abs(p - q);
        }
    return r;
}

/* ---- function-level batch: one thread per comparison -------------------------------------------- */
__global__ __launch_bounds__(64) void k_me_cmp(int kind, int width, int h, const uint8_t *blk1, const int32_t *off1,
                                               const uint8_t *blk2, const int32_t *off2, ptrdiff_t stride, int32_t *out, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n)
        return;
    const uint8_t *a = blk1 + off1[i], *b = blk2 + off2[i];
    out[i] = kind == FFHIP_ME_SAD ? sad_bytes(a, stride, b, stride, width, h) : kind == FFHIP_ME_SATD ? satd_block(a, stride, b, stride, width, h)
                                                                               : cmp_other(kind, a, b, stride, width, h);
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
typedef short me_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short me_u2 __attribute__((ext_vector_type(2)));

/* 8-point Hadamard of 8 bytes down a column (stride bytes apart), packed as four int16 pairs (c0,c1)(c2,c3)... */
template <typename P>
__device__ __forceinline__ uint4 me_hadamard_col(P p, int stride)
{
    int t[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
        t[j] = p[j * stride];
#pragma unroll
    for (int span = 1; span < 8; span <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (!(i & span)) {
                const int a = t[i], b = t[i + span];
                t[i] = a + b;
                t[i + span] = a - b;
            }
    uint4 o;
    o.x = (uint32_t)(t[0] & 0xFFFF) | ((uint32_t)t[1] << 16);
    o.y = (uint32_t)(t[2] & 0xFFFF) | ((uint32_t)t[3] << 16);
    o.z = (uint32_t)(t[4] & 0xFFFF) | ((uint32_t)t[5] << 16);
    o.w = (uint32_t)(t[6] & 0xFFFF) | ((uint32_t)t[7] << 16);
    return o;
}

/*
 * sum |H8 (a - b) H8^T| of one 8x8 block from the VERTICAL Hadamard columns of a and of b (me_hadamard_col): the
 * transform is separable and exact in integers, so H(a) - H(b) = H(a - b) and the order of the two passes is free.
 * The horizontal pass runs across the 8 columns on packed int16 pairs (|coefficient| <= 255 * 64 < 2^15).  Column 0 enters every
 * output with a plus sign (the Hadamard matrix's first column), so 0x8000 added to its four dwords (by the caller: va[0] arrives
 * with its halves' top bits flipped) leaves every
 * output biased by 0x8000 (mod 2^16, no wrap: 16448 .. 49088), and v_sad_u16 against the bias is the absolute sum of two
 * coefficients in one instruction (round 4; before: |p + q| + |p - q| = 2 max(|p|, |q|) through negate / max / max / dot, six
 * instructions per four coefficients where this takes four).
 */
__device__ __forceinline__ uint32_t me_satd8_cols(const uint4 *va, const uint4 *vb, uint32_t acc)
{
    me_s2 d[8][4];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint4 a = va[j], b = vb[j];
        d[j][0] = __builtin_bit_cast(me_s2, a.x) - __builtin_bit_cast(me_s2, b.x);
        d[j][1] = __builtin_bit_cast(me_s2, a.y) - __builtin_bit_cast(me_s2, b.y);
        d[j][2] = __builtin_bit_cast(me_s2, a.z) - __builtin_bit_cast(me_s2, b.z);
        d[j][3] = __builtin_bit_cast(me_s2, a.w) - __builtin_bit_cast(me_s2, b.w);
    }
#pragma unroll
    for (int span = 1; span < 4; span <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (!(i & span))
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const me_s2 p = d[i][m], q = d[i + span][m];
                    d[i][m] = p + q;
                    d[i + span][m] = p - q;
                }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const me_s2 p = d[i][m], q = d[i + 4][m]; /* p carries the bias, q does not */
            acc = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, p + q), 0x80008000u, acc);
            acc = __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, p - q), 0x80008000u, acc);
        }
    return acc;
}

/* a wave's LDS operations execute in order and the waves of a workgroup share nothing: no barrier */
__device__ __forceinline__ void me_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* WPB independent waves (macroblocks bx .. bx + WPB - 1 of a row) per workgroup: a CU holds only about 16 workgroups whatever
 * their size, so one-wave workgroups leave half of its 32 wave slots empty (measured: 4.5 waves per SIMD) */
template <int KIND, int MB, bool SHARE = false, int QUAD = 0, int WPB = 1>
__global__ __launch_bounds__(64 * WPB) void k_me_esa(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride,
                                                     size_t frame_pitch, int R, int16_t *mv_out, uint32_t *cost_out, int lds_per_wave)
{
    extern __shared__ __align__(16) uint8_t lds_all[];
    constexpr int LOG2 = MB == 16 ? 4 : 3;
    const int bw = width >> LOG2, bh = height >> LOG2;
    const int wave = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int bx = blockIdx.x * WPB + wave, by = blockIdx.y, f = blockIdx.z;
    if (WPB > 1 && bx >= bw)
        return;
    uint8_t *lds = lds_all + wave * lds_per_wave;
    const int lane = threadIdx.x & 63;
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

    /* staging with dword loads at byte-exact (unaligned) global addresses, no division: lane -> (row, dword) through a
     * power-of-two dwords-per-row; the dword that would cross the right picture edge is fetched bytewise */
    {
        constexpr int DPR = MB / 4;
        for (int i = lane; i < MB * DPR; i += 64) {
            const int r = i / DPR, j = i % DPR;
            const uint8_t *p = cf + (ptrdiff_t)(y_mb + r) * stride + x_mb + 4 * j;
            uint32_t v;
            __builtin_memcpy(&v, p, 4);
            *reinterpret_cast<uint32_t *>(cblk + r * MB + 4 * j) = v;
        }
        const int dwr = (wcols + 3) >> 2;
        int lg = 3;
        while ((1 << lg) < dwr)
            lg++;
        for (int i = lane; i < (wrows << lg); i += 64) {
            const int r = i >> lg, j = i & ((1 << lg) - 1);
            if (j < dwr) {
                const int xb = x0 + 4 * j;
                const uint8_t *p = rf + (ptrdiff_t)(y0 + r) * stride + xb;
                uint32_t v;
                if (xb + 4 <= width)
                    __builtin_memcpy(&v, p, 4);
                else
                    v = (uint32_t)p[0] | (xb + 1 < width ? (uint32_t)p[1] << 8 : 0) | (xb + 2 < width ? (uint32_t)p[2] << 16 : 0);
                *reinterpret_cast<uint32_t *>(win + r * pitch + 4 * j) = v;
            }
        }
    }
    me_wave_sync();
    /* SATD with SHARED column transforms: the vertical Hadamard of a window column segment (8 rows) serves the 8
     * candidates that contain it, so it is computed once per macroblock into LDS — va: the current block's columns
     * per 8-row band, vb[r][c]: window column c, rows r..r+7 */
    uint4 *va = reinterpret_cast<uint4 *>(lds + ((MB * MB + (2 * R + MB) * pitch + 15) & ~15));
    uint4 *vb = va + (MB / 8) * MB;
    const int vrows = wrows - 7;
    if (SHARE) {
        for (int i = lane; i < (MB / 8) * MB; i += 64) {
            uint4 h = me_hadamard_col(cblk + (i / MB) * 8 * MB + (i % MB), MB);
            if (!(i & 7)) { /* column 0 of an 8x8 block carries me_satd8_cols' bias */
                h.x ^= 0x80008000u; h.y ^= 0x80008000u; h.z ^= 0x80008000u; h.w ^= 0x80008000u;
            }
            va[i] = h;
        }
        for (int i = lane; i < vrows * wcols; i += 64) {
            const int r = i / wcols, c = i - r * wcols;
            vb[i] = me_hadamard_col(win + r * pitch + c, pitch);
        }
        me_wave_sync();
    }

    uint32_t best = 0xFFFFFFFFu, best_ci = 0xFFFFFFFFu, cost0 = 0;
    const int ci0 = (y_mb - y0) * ncx + (x_mb - x0);
    int l0 = ci0 & 63; /* the lane that meets the zero-MV candidate */
    if (QUAD == 3) {
        /* SAD 16x16, four VERTICALLY adjacent candidates per lane (one column position cx, rows cy0 .. cy0 + 3): the 19 window
         * rows they cover are fetched once each — five aligned dwords, funnel-shifted by the lane's byte phase — and every row
         * meets up to four rows of the current block, which sits in scalar registers: 256 v_sad_u8 (full rate) + 76
         * v_alignbyte + 38 LDS reads per four candidates, against 64 + 64 + 32 per candidate in the one-candidate form and the
         * slow-issuing v_qsad_pk_u16_u8 of the horizontal quad form. */
        uint32_t cb[16][4];
#pragma unroll
        for (int y = 0; y < 16; y++) {
            const uint4 c = *reinterpret_cast<const uint4 *>(cblk + 16 * y);
            cb[y][0] = __builtin_amdgcn_readfirstlane(c.x);
            cb[y][1] = __builtin_amdgcn_readfirstlane(c.y);
            cb[y][2] = __builtin_amdgcn_readfirstlane(c.z);
            cb[y][3] = __builtin_amdgcn_readfirstlane(c.w);
        }
        const int ngy = (ncy + 3) >> 2;
        {
            const int cy_z = ci0 / ncx, cx_z = ci0 - cy_z * ncx;
            l0 = ((cy_z >> 2) * ncx + cx_z) & 63;
        }
        for (int g = lane; g < ncx * ngy; g += 64) {
            const int gy = g / ncx, cx = g - gy * ncx, cy0 = 4 * gy;
            const uint32_t sh = (uint32_t)cx & 3u;
            const uint8_t *col = win + (cx & ~3);
            uint32_t cost[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int r = 0; r < 19; r++) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(col + min(cy0 + r, wrows - 1) * pitch);
                const uint4 d = *reinterpret_cast<const uint4 *>(q);
                const uint32_t e = q[4];
                const uint32_t w[4] = { __builtin_amdgcn_alignbyte(d.y, d.x, sh), __builtin_amdgcn_alignbyte(d.z, d.y, sh),
                                        __builtin_amdgcn_alignbyte(d.w, d.z, sh), __builtin_amdgcn_alignbyte(e, d.w, sh) };
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int y = r - k;
                    if (y >= 0 && y < 16) {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            cost[k] = __builtin_amdgcn_sad_u8(cb[y][j], w[j], cost[k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ci = (cy0 + k) * ncx + cx;
                if (cy0 + k < ncy) {
                    if (ci == ci0)
                        cost0 = cost[k];
                    if (cost[k] < best) {
                        best = cost[k];
                        best_ci = (uint32_t)ci;
                    }
                }
            }
        }
    } else if (QUAD) {
        /* SAD 16x16, four horizontally adjacent candidates per lane: their rows are the same five ALIGNED window dwords
         * at byte offsets 0..3, the current block sits in scalar registers (it is the same for every lane) — the
         * one-candidate-per-lane form spent more on unaligned LDS fetches than on differences */
        uint32_t cb[16][4];
#pragma unroll
        for (int y = 0; y < 16; y++) {
            const uint4 c = *reinterpret_cast<const uint4 *>(cblk + 16 * y);
            cb[y][0] = __builtin_amdgcn_readfirstlane(c.x);
            cb[y][1] = __builtin_amdgcn_readfirstlane(c.y);
            cb[y][2] = __builtin_amdgcn_readfirstlane(c.z);
            cb[y][3] = __builtin_amdgcn_readfirstlane(c.w);
        }
        const int ngx = (ncx + 3) >> 2;
        l0 = ((ci0 / ncx) * ngx + ((ci0 % ncx) >> 2)) & 63;
        for (int g = lane; g < ngx * ncy; g += 64) {
            const int cy = g / ngx, cx0 = 4 * (g - cy * ngx);
            /* v_qsad_pk_u16_u8: the four SADs of a 4-byte reference against the four byte positions of an 8-byte
             * source, accumulated in four packed u16 (a 16x16 SAD is at most 65280) — one instruction per block dword */
            uint32_t cost[4] = { 0, 0, 0, 0 };
            if (QUAD == 1) {
                uint64_t acc = 0;
#pragma unroll
                for (int y = 0; y < 16; y++) {
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(win + (cy + y) * pitch + cx0);
                    const uint4 d = *reinterpret_cast<const uint4 *>(q);
                    const uint32_t w[5] = { d.x, d.y, d.z, d.w, q[4] };
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w[j + 1] << 32) | w[j], cb[y][j], acc);
                }
                cost[0] = (uint32_t)(acc & 0xFFFF); cost[1] = (uint32_t)((acc >> 16) & 0xFFFF);
                cost[2] = (uint32_t)((acc >> 32) & 0xFFFF); cost[3] = (uint32_t)(acc >> 48);
            } else {
                /* the same with v_sad_u8 on funnel-shifted dwords (constant shifts) */
#pragma unroll
                for (int y = 0; y < 16; y++) {
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(win + (cy + y) * pitch + cx0);
                    const uint4 d = *reinterpret_cast<const uint4 *>(q);
                    const uint32_t w[5] = { d.x, d.y, d.z, d.w, q[4] };
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        cost[0] = __builtin_amdgcn_sad_u8(cb[y][j], w[j], cost[0]);
                        cost[1] = __builtin_amdgcn_sad_u8(cb[y][j], __builtin_amdgcn_alignbyte(w[j + 1], w[j], 1), cost[1]);
                        cost[2] = __builtin_amdgcn_sad_u8(cb[y][j], __builtin_amdgcn_alignbyte(w[j + 1], w[j], 2), cost[2]);
                        cost[3] = __builtin_amdgcn_sad_u8(cb[y][j], __builtin_amdgcn_alignbyte(w[j + 1], w[j], 3), cost[3]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ci = cy * ncx + cx0 + k;
                if (cx0 + k < ncx) {
                    if (ci == ci0)
                        cost0 = cost[k];
                    if (cost[k] < best) {
                        best = cost[k];
                        best_ci = (uint32_t)ci;
                    }
                }
            }
        }
    } else
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
        } else if (SHARE) {
            cost = 0;
#pragma unroll
            for (int sy = 0; sy < MB / 8; sy++)
#pragma unroll
                for (int sx = 0; sx < MB / 8; sx++)
                    cost = me_satd8_cols(va + sy * MB + 8 * sx, vb + (cy + 8 * sy) * wcols + cx + 8 * sx, cost);
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

/*
 * k_me_esa_g — SAD, 16x16 macroblocks: the search of NMB = 8 horizontally adjacent macroblocks by one workgroup.
 *
 * What the one-wave-per-macroblock forms above pay per candidate beside its 64 v_sad_u8 — the window's LDS reads, the funnel
 * shifts that bring a byte phase into place, the staging, the wave reduction — is shared here along every axis that offers it:
 *   - one reference window, (16 + 2R) rows x (8 * 16 + 2R) columns, is staged once for the eight macroblocks (their windows
 *     overlap by 2R columns out of 16 + 2R);
 *   - a lane takes a 4 x 4 GROUP of candidates of one macroblock: four byte phases of the same aligned dwords times four vertical
 *     offsets.  The 19 window rows under the group are read once each (five dwords, two LDS instructions), shifted once into the
 *     phases 1..3 (twelve v_alignbyte), and every shifted dword meets the (up to four) rows of the current block it lies under:
 *     1024 v_sad_u8 + 228 v_alignbyte + 38 LDS reads per 16 candidates, against 1024 + 768 + 512 in the one-candidate form;
 *   - the current block's rows come from LDS as they are needed (one broadcast read per window row, four rows live at a time), so
 *     the lanes of a wave may belong to different macroblocks and a wave is full whatever R is: R = 7 has 16 groups per
 *     macroblock, 128 per workgroup;
 *   - the winner is ONE 32-bit key per candidate, cost << 12 | index, index 0 for the zero motion vector and raster index + 1 for
 *     the others: its minimum is the reference's "first minimum in raster order wins, the zero vector wins ties"
 *     (libavfilter/motion_estimation.c:79-100); a v_or3 masks candidates outside the picture, ds_min_u32 merges the lanes.
 * R <= 30 (the index must fit 12 bits; a 16x16 SAD fits 16).
 */
#define ESA_G_NMB 8
__global__ __launch_bounds__(256) void k_me_esa_g(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride, size_t frame_pitch,
                                                  int R, int16_t *mv_out, uint32_t *cost_out, int pitch)
{
    extern __shared__ __align__(16) uint8_t lds_all[];
    const int bw = width >> 4, bh = height >> 4;
    const int bx0 = blockIdx.x * ESA_G_NMB, by = blockIdx.y, f = blockIdx.z;
    const int nmb = min(ESA_G_NMB, bw - bx0);
    const int tid = threadIdx.x, T = blockDim.x;
    const int lim_x = (bw - 1) << 4, lim_y = (bh - 1) << 4;
    const int y_mb = by << 4, y0 = max(y_mb - R, 0), y1 = min(y_mb + R, lim_y), ncy = y1 - y0 + 1, wrows = ncy + 15;
    const int xf = bx0 << 4, wx0 = max(xf - R, 0), wx1 = min(xf + ((nmb - 1) << 4) + R, lim_x) + 15; /* window columns wx0 .. wx1 */
    uint32_t *keys = reinterpret_cast<uint32_t *>(lds_all);          /* [NMB] */
    uint8_t *cblk = lds_all + 64;                                    /* [NMB][16][16] */
    uint8_t *win = cblk + ESA_G_NMB * 256;                           /* [wrows][pitch] */
    const uint8_t *cf = cur + (size_t)f * frame_pitch, *rf = ref + (size_t)f * frame_pitch;
    if (tid < ESA_G_NMB)
        keys[tid] = 0xFFFFFFFFu;
    /* staging: 16-byte loads at byte-exact addresses (rows on a 16-chunk grid: no division); the chunk that would cross the right
     * picture edge goes bytewise.  `pitch` is a multiple of 16. */
    {
        const int n16 = (wx1 - wx0 + 1 + 15) >> 4;              /* <= 12: R <= 30 */
        const int ncur = nmb * 16, tot = ncur + (wrows << 4);
        for (int i = tid; i < tot; i += T) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i < ncur) {   /* the current blocks: row r of macroblock m */
                const int m = i >> 4, r = i & 15;
                __builtin_memcpy(&v, cf + (ptrdiff_t)(y_mb + r) * stride + xf + 16 * m, 16);
                *reinterpret_cast<uint4 *>(cblk + 16 * i) = v;
            } else {
                const int r = (i - ncur) >> 4, j = (i - ncur) & 15, xb = wx0 + 16 * j;
                if (j < n16) {
                    const uint8_t *p = rf + (ptrdiff_t)(y0 + r) * stride + xb;
                    if (xb + 16 <= width) {
                        __builtin_memcpy(&v, p, 16);
                    } else {
                        uint32_t w[4] = { 0, 0, 0, 0 };
                        for (int b = 0; b < 16 && xb + b < width; b++)
                            w[b >> 2] |= (uint32_t)p[b] << (8 * (b & 3));
                        v = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                    *reinterpret_cast<uint4 *>(win + r * pitch + 16 * j) = v;
                }
            }
        }
    }
    __syncthreads();
    /* groups per macroblock: columns are grouped on the window's dword grid, so a macroblock whose first candidate column is
     * not a multiple of four from the window's start (left picture edge only) has its candidates begin inside the first group */
    const int ngx = (2 * R + 1 + (wx0 == 0 ? (-R & 3) : 0) + 3) >> 2, ngy = (2 * R + 1 + 3) >> 2, G = ngx * ngy;
    for (int it = tid; it < nmb * G; it += T) {
        const int m = it / G, g = it - m * G, gy = g / ngx, gx = g - gy * ngx;
        const int x_mb = xf + (m << 4), x0 = max(x_mb - R, 0), x1 = min(x_mb + R, lim_x), ncx = x1 - x0 + 1;
        const int xo = x0 - wx0, off = xo & 3, cy0 = 4 * gy;
        const uint8_t *cbm = cblk + m * 256;
        uint32_t cb[4][4]; /* the four rows of the current block a window row lies under: row y lives in cb[y & 3] */
        const uint8_t *col = win + (xo - off) + 4 * gx;
        uint32_t cost[4][4]; /* [vertical offset][byte phase] */
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                cost[k][j] = 0;
#pragma unroll
        for (int r = 0; r < 19; r++) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(col + min(cy0 + r, wrows - 1) * pitch);
            const uint4 d = *reinterpret_cast<const uint4 *>(q);
            const uint32_t e = q[4];
            const uint32_t w0[5] = { d.x, d.y, d.z, d.w, e };
            if (r < 16) { /* a broadcast read: every lane of the macroblock asks for the same 16 bytes */
                const uint4 c = *reinterpret_cast<const uint4 *>(cbm + 16 * r);
                cb[r & 3][0] = c.x; cb[r & 3][1] = c.y; cb[r & 3][2] = c.z; cb[r & 3][3] = c.w;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    w[i] = j ? __builtin_amdgcn_alignbyte(w0[i + 1], w0[i], j) : w0[i];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int y = r - k;
                    if (y >= 0 && y < 16) {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            cost[k][j] = __builtin_amdgcn_sad_u8(cb[y & 3][i], w[i], cost[k][j]);
                    }
                }
            }
        }
        /* keys: candidate (cx, cy) = (4 gx + j - off, cy0 + k), valid inside [0, ncx) x [0, ncy) */
        const int zx = x_mb - x0, zy = y_mb - y0; /* the zero vector's candidate */
        uint32_t best = 0xFFFFFFFFu;
        uint32_t mx[4], my[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            mx[j] = (uint32_t)(4 * gx + j - off) < (uint32_t)ncx ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 4; k++)
            my[k] = cy0 + k < ncy ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int cy = cy0 + k;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int cx = 4 * gx + j - off;
                const uint32_t idx = (cx == zx && cy == zy) ? 0u : (uint32_t)(cy * ncx + cx + 1);
                best = min(best, ((cost[k][j] << 12) + idx) | mx[j] | my[k]);
            }
        }
        atomicMin(&keys[m], best);
    }
    __syncthreads();
    if (tid < nmb) {
        const uint32_t key = keys[tid], idx = key & 0xFFF;
        const int x_mb = xf + (tid << 4), x0 = max(x_mb - R, 0), x1 = min(x_mb + R, lim_x), ncx = x1 - x0 + 1;
        int mvx = x_mb, mvy = y_mb;
        if (idx) {
            mvx = x0 + (int)((idx - 1) % (uint32_t)ncx);
            mvy = y0 + (int)((idx - 1) / (uint32_t)ncx);
        }
        const size_t b = ((size_t)f * bh + by) * bw + bx0 + tid;
        mv_out[2 * b] = (int16_t)mvx;
        mv_out[2 * b + 1] = (int16_t)mvy;
        cost_out[b] = key >> 12;
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
    /* SAD: four macroblocks (waves) per workgroup, FFHIP_ME_WPB=1 for the one-wave form */
    const char *ew = FFHIP_KNOB("FFHIP_ME_WPB");
    const int lpw = (int)((lds + 15) & ~(size_t)15);
    const bool w4 = !(ew && ew[0] == '1') && (size_t)lpw * 4 <= 64 * 1024; /* large search ranges: one wave, one window */
    const dim3 grid4(cdiv(bw, 4), bh, nframes), block4(256);
#define ESA(K, M) hipLaunchKernelGGL((k_me_esa<K, M>), grid, block, lds, stream, cur, ref, width, height, stride, frame_pitch, R, mv_out, cost_out, 0)
#define ESA4(K, M, Q) hipLaunchKernelGGL((k_me_esa<K, M, false, Q, 4>), grid4, block4, (size_t)lpw * 4, stream, cur, ref, width, height, stride, frame_pitch, R, mv_out, cost_out, lpw)
    if (cost_kind == FFHIP_ME_SAD) {
        /* measured (4K, 8 frame pairs): R = 7: one candidate per lane 448 M, quad + v_qsad_pk_u16_u8 480 M, quad + v_sad_u8 433 M
         * MB-searches/s; R = 16: 122 / 116 / 114 M.  v_qsad_pk_u16_u8 issues at about 1/14 rate on gfx950, which eats what
         * the four-fold saving in LDS reads buys; the quad form is the default only while the candidates fit one pass.
         * FFHIP_ME_SAD_QUAD = 0 / 1 / 2 / 3 forces a form.  Round 2 measured two more hypotheses (tools/run_esa.py): four VERTICALLY
         * adjacent candidates per lane sharing their window rows (form 3: a quarter of the LDS reads and alignbytes, v_sad_u8 only):
         * 418 M at R = 7, 100 M at R = 16 — slower than both; and four macroblock waves per workgroup instead of one (a CU's workgroup
         * slots): + 2..6 %, kept.  Every form sits at 60 % VALU issue with the rest in LDS round trips per candidate row. */
        const char *eq = FFHIP_KNOB("FFHIP_ME_SAD_QUAD");
        const bool one_pass = ((2 * R + 1 + 3) / 4) * (2 * R + 1) <= 64;
        if (mb_size == 16 && !eq && R <= 30) {
            /* round 3: eight macroblocks per workgroup, 4 x 4 candidate groups per lane (k_me_esa_g) */
            /* rows of whole 16-byte chunks + the fifth dword of the last group; an odd number of chunks, so that the four rows a
             * wave's groups start on (4 apart) fall on different banks */
            const int gpitch = (((ESA_G_NMB * 16 + 2 * R + 8 + 15) >> 4) | 1) << 4;
            const size_t glds = 64 + ESA_G_NMB * 256 + (size_t)(2 * R + 16) * gpitch;
            const int ng = ((2 * R + 1 + 3 + 3) / 4) * ((2 * R + 1 + 3) / 4) * ESA_G_NMB;
            const int threads = ng >= 256 ? 256 : (ng + 63) & ~63;
            hipLaunchKernelGGL(k_me_esa_g, dim3(cdiv(bw, ESA_G_NMB), bh, nframes), dim3(threads), glds, stream, cur, ref, width, height, stride,
                               frame_pitch, R, mv_out, cost_out, gpitch);
        } else if (mb_size == 16 && eq && eq[0] == '3') {
            if (w4) ESA4(FFHIP_ME_SAD, 16, 3);
            else hipLaunchKernelGGL((k_me_esa<FFHIP_ME_SAD, 16, false, 3>), grid, block, lds, stream, cur, ref, width, height, stride,
                                    frame_pitch, R, mv_out, cost_out, 0);
        } else if (mb_size == 16 && !eq && !one_pass) {
            if (w4) ESA4(FFHIP_ME_SAD, 16, 0); else ESA(FFHIP_ME_SAD, 16);
        } else if (mb_size == 16 && eq && eq[0] == '2') {
            hipLaunchKernelGGL((k_me_esa<FFHIP_ME_SAD, 16, false, 2>), grid, block, lds, stream, cur, ref, width, height, stride,
                               frame_pitch, R, mv_out, cost_out, 0);
        } else if (mb_size == 16 && !(eq && eq[0] == '0')) {
            if (w4) ESA4(FFHIP_ME_SAD, 16, 1);
            else hipLaunchKernelGGL((k_me_esa<FFHIP_ME_SAD, 16, false, 1>), grid, block, lds, stream, cur, ref, width, height, stride,
                                    frame_pitch, R, mv_out, cost_out, 0);
        } else if (mb_size == 16) {
            if (w4) ESA4(FFHIP_ME_SAD, 16, 0); else ESA(FFHIP_ME_SAD, 16);
        } else {
            if (w4) ESA4(FFHIP_ME_SAD, 8, 0); else ESA(FFHIP_ME_SAD, 8);
        }
    } else {
        /* round 5: the 2-D transform as a dense int8 product on the matrix cores (me_satd.hip); FFHIP_ME_SATD_SHARE set: the VALU forms —
         * shared column transforms when their LDS plane fits (R <= 24 at 16x16), FFHIP_ME_SATD_SHARE=0: per-candidate */
        const char *es = FFHIP_KNOB("FFHIP_ME_SATD_SHARE");
        if (!es && ffhip_launch_me_esa_satd_mx(cur, ref, width, height, stride, frame_pitch, nframes, mb_size, R, mv_out, cost_out, stream)) {
            LAUNCH_CHECK();
            return 0;
        }
        const size_t vsz = ((size_t)(mb_size / 8) * mb_size + (size_t)(2 * R + mb_size - 7) * (2 * R + mb_size)) * 16;
        const size_t lds_s = ((lds + 15) & ~(size_t)15) + vsz;
        if (lds_s <= 64 * 1024 && !(es && es[0] == '0')) {
#define ESAS(M) hipLaunchKernelGGL((k_me_esa<FFHIP_ME_SATD, M, true, 0>), grid, block, lds_s, stream, cur, ref, width, height, stride, frame_pitch, R, mv_out, cost_out, 0)
            if (mb_size == 16) ESAS(16); else ESAS(8);
#undef ESAS
        } else {
            if (mb_size == 16) ESA(FFHIP_ME_SATD, 16); else ESA(FFHIP_ME_SATD, 8);
        }
    }
#undef ESA
#undef ESA4
    LAUNCH_CHECK();
    return 0;
}
