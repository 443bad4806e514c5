/*
 * hevc_idct.hip — HEVC inverse transforms, 8-bit, batched.
 *
 * Bit-exact restatement of idct_{4,8,16,32}, idct_*_dc, transform_4x4_luma and add_residual
 * (libavcodec/hevc/dsp_template.c:46-59,155-188,192-300).  A 1-D pass of the reference is a partial butterfly that sums
 * exact integers, i.e. the product with the HEVC core matrix restricted to the coefficients its `end` limits keep
 * (odd k < end; k = 2 mod 4 of the 32-point transform: k/2 < end/2; the inner 8-/4-point stages always), followed by
 * av_clip_int16((sum + add) >> shift).  col_limit shrinks by 4 after columns 4, 8, ... of the first pass.
 *
 * GPU design: a transform unit is N lanes (N = size), 64 / N units per wave, the block staged in wave-private LDS.
 * Lane (unit, i) transforms column i, then row i: it zeroes the inputs its limit drops (a per-lane select, the limits
 * differ per unit and column) and accumulates the even and the odd basis functions separately, E_j and O_j, so that
 * outputs j and N-1-j share their multiplies, two per v_dot2_i32_i16 (N*N/4 instructions per vector).  The matrix
 * entries are wave-uniform: int16 pairs in a __constant__ table, read with scalar loads and used as SGPR operands.  Residuals are written back
 * in place (the reference leaves them in coeffs) and added to the picture with packed byte stores.
 */
#include <mutex>
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "h264_kernels.h"

/* |64 sqrt2 cos(m pi / 64)| as the standard rounds it; T32[k][i] = +-g[fold((2i+1)k mod 128)] */
static int8_t hevc_t32_host[32][32];
static std::once_flag hevc_tab_once;
/* the same matrix as int16 PAIRS for v_dot2_i32_i16, per transform size N (offset hevc_pk_off(N)): entry [j][q][0] =
 * (T_N[4q][j], T_N[4q+2][j]) (even basis functions), [j][q][1] = (T_N[4q+1][j], T_N[4q+3][j]) (odd), j < N/2, q < N/4 */
__constant__ uint32_t hevc_pk[352];
static uint32_t hevc_pk_host[352];
__host__ __device__ constexpr int hevc_pk_off(int n) { return n == 4 ? 0 : n == 8 ? 4 : n == 16 ? 4 + 16 : 4 + 16 + 64; }

static void hevc_build_table()
{
    static const int g[32] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                               64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4 };
    for (int k = 0; k < 32; k++)
        for (int i = 0; i < 32; i++) {
            const int m = ((2 * i + 1) * k) & 127;
            int v;
            if (k == 0) v = 64;
            else if (m < 32) v = g[m];
            else if (m == 32 || m == 96) v = 0;
            else if (m < 64) v = -g[64 - m];
            else if (m < 96) v = -g[m - 64];
            else v = g[128 - m];
            hevc_t32_host[k][i] = (int8_t)v;
        }
    for (int n = 4; n <= 32; n *= 2) {
        const int sc = 32 / n;
        uint32_t *t = hevc_pk_host + hevc_pk_off(n);
        for (int j = 0; j < n / 2; j++)
            for (int q = 0; q < n / 4; q++)
                for (int odd = 0; odd < 2; odd++) {
                    const int a = hevc_t32_host[(4 * q + odd) * sc][j], b = hevc_t32_host[(4 * q + 2 + odd) * sc][j];
                    t[(j * (n / 4) + q) * 2 + odd] = (uint32_t)(a & 0xFFFF) | ((uint32_t)b << 16);
                }
    }
}

__device__ __forceinline__ int hevc_clip16(int v) { return min(max(v, -32768), 32767); }

/* one 1-D pass over the vector at src (stride sstep int16) into dst (stride dstep): N outputs */
template <int N>
__device__ __forceinline__ void hevc_pass(int16_t *dst, int dstep, const int16_t *src, int sstep, int end, int shift)
{
    constexpr int SC = 32 / N;
    int s[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        bool keep = true;
        if (N > 4) {
            if (k & 1) keep = k < end;
            else if (N == 32 && (k & 3) == 2) keep = (k >> 1) < (end >> 1);
        }
        const int v = src[k * sstep];
        s[k] = keep ? v : 0;
    }
    const int add = 1 << (shift - 1);
    /* inputs as int16 pairs (s[4q], s[4q+2]) and (s[4q+1], s[4q+3]): two multiply-adds per v_dot2_i32_i16 against the
     * wave-uniform coefficient pairs */
    typedef short hv_s2 __attribute__((ext_vector_type(2)));
    hv_s2 pe[N / 4], po[N / 4];
#pragma unroll
    for (int q = 0; q < N / 4; q++) {
        pe[q] = hv_s2{ (short)s[4 * q], (short)s[4 * q + 2] };
        po[q] = hv_s2{ (short)s[4 * q + 1], (short)s[4 * q + 3] };
    }
    const uint32_t *tab = hevc_pk + hevc_pk_off(N);
    (void)SC;
#pragma unroll
    for (int j = 0; j < N / 2; j++) {
        int e = 0, o = 0;
#pragma unroll
        for (int q = 0; q < N / 4; q++) {
            e = __builtin_amdgcn_sdot2(pe[q], __builtin_bit_cast(hv_s2, tab[(j * (N / 4) + q) * 2]), e, false);
            o = __builtin_amdgcn_sdot2(po[q], __builtin_bit_cast(hv_s2, tab[(j * (N / 4) + q) * 2 + 1]), o, false);
        }
        dst[j * dstep] = (int16_t)hevc_clip16((e + o + add) >> shift);
        dst[(N - 1 - j) * dstep] = (int16_t)hevc_clip16((e - o + add) >> shift);
    }
}

__device__ __forceinline__ void hevc_dst4(int16_t *dst, const int16_t *src, int step, int shift)
{
    const int add = 1 << (shift - 1);
    const int s0 = src[0], s1 = src[step], s2 = src[2 * step], s3 = src[3 * step];
    const int c0 = s0 + s2, c1 = s2 + s3, c2 = s0 - s3, c3 = 74 * s1;
    dst[0]        = (int16_t)hevc_clip16((29 * c0 + 55 * c1 + c3 + add) >> shift);
    dst[step]     = (int16_t)hevc_clip16((55 * c2 - 29 * c1 + c3 + add) >> shift);
    dst[2 * step] = (int16_t)hevc_clip16((74 * (s0 - s2 + s3) + add) >> shift);
    dst[3 * step] = (int16_t)hevc_clip16((55 * c0 + 29 * c2 - c3 + add) >> shift);
}

__device__ __forceinline__ void hevc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* bd: the depth the template was instantiated for in the reference (hevc/dsp.c:133-196): it sets the second-pass shift 20 - bd, the
 * DC shift 14 - bd, dequant's 15 - bd - log2 and the pixel type / clip of add_residual (uint16_t above 8 bits, stride in bytes) */
template <int LOG2>
__global__ __launch_bounds__(256) void k_hevc_idct(int kind, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipHevcTU *tus, int n, int bd)
{
    constexpr int N = 1 << LOG2, UPW = 64 / N; /* units per wave */
    __shared__ __align__(16) int16_t lds[4][64 * N];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int u0 = (blockIdx.x * 4 + wave) * UPW;
    if (u0 >= n)
        return;
    int16_t *blk = lds[wave];
    const int ul = lane / N, i = lane % N; /* my unit of the wave, my column / row */
    const int u = u0 + ul;
    const bool live = u < n;
    const FFHipHevcTU tu = tus[live ? u : u0];
    int16_t *cg = coeffs + tu.coeff_offset;

    /* ---- stage the wave's units: UPW blocks of N*N int16, 16 bytes per lane and step where the block is 16-byte aligned (one
     *      descriptor read and one load per lane for 8x8), else as dwords ---- */
    constexpr int DW = N * N / 2; /* dwords per unit */
    constexpr int Q4 = DW / 4;    /* 16-byte pieces per unit */
    for (int t = lane; t < UPW * Q4; t += 64) {
        const int b = t / Q4, w = t % Q4;
        if (u0 + b < n) {
            const int16_t *g = coeffs + tus[u0 + b].coeff_offset;
            if (!(reinterpret_cast<uintptr_t>(g) & 15)) {
                reinterpret_cast<uint4 *>(blk)[t] = reinterpret_cast<const uint4 *>(g)[w];
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    reinterpret_cast<uint32_t *>(blk)[4 * t + k] = reinterpret_cast<const uint32_t *>(g)[4 * w + k];
            }
        }
    }
    hevc_wave_sync();
    int16_t *mine = blk + ul * N * N;
    if (kind == FFHIP_HEVC_IDCT) {
        const int limit = min(tu.col_limit, N);
        /* first pass: column i; limit2 has shrunk by 4 for every column 4, 8, ... before mine while it was < N */
        int limit2 = min(tu.col_limit + 4, N);
        for (int c = 4; c < i; c += 4)
            if (limit2 < N)
                limit2 -= 4;
        hevc_pass<N>(mine + i, N, mine + i, N, limit2, 7);
        hevc_wave_sync();
        hevc_pass<N>(mine + i * N, 1, mine + i * N, 1, limit, 20 - bd);
    } else if (kind == FFHIP_HEVC_IDCT_DC) {
        const int v = ((((int)mine[0] + 1) >> 1) + (1 << (13 - bd))) >> (14 - bd);
        hevc_wave_sync();
#pragma unroll
        for (int k = 0; k < N; k++)
            mine[i * N + k] = (int16_t)v;
    } else if (kind == FFHIP_HEVC_DST_4X4) {
        if (N == 4) {
            hevc_dst4(mine + i, mine + i, 4, 7);
            hevc_wave_sync();
            hevc_dst4(mine + 4 * i, mine + 4 * i, 1, 20 - bd);
        }
    } else if (kind == FFHIP_HEVC_DEQUANT) {
        /* dequant (hevc/dsp_template.c:110-143): (c + 2^(shift-1)) >> shift, shift = 15 - bd - log2; nothing at shift 0, and above
         * 10 bits a negative shift is a left shift of the coefficient read as uint16; lane i = row i */
        const int shift = 15 - bd - LOG2;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const int c = mine[i * N + k];
            mine[i * N + k] = (int16_t)(shift > 0 ? (c + (1 << (shift - 1))) >> shift : shift < 0 ? (int)((uint32_t)(uint16_t)c << -shift) : c);
        }
    } else if (kind == FFHIP_HEVC_RDPCM_H || kind == FFHIP_HEVC_RDPCM_V) {
        /* transform_rdpcm (hevc/dsp_template.c:85-105): running sums in int16 arithmetic along a row (mode 0: lane i = row i)
         * or down a column (mode 1: lane i = column i) */
        const int step = kind == FFHIP_HEVC_RDPCM_H ? 1 : N;
        int16_t *v = mine + (kind == FFHIP_HEVC_RDPCM_H ? i * N : i);
        int acc = v[0];
#pragma unroll
        for (int k = 1; k < N; k++) {
            acc = (int16_t)(acc + v[k * step]);
            v[k * step] = (int16_t)acc;
        }
    }
    hevc_wave_sync();
    /* ---- residual back in place; picture += residual (row i of my unit) ---- */
    if (kind != FFHIP_HEVC_ADD_ONLY) {
        for (int t = lane; t < UPW * Q4; t += 64) {
            const int b = t / Q4, w = t % Q4;
            if (u0 + b < n) {
                int16_t *g = coeffs + tus[u0 + b].coeff_offset;
                if (!(reinterpret_cast<uintptr_t>(g) & 15)) {
                    reinterpret_cast<uint4 *>(g)[w] = reinterpret_cast<const uint4 *>(blk)[t];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        reinterpret_cast<uint32_t *>(g)[4 * w + k] = reinterpret_cast<const uint32_t *>(blk)[4 * t + k];
                }
            }
        }
    }
    (void)cg;
    if (dst) {
        /* picture += residual, one row per lane.  Rows are handed out ROW-major over the wave's units (lane = row * UPW + unit): the
         * units of a wave are usually neighbours in the picture, so consecutive lanes then touch consecutive pieces of one picture row
         * (unit-major, lanes 0 .. N-1 walk down the N rows of one block: N short pieces of N different lines per N lanes) */
        const int ul2 = UPW > 1 ? lane % UPW : 0, i2 = UPW > 1 ? lane / UPW : lane;
        const int u2 = u0 + ul2;
        if (u2 < n) {
            const int32_t doff = UPW > 1 ? tus[u2].dst_offset : tu.dst_offset;
            if (doff >= 0) {
                const int16_t *r = blk + ul2 * N * N + i2 * N;
                int z[N];
#pragma unroll
                for (int x = 0; x < N; x++)
                    z[x] = r[x];
                ffhip_add_row<N>(dst + doff + (ptrdiff_t)i2 * stride, z, bd);
            }
        }
    }
}

/* ================================================================================================== */
/*
 * k_hevc_idct32_mfma — the 32x32 inverse transform on the matrix cores (north_star: "MFMA only if the product genuinely becomes
 * a dense contraction" — this one is: a 32x32 int8 matrix, |T| <= 90, times a 32x32 int16 block, twice).
 * One wave per transform unit, v_mfma_i32_32x32x32_i8 (A: lane = row l & 31, 16 K-bytes of group l >> 5; B: lane = column; D: lane =
 * column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5) for r = 0..15).  int16 data is split into a signed high byte and a low byte
 * biased by -128 (i8 MFMA is signed x signed): sum T x = 256 sum T hi + sum T (lo - 128) + 128 sum T, the last term a per-output
 * constant seeded into the accumulator between the two MFMAs, so a pass is  acc = mfma(hi); acc = (acc << 8) + bias; acc = mfma(lo').
 *   pass 1 (columns)  D1 = X^T . T: lane (c, g) loads X[16 g + s][c], s = 0..15, zeroes what its column's limit drops; the result
 *                     leaves lane (j, g) holding Y[j][c] for the 16 c of ITS row set — which is exactly an A operand of
 *   pass 2 (rows)     D2 = Y . T  with K enumerated in that order: the B table of pass 2 is T with its rows permuted to match, so the
 *                     first pass's registers go back in after >> 7, clip, the second limit's zeroing and the byte split — no LDS, no
 *                     transposition.
 * Exact: int32 accumulation of exact products, |sums| <= 32 * 90 * 32768 < 2^31.  Residuals are written back in place, then added
 * to the picture (8-bit or 16-bit samples).  The VALU kernel above needs N^2/4 = 256 v_dot2 per vector with a scalar coefficient
 * load behind each (measured 14 % of the dot issue rate); here the contraction costs 4 MFMAs per block.
 */
typedef int hm_i4 __attribute__((ext_vector_type(4)));
typedef int hm_i16 __attribute__((ext_vector_type(16)));
struct HevcMfmaTab { int8_t b1[64][16], b2[64][16]; int32_t sum[32]; }; /* sum[j] = 128 * sum_k T[k][j] */
static std::once_flag hm_once;

__device__ __forceinline__ bool hm_keep(int k, int end) /* hevc_pass<32>'s rule */
{
    return (k & 1) ? k < end : ((k & 3) == 2 ? (k >> 1) < (end >> 1) : true);
}
/* 16 int16 values -> the high-byte plane and the (low byte - 128) plane, 4 bytes per dword in order */
__device__ __forceinline__ void hm_split(const int (&v)[16], hm_i4 &hi, hm_i4 &lo)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t p01 = ((uint32_t)v[4 * q] & 0xFFFFu) | ((uint32_t)v[4 * q + 1] << 16);
        const uint32_t p23 = ((uint32_t)v[4 * q + 2] & 0xFFFFu) | ((uint32_t)v[4 * q + 3] << 16);
        hi[q] = (int)__builtin_amdgcn_perm(p23, p01, 0x07050301u);
        lo[q] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
    }
}

__global__ __launch_bounds__(256) void k_hevc_idct32_mfma(int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipHevcTU *tus, int n, int bd,
                                                          const HevcMfmaTab *tab)
{
    /* the matrix core wants a lane to hold (part of) a COLUMN of the unit and hands back columns; memory wants rows.  Both
     * transpositions go through 2 KB of LDS per wave: a lane moves 32 contiguous bytes of a coefficient row in and out and 16
     * picture samples, instead of sixteen 2-byte accesses each way plus sixteen byte-wide read-modify-writes */
    __shared__ __align__(16) int16_t lds[4][1024];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + wave;
    if (u >= n)
        return;
    const FFHipHevcTU tu = tus[u];
    const int col_limit = __builtin_amdgcn_readfirstlane(tu.col_limit);
    int16_t *X = coeffs + __builtin_amdgcn_readfirstlane(tu.coeff_offset);
    int16_t *L = lds[wave];
    /* row view: lane = (row rr, half rh): 16 samples */
    const int rr = lane >> 1, rh = lane & 1;
    int16_t *rowp = X + rr * 32 + 16 * rh;
    const bool al16 = !(reinterpret_cast<uintptr_t>(X) & 15);
    if (al16) {
        reinterpret_cast<uint4 *>(L + rr * 32 + 16 * rh)[0] = reinterpret_cast<const uint4 *>(rowp)[0];
        reinterpret_cast<uint4 *>(L + rr * 32 + 16 * rh)[1] = reinterpret_cast<const uint4 *>(rowp)[1];
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++)
            reinterpret_cast<uint32_t *>(L + rr * 32 + 16 * rh)[k] = reinterpret_cast<const uint32_t *>(rowp)[k];
    }
    hevc_wave_sync();
    const int c = lane & 31, g = lane >> 5;
    const hm_i4 B1 = reinterpret_cast<const hm_i4 *>(tab->b1)[lane], B2 = reinterpret_cast<const hm_i4 *>(tab->b2)[lane];
    const int sum_t = tab->sum[c];
    /* ---- pass 1: my column c, rows 16 g .. 16 g + 15; limit2 has shrunk by 4 for every column 4, 8, ... before mine ---- */
    int limit2 = min(col_limit + 4, 32);
    for (int q = 4; q < c; q += 4)
        if (limit2 < 32)
            limit2 -= 4;
    int v[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const int k = 16 * g + s, x = L[k * 32 + c];
        v[s] = hm_keep(k, limit2) ? x : 0;
    }
    hm_i4 ahi, alo;
    hm_split(v, ahi, alo);
    hm_i16 acc = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, B1, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++)
        acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)(sum_t + 64));
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, B1, acc, 0, 0, 0);
    /* ---- Y[j = c][cc], cc = (r & 3) + 8 (r >> 2) + 4 g: >> 7, clip, the second pass's limit ---- */
    const int limit = min(col_limit, 32);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int cc = (r & 3) + 8 * (r >> 2) + 4 * g;
        const int y = hevc_clip16(acc[r] >> 7);
        v[r] = hm_keep(cc, limit) ? y : 0;
    }
    hm_split(v, ahi, alo);
    const int shift2 = 20 - bd;
#pragma unroll
    for (int r = 0; r < 16; r++)
        acc[r] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, B2, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++)
        acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)(sum_t + (1 << (shift2 - 1))));
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, B2, acc, 0, 0, 0);
    /* ---- Z[j][m = c], j = (r & 3) + 8 (r >> 2) + 4 g: back through LDS into rows; residual in place, picture += residual ---- */
    hevc_wave_sync(); /* every lane has read its inputs */
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * g;
        L[j * 32 + c] = (int16_t)hevc_clip16(acc[r] >> shift2);
    }
    hevc_wave_sync();
    const uint4 r0 = reinterpret_cast<const uint4 *>(L + rr * 32 + 16 * rh)[0], r1 = reinterpret_cast<const uint4 *>(L + rr * 32 + 16 * rh)[1];
    const uint32_t rw[8] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w };
    if (al16) {
        reinterpret_cast<uint4 *>(rowp)[0] = r0;
        reinterpret_cast<uint4 *>(rowp)[1] = r1;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++)
            reinterpret_cast<uint32_t *>(rowp)[k] = rw[k];
    }
    if (!dst || tu.dst_offset < 0)
        return;
    int z[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        z[2 * k] = (int)(int16_t)(rw[k] & 0xFFFF);
        z[2 * k + 1] = (int)(int16_t)(rw[k] >> 16);
    }
    ffhip_add_row<16>(dst + tu.dst_offset + (ptrdiff_t)rr * stride + 16 * rh * (bd > 8 ? 2 : 1), z, bd);
}

/*
 * k_hevc_idct16_mfma — two 16x16 units per wave on the same v_mfma_i32_32x32x32_i8: the operands are BLOCK-DIAGONAL (unit b in
 * rows / columns 16 b .. 16 b + 15), and so is K: of a lane's 16 K-slots, slots 8 b .. 8 b + 7 belong to unit b.  Lane (row c' = 16 b + c,
 * group g) therefore carries 8 values of ITS unit (pass 1: X_b[8 g + t][c], t = 0..7) and whatever in the other 8 slots — those meet
 * only B entries of the other unit's columns, i.e. they land in the off-diagonal quadrants of D, which nobody reads.  The D
 * registers a lane needs are r = 8 b .. 8 b + 7 (rows 16 b + (r & 3) + 8 ((r >> 2) & 1) + 4 g), and they ARE the A operand of pass 2
 * in place, with the second B table's rows permuted to that order — as in the 32x32 kernel: no LDS, no transposition.  Half of
 * the matrix core's work is thrown away; what is bought is the VALU: 64 v_dot2 per vector and a scalar load behind each.
 */
__global__ __launch_bounds__(256) void k_hevc_idct16_mfma(int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipHevcTU *tus, int n, int bd,
                                                          const HevcMfmaTab *tab)
{
    /* the matrix core wants a lane to hold a COLUMN of its unit and hands back a column; memory wants rows.  Both transpositions go
     * through 1 KB of LDS per wave: a lane moves 16 contiguous bytes of a coefficient row in and out and 8 picture samples, instead
     * of eight 2-byte accesses each way plus eight byte-wide read-modify-writes (measured: 1.56 -> 1.80 G units/s; a wave
     * looping over unit pairs with the next pair in flight was SLOWER than one pair per wave: 1.59) */
    __shared__ __align__(16) int16_t lds[4][2 * 256];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int u0 = (blockIdx.x * 4 + wave) * 2;
    if (u0 >= n)
        return;
    int16_t *L = lds[wave];
    /* row view of the wave's two units: lane = (unit rb, row rr, half rh): 8 samples */
    const int rb = lane >> 5, rr = (lane >> 1) & 15, rh = lane & 1;
    const bool rlive = u0 + rb < n;
    const FFHipHevcTU rtu = tus[rlive ? u0 + rb : u0];
    int16_t *rowp = coeffs + rtu.coeff_offset + rr * 16 + 8 * rh;
    const bool al16 = !(reinterpret_cast<uintptr_t>(rowp) & 15);
    {
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (rlive) {
            if (al16) {
                raw = *reinterpret_cast<const uint4 *>(rowp);
            } else {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(rowp);
                raw = make_uint4(p[0], p[1], p[2], p[3]);
            }
        }
        *reinterpret_cast<uint4 *>(L + rb * 256 + rr * 16 + 8 * rh) = raw;
    }
    hevc_wave_sync();
    /* column view: my unit, my column (pass 1) / row (pass 2) / column (output) */
    const int cp = lane & 31, g = lane >> 5, b = cp >> 4, c = cp & 15;
    const hm_i4 B1 = reinterpret_cast<const hm_i4 *>(tab->b1)[lane], B2 = reinterpret_cast<const hm_i4 *>(tab->b2)[lane];
    const int sum_t = tab->sum[cp];
    const int col_limit = __shfl(rtu.col_limit, 32 * b, 64); /* unit b's descriptor sits in the lanes of row-view unit b */
    /* ---- pass 1: column c of unit b, rows 8 g .. 8 g + 7 in slots 8 b .. 8 b + 7 ---- */
    int limit2 = min(col_limit + 4, 16);
    for (int q = 4; q < c; q += 4)
        if (limit2 < 16)
            limit2 -= 4;
    int v[16];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int k = 8 * g + t, x = L[b * 256 + k * 16 + c];
        const int val = ((k & 1) && k >= limit2) ? 0 : x; /* hevc_pass<16>'s rule: odd inputs beyond the limit are dropped */
        v[t] = b ? 0 : val;
        v[8 + t] = b ? val : 0;
    }
    hm_i4 ahi, alo;
    hm_split(v, ahi, alo);
    hm_i16 acc = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, B1, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++)
        acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)(sum_t + 64));
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, B1, acc, 0, 0, 0);
    /* ---- Y_b[j = c][cc], cc = (r & 3) + 8 ((r >> 2) & 1) + 4 g for r = 8 b .. 8 b + 7: >> 7, clip, the second pass's limit ---- */
    const int limit = min(col_limit, 16);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int cc = (r & 3) + 8 * ((r >> 2) & 1) + 4 * g;
        const int y = hevc_clip16(acc[r] >> 7);
        v[r] = ((r >> 3) == b && !((cc & 1) && cc >= limit)) ? y : 0;
    }
    hm_split(v, ahi, alo);
    const int shift2 = 20 - bd;
#pragma unroll
    for (int r = 0; r < 16; r++)
        acc[r] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, B2, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++)
        acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)(sum_t + (1 << (shift2 - 1))));
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, B2, acc, 0, 0, 0);
    /* ---- Z_b[j][m = c], j = (r & 3) + 8 ((r >> 2) & 1) + 4 g for r = 8 b .. 8 b + 7: back through LDS into rows ---- */
    hevc_wave_sync(); /* every lane has read its inputs */
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int j = (t & 3) + 8 * (t >> 2) + 4 * g;
        L[b * 256 + j * 16 + c] = (int16_t)hevc_clip16((b ? acc[8 + t] : acc[t]) >> shift2);
    }
    hevc_wave_sync();
    if (!rlive)
        return;
    const uint4 res = *reinterpret_cast<const uint4 *>(L + rb * 256 + rr * 16 + 8 * rh);
    if (al16) {
        *reinterpret_cast<uint4 *>(rowp) = res;
    } else {
        uint32_t *p = reinterpret_cast<uint32_t *>(rowp);
        p[0] = res.x; p[1] = res.y; p[2] = res.z; p[3] = res.w;
    }
    if (!dst || rtu.dst_offset < 0)
        return;
    const uint32_t rw[4] = { res.x, res.y, res.z, res.w };
    int z[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        z[2 * k] = (int)(int16_t)(rw[k] & 0xFFFF);
        z[2 * k + 1] = (int)(int16_t)(rw[k] >> 16);
    }
    ffhip_add_row<8>(dst + rtu.dst_offset + (ptrdiff_t)rr * stride + 8 * rh * (bd > 8 ? 2 : 1), z, bd);
}

/* the host tables are built once; their device copies (the __constant__ hevc_pk image and the two MFMA operand tables) exist per
 * device, uploaded when a device first runs a transform */
static HevcMfmaTab *g_hm_tab16_dev[64];
static HevcMfmaTab *g_hm_tab_dev[64];
static HevcMfmaTab hm_host, hm_host16;
static FFHipPerDeviceOnce hevc_pk_dev_once, hm_dev_once;

static int hevc_pk_upload()
{
    std::call_once(hevc_tab_once, [] { hevc_build_table(); });
    if (hevc_pk_dev_once.enter()) {
        const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(hevc_pk), hevc_pk_host, sizeof(hevc_pk_host));
        hevc_pk_dev_once.leave(e == hipSuccess);
        if (e != hipSuccess) {
            ffhip_set_error("ffhip_hevc_idct: coefficient table upload failed: %s", hipGetErrorString(e));
            return FFHIP_EIO;
        }
    }
    return 0;
}

static int hm_tab_init()
{
    const int r = hevc_pk_upload();
    if (r < 0)
        return r;
    std::call_once(hm_once, [] {
        HevcMfmaTab &h = hm_host;
        for (int l = 0; l < 64; l++) {
            const int j = l & 31, g = l >> 5;
            for (int s = 0; s < 16; s++) {
                h.b1[l][s] = hevc_t32_host[16 * g + s][j];                              /* T[k = 16 g + s][j]               */
                h.b2[l][s] = hevc_t32_host[(s & 3) + 8 * (s >> 2) + 4 * g][j];        /* T[c = the D row of register s][m] */
            }
        }
        for (int j = 0; j < 32; j++) {
            int t = 0;
            for (int k = 0; k < 32; k++)
                t += hevc_t32_host[k][j];
            h.sum[j] = 128 * t;
        }
        /* the block-diagonal pair of 16-point matrices (T16[k][j] = T32[2 k][j]): slot s of group g belongs to unit s >> 3 and is
         * row 8 g + (s & 7) of T16 in pass 1, row (s & 3) + 8 ((s >> 2) & 1) + 4 g in pass 2 */
        HevcMfmaTab &h16 = hm_host16;
        for (int l = 0; l < 64; l++) {
            const int jp = l & 31, g = l >> 5, j = jp & 15;
            for (int sl = 0; sl < 16; sl++) {
                const bool mine = (sl >> 3) == (jp >> 4);
                h16.b1[l][sl] = mine ? hevc_t32_host[2 * (8 * g + (sl & 7))][j] : 0;
                h16.b2[l][sl] = mine ? hevc_t32_host[2 * ((sl & 3) + 8 * ((sl >> 2) & 1) + 4 * g)][j] : 0;
            }
        }
        for (int jp = 0; jp < 32; jp++) {
            int t = 0;
            for (int k = 0; k < 16; k++)
                t += hevc_t32_host[2 * k][jp & 15];
            h16.sum[jp] = 128 * t;
        }
    });
    if (hm_dev_once.enter()) {
        const int d = ffhip_current_device();
        HevcMfmaTab *t32 = nullptr, *t16 = nullptr;
        const bool ok = hipMalloc(reinterpret_cast<void **>(&t32), sizeof(HevcMfmaTab)) == hipSuccess &&
                        hipMalloc(reinterpret_cast<void **>(&t16), sizeof(HevcMfmaTab)) == hipSuccess &&
                        hipMemcpy(t32, &hm_host, sizeof(HevcMfmaTab), hipMemcpyHostToDevice) == hipSuccess &&
                        hipMemcpy(t16, &hm_host16, sizeof(HevcMfmaTab), hipMemcpyHostToDevice) == hipSuccess;
        if (ok) {
            g_hm_tab_dev[d] = t32;
            g_hm_tab16_dev[d] = t16;
        } else {
            (void)hipFree(t32);
            (void)hipFree(t16);
        }
        hm_dev_once.leave(ok);
        if (!ok) {
            ffhip_set_error("ffhip_hevc_idct: table upload failed");
            return FFHIP_EIO;
        }
    }
    return 0;
}

int ffhip_launch_hevc_idct(int kind, int log2_size, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipHevcTU *tus, int n,
                           hipStream_t stream)
{
    return ffhip_launch_hevc_idct_bd(8, kind, log2_size, coeffs, dst, stride, tus, n, stream);
}

int ffhip_launch_hevc_idct_bd(int bd, int kind, int log2_size, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipHevcTU *tus, int n,
                              hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (bd != 8 && bd != 10 && bd != 12) {
        ffhip_set_error("ffhip_hevc: bit depth %d (8, 10 and 12 are built)", bd);
        return FFHIP_EINVAL;
    }
    {
        const int r = hevc_pk_upload();
        if (r < 0)
            return r;
    }
    {
        const char *eo = FFHIP_KNOB("FFHIP_HEVC_IDCT32_VALU"); /* measured variant: the dot2 kernel for 32x32 as well */
        if (kind == FFHIP_HEVC_IDCT && log2_size == 5 && !(eo && eo[0] == '1')) {
            const int r = hm_tab_init();
            if (r < 0)
                return r;
            hipLaunchKernelGGL(k_hevc_idct32_mfma, dim3(cdiv(n, 4)), dim3(256), 0, stream, coeffs, dst, stride, tus, n, bd, g_hm_tab_dev[ffhip_current_device()]);
            LAUNCH_CHECK();
            return 0;
        }
        /* 16x16: with 16-byte staging the dot2 kernel (0.55 of HBM) beats the two-units-per-MFMA kernel (0.34, half of whose matrix
         * work is discarded and whose columns move through LDS 2 bytes at a time); the latter stays as a measured variant */
        const char *e16 = FFHIP_KNOB("FFHIP_HEVC_IDCT16_MFMA");
        if (kind == FFHIP_HEVC_IDCT && log2_size == 4 && e16 && e16[0] == '1') {
            const int r = hm_tab_init();
            if (r < 0)
                return r;
            hipLaunchKernelGGL(k_hevc_idct16_mfma, dim3(cdiv(n, 8)), dim3(256), 0, stream, coeffs, dst, stride, tus, n, bd, g_hm_tab16_dev[ffhip_current_device()]);
            LAUNCH_CHECK();
            return 0;
        }
    }
    const int upw = 64 >> log2_size;
    const dim3 grid(cdiv(n, 4 * upw)), block(256);
    switch (log2_size) {
    case 2: hipLaunchKernelGGL(k_hevc_idct<2>, grid, block, 0, stream, kind, coeffs, dst, stride, tus, n, bd); break;
    case 3: hipLaunchKernelGGL(k_hevc_idct<3>, grid, block, 0, stream, kind, coeffs, dst, stride, tus, n, bd); break;
    case 4: hipLaunchKernelGGL(k_hevc_idct<4>, grid, block, 0, stream, kind, coeffs, dst, stride, tus, n, bd); break;
    case 5: hipLaunchKernelGGL(k_hevc_idct<5>, grid, block, 0, stream, kind, coeffs, dst, stride, tus, n, bd); break;
    default:
        ffhip_set_error("ffhip_hevc_idct: log2_size %d outside 2..5", log2_size);
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}

/* ================================================================================================== */
/*
 * HEVC deblocking, 8-bit: hevc_{h,v}_loop_filter_{luma,chroma} (libavcodec/hevc/dsp_template.c:834-929) with the
 * strong / weak / chroma filters of libavcodec/h26x/h2656_deblock_template.c:25-104, batched over edge segments whose
 * pixels are disjoint (all vertical edges of a picture, then all horizontal ones: HEVC has no order inside a direction).
 * 8 lanes per segment, one per sample line; the two 4-line groups decide from their lines 0 and 3, which the lanes of a
 * group exchange with DPP-style shuffles.  Every read of a line happens before any write of it.
 */
__device__ __forceinline__ int hv_abs(int v) { return v < 0 ? -v : v; }

/* PIX = uint8_t (bd 8) / uint16_t: beta and tc arrive in 8-bit units and are scaled as the reference's templates scale them
 * (beta <<= BIT_DEPTH - 8, tc = _tc[j] << (BIT_DEPTH - 8): hevc/dsp_template.c:845,862,907); stride and offsets in bytes */
template <typename PIX>
__device__ __forceinline__ void hevc_lf_lines(uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, int e, int lane_, int bd)
{
    constexpr int PS = (int)sizeof(PIX);
    const int maxv = (1 << bd) - 1;
    auto clipp = [&](int v) { return min(max(v, 0), maxv); };
    const int line = lane_ & 7, j = line >> 2;
    const bool live = e < n;
    const FFHipHevcEdge ed = edges[live ? e : 0];
    const bool vertical = ed.kind & 1, chroma = ed.kind & 2;
    const ptrdiff_t st = stride / (ptrdiff_t)sizeof(PIX), xs = vertical ? 1 : st, ys = vertical ? st : 1;
    PIX *pix = reinterpret_cast<PIX *>(base + ed.offset) + (ptrdiff_t)line * ys;
    const int tc = ed.tc[j] << (bd - 8), no_p = ed.no_p[j], no_q = ed.no_q[j], beta = ed.beta << (bd - 8);
    /* a vertical edge's line is 8 contiguous samples p3 .. q3: two dwords (two 8-byte words at 16 bits) when p3 is so aligned */
    const bool wide = live && vertical && !chroma && !(reinterpret_cast<uintptr_t>(pix - 4) & (PS == 1 ? 3 : 7));
    int v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; /* p3 p2 p1 p0 q0 q1 q2 q3 */
    if (wide) {
        if constexpr (PS == 1) {
            const uint32_t a = reinterpret_cast<const uint32_t *>(pix - 4)[0], c = reinterpret_cast<const uint32_t *>(pix - 4)[1];
            v[0] = a & 255; v[1] = (a >> 8) & 255; v[2] = (a >> 16) & 255; v[3] = a >> 24;
            v[4] = c & 255; v[5] = (c >> 8) & 255; v[6] = (c >> 16) & 255; v[7] = c >> 24;
        } else {
            const uint2 a = reinterpret_cast<const uint2 *>(pix - 4)[0], c = reinterpret_cast<const uint2 *>(pix - 4)[1];
            v[0] = a.x & 0xFFFF; v[1] = a.x >> 16; v[2] = a.y & 0xFFFF; v[3] = a.y >> 16;
            v[4] = c.x & 0xFFFF; v[5] = c.x >> 16; v[6] = c.y & 0xFFFF; v[7] = c.y >> 16;
        }
    } else if (live) {
        v[2] = pix[-2 * xs]; v[3] = pix[-xs]; v[4] = pix[0]; v[5] = pix[xs];
        if (!chroma) {
            v[0] = pix[-4 * xs]; v[1] = pix[-3 * xs]; v[6] = pix[2 * xs]; v[7] = pix[3 * xs];
        }
    }
    const int p3 = v[0], p2 = v[1], p1 = v[2], p0 = v[3], q0 = v[4], q1 = v[5], q2 = v[6], q3 = v[7];
    if (chroma) {
        if (live && tc > 0) {
            const int delta = clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
            if (!no_p) pix[-xs] = (PIX)clipp(p0 + delta);
            if (!no_q) pix[0] = (PIX)clipp(q0 - delta);
        }
        return;
    }
    /* decisions of my group from its lines 0 and 3 (all 64 lanes take part in the shuffles) */
    const int dp = hv_abs(p2 - 2 * p1 + p0), dq = hv_abs(q2 - 2 * q1 + q0);
    const int flat = hv_abs(p3 - p0) + hv_abs(q3 - q0), step = hv_abs(p0 - q0);
    const int l0 = (lane_ & 63) & ~3, l3 = l0 + 3;
    const int dp0 = __shfl(dp, l0, 64), dp3 = __shfl(dp, l3, 64), dq0 = __shfl(dq, l0, 64), dq3 = __shfl(dq, l3, 64);
    const int flat0 = __shfl(flat, l0, 64), flat3 = __shfl(flat, l3, 64), step0 = __shfl(step, l0, 64), step3 = __shfl(step, l3, 64);
    if (!live)
        return;
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta)
        return;
    unsigned ch = 0; /* bit k: v[k] changed */
    const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
    if (flat0 < beta_3 && step0 < tc25 && flat3 < beta_3 && step3 < tc25 && (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
        const int t = tc << 1;
        if (!no_p) {
            v[3] = p0 + clip3(((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0, -t, t);
            v[2] = p1 + clip3(((p2 + p1 + p0 + q0 + 2) >> 2) - p1, -t, t);
            v[1] = p2 + clip3(((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2, -t, t);
            ch |= 0x0E;
        }
        if (!no_q) {
            v[4] = q0 + clip3(((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0, -t, t);
            v[5] = q1 + clip3(((p0 + q0 + q1 + q2 + 2) >> 2) - q1, -t, t);
            v[6] = q2 + clip3(((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2, -t, t);
            ch |= 0x70;
        }
    } else {
        const int side = (beta + (beta >> 1)) >> 3;
        const int nd_p = dp0 + dp3 < side ? 2 : 1, nd_q = dq0 + dq3 < side ? 2 : 1, tc_2 = tc >> 1;
        int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
        if (hv_abs(delta) < 10 * tc) {
            delta = clip3(delta, -tc, tc);
            if (!no_p) { v[3] = clipp(p0 + delta); ch |= 0x08; }
            if (!no_q) { v[4] = clipp(q0 - delta); ch |= 0x10; }
            if (!no_p && nd_p > 1) { v[2] = clipp(p1 + clip3((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc_2, tc_2)); ch |= 0x04; }
            if (!no_q && nd_q > 1) { v[5] = clipp(q1 + clip3((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc_2, tc_2)); ch |= 0x20; }
        }
    }
    if (!ch)
        return;
    if (wide) {
        if constexpr (PS == 1) {
            if (ch & 0x0F)
                reinterpret_cast<uint32_t *>(pix - 4)[0] = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
            if (ch & 0xF0)
                reinterpret_cast<uint32_t *>(pix - 4)[1] = (uint32_t)v[4] | (uint32_t)v[5] << 8 | (uint32_t)v[6] << 16 | (uint32_t)v[7] << 24;
        } else {
            if (ch & 0x0F)
                reinterpret_cast<uint2 *>(pix - 4)[0] = make_uint2((uint32_t)v[0] | (uint32_t)v[1] << 16, (uint32_t)v[2] | (uint32_t)v[3] << 16);
            if (ch & 0xF0)
                reinterpret_cast<uint2 *>(pix - 4)[1] = make_uint2((uint32_t)v[4] | (uint32_t)v[5] << 16, (uint32_t)v[6] | (uint32_t)v[7] << 16);
        }
    } else {
#pragma unroll
        for (int k = 1; k < 7; k++)
            if (ch >> k & 1)
                pix[(k - 4) * xs] = (PIX)v[k];
    }
}

/*
 * k_hevc_loop_filter_h — HORIZONTAL edges (the filter runs down a column, a segment's 8 lines lie side by side in a row) with a lane
 * per 4-line group: the group's columns are one dword (8 bytes at 16 bits) of each row, so a wave moves 32 segments with dword
 * accesses instead of 8 segments byte by byte, and a group decides from its own lines 0 and 3 without a shuffle.  The sample-per-
 * lane kernel above stays for vertical edges and unaligned planes.  One launch serves a batch: each wave looks at its 32 records
 * and takes this path only when all of them are horizontal (a picture's batch is all vertical, then all horizontal).
 */
template <typename PIX>
__device__ __forceinline__ void hevc_lf_hgroup(uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge &ed, int j, int bd)
{
    constexpr int PS = (int)sizeof(PIX);
    using ROW = typename std::conditional<PS == 1, uint32_t, uint2>::type;
    const bool chroma = ed.kind & 2;
    const int maxv = (1 << bd) - 1;
    auto clipp = [&](int v) { return min(max(v, 0), maxv); };
    const int tc = ed.tc[j] << (bd - 8), no_p = ed.no_p[j], no_q = ed.no_q[j], beta = ed.beta << (bd - 8);
    uint8_t *pix = base + ed.offset + 4 * j * PS; /* the group's first column on row q0 */
    auto ld = [&](int r, int (&v)[4]) {
        const ROW w = *reinterpret_cast<const ROW *>(pix + (ptrdiff_t)r * stride);
        if constexpr (PS == 1) {
            v[0] = w & 255; v[1] = (w >> 8) & 255; v[2] = (w >> 16) & 255; v[3] = w >> 24;
        } else {
            v[0] = w.x & 0xFFFF; v[1] = w.x >> 16; v[2] = w.y & 0xFFFF; v[3] = w.y >> 16;
        }
    };
    auto st = [&](int r, const int (&v)[4]) {
        ROW w;
        if constexpr (PS == 1)
            w = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
        else
            w = make_uint2((uint32_t)v[0] | (uint32_t)v[1] << 16, (uint32_t)v[2] | (uint32_t)v[3] << 16);
        *reinterpret_cast<ROW *>(pix + (ptrdiff_t)r * stride) = w;
    };
    int p1[4], p0[4], q0[4], q1[4];
    ld(-2, p1); ld(-1, p0); ld(0, q0); ld(1, q1);
    if (chroma) {
        if (tc <= 0)
            return;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int delta = clip3((((q0[c] - p0[c]) * 4) + p1[c] - q1[c] + 4) >> 3, -tc, tc);
            if (!no_p) p0[c] = clipp(p0[c] + delta);
            if (!no_q) q0[c] = clipp(q0[c] - delta);
        }
        if (!no_p) st(-1, p0);
        if (!no_q) st(0, q0);
        return;
    }
    int p3[4], p2[4], q2[4], q3[4];
    ld(-4, p3); ld(-3, p2); ld(2, q2); ld(3, q3);
    auto dpf = [&](int c) { return hv_abs(p2[c] - 2 * p1[c] + p0[c]); };
    auto dqf = [&](int c) { return hv_abs(q2[c] - 2 * q1[c] + q0[c]); };
    const int dp0 = dpf(0), dp3 = dpf(3), dq0 = dqf(0), dq3 = dqf(3);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta)
        return;
    const int flat0 = hv_abs(p3[0] - p0[0]) + hv_abs(q3[0] - q0[0]), flat3 = hv_abs(p3[3] - p0[3]) + hv_abs(q3[3] - q0[3]);
    const int step0 = hv_abs(p0[0] - q0[0]), step3 = hv_abs(p0[3] - q0[3]);
    const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
    if (flat0 < beta_3 && step0 < tc25 && flat3 < beta_3 && step3 < tc25 && (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
        const int t = tc << 1;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int P3 = p3[c], P2 = p2[c], P1 = p1[c], P0 = p0[c], Q0 = q0[c], Q1 = q1[c], Q2 = q2[c], Q3 = q3[c];
            if (!no_p) {
                p0[c] = P0 + clip3(((P2 + 2 * P1 + 2 * P0 + 2 * Q0 + Q1 + 4) >> 3) - P0, -t, t);
                p1[c] = P1 + clip3(((P2 + P1 + P0 + Q0 + 2) >> 2) - P1, -t, t);
                p2[c] = P2 + clip3(((2 * P3 + 3 * P2 + P1 + P0 + Q0 + 4) >> 3) - P2, -t, t);
            }
            if (!no_q) {
                q0[c] = Q0 + clip3(((P1 + 2 * P0 + 2 * Q0 + 2 * Q1 + Q2 + 4) >> 3) - Q0, -t, t);
                q1[c] = Q1 + clip3(((P0 + Q0 + Q1 + Q2 + 2) >> 2) - Q1, -t, t);
                q2[c] = Q2 + clip3(((2 * Q3 + 3 * Q2 + Q1 + Q0 + P0 + 4) >> 3) - Q2, -t, t);
            }
        }
        if (!no_p) { st(-1, p0); st(-2, p1); st(-3, p2); }
        if (!no_q) { st(0, q0); st(1, q1); st(2, q2); }
    } else {
        const int side = (beta + (beta >> 1)) >> 3;
        const int nd_p = dp0 + dp3 < side ? 2 : 1, nd_q = dq0 + dq3 < side ? 2 : 1, tc_2 = tc >> 1;
        bool any = false;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int P2 = p2[c], P1 = p1[c], P0 = p0[c], Q0 = q0[c], Q1 = q1[c], Q2 = q2[c];
            int delta = (9 * (Q0 - P0) - 3 * (Q1 - P1) + 8) >> 4;
            if (hv_abs(delta) < 10 * tc) {
                any = true;
                delta = clip3(delta, -tc, tc);
                if (!no_p) p0[c] = clipp(P0 + delta);
                if (!no_q) q0[c] = clipp(Q0 - delta);
                if (!no_p && nd_p > 1)
                    p1[c] = clipp(P1 + clip3((((P2 + P0 + 1) >> 1) - P1 + delta) >> 1, -tc_2, tc_2));
                if (!no_q && nd_q > 1)
                    q1[c] = clipp(Q1 + clip3((((Q2 + Q0 + 1) >> 1) - Q1 - delta) >> 1, -tc_2, tc_2));
            }
        }
        if (any) {
            if (!no_p) { st(-1, p0); if (nd_p > 1) st(-2, p1); }
            if (!no_q) { st(0, q0); if (nd_q > 1) st(1, q1); }
        }
    }
}

template <typename PIX>
__global__ __launch_bounds__(256) void k_hevc_loop_filter_w(uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, int bd)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6)), lane = threadIdx.x & 63;
    const int e0 = wave * 32;
    if (e0 >= n)
        return;
    /* horizontal everywhere in my 32 records (and rows that take dword / 8-byte accesses)? */
    const int e = e0 + (lane >> 1);
    const bool live = e < n;
    FFHipHevcEdge ed = {};
    if (live)
        ed = edges[e];
    const bool ok = !live || (!(ed.kind & 1) && !((reinterpret_cast<uintptr_t>(base) + (size_t)ed.offset) & (sizeof(PIX) == 1 ? 3 : 7)));
    if (__builtin_amdgcn_read_exec() == __ballot(ok) && !((size_t)stride & (sizeof(PIX) == 1 ? 3 : 7))) {
        if (live)
            hevc_lf_hgroup<PIX>(base, stride, ed, lane & 1, bd);
        return;
    }
    /* otherwise: the sample-per-lane rules, 8 records at a time */
    for (int r = 0; r < 4; r++)
        hevc_lf_lines<PIX>(base, stride, edges, n, e0 + 8 * r + (lane >> 3), lane, bd);
}

int ffhip_launch_hevc_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, hipStream_t stream)
{
    return ffhip_launch_hevc_loop_filter_bd(8, base, stride, edges, n, stream);
}

int ffhip_launch_hevc_loop_filter_bd(int bd, uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (bd == 8)
        hipLaunchKernelGGL(k_hevc_loop_filter_w<uint8_t>, dim3(cdiv(n, 128)), dim3(256), 0, stream, base, stride, edges, n, 8);
    else if ((bd == 10 || bd == 12) && !(((uintptr_t)base | (size_t)stride) & 1))
        hipLaunchKernelGGL(k_hevc_loop_filter_w<uint16_t>, dim3(cdiv(n, 128)), dim3(256), 0, stream, base, stride, edges, n, bd);
    else {
        ffhip_set_error("ffhip_hevc_loop_filter: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}
static_assert(sizeof(FFHipHevcEdge) == 16, "FFHipHevcEdge is a 16-byte record");
static_assert(sizeof(FFHipHevcTU) == 12, "FFHipHevcTU is a 12-byte record");

/* ================================================================================================== */
/*
 * HEVC sample adaptive offset, 8-bit: sao_band_filter / sao_edge_filter (libavcodec/h26x/h2656_sao_template.c:24-84),
 * batched: one wave per block, a lane per 4 horizontally adjacent samples = one dword.  Band: offset by the sample's 5-bit band
 * when it is one of the 4 signalled ones.  Edge: sign(c - a) + sign(c - b) against the two neighbours of the class's direction
 * selects one of 5 offsets.  Pure streaming (2 B per sample), so the arithmetic is packed: the class index of each of the four
 * samples becomes a selector byte and ONE v_perm_b32 looks the four offsets up in an 8-entry byte table (offsets stored + 128;
 * the sample's band / sign sums are computed on even and odd bytes as 2 x 16-bit lanes), the add and the clip run as packed
 * 16-bit.  Offsets outside int8 (impossible at 8 bits, but the signature is int16) and partial groups take the bytewise path.
 */
static_assert(sizeof(FFHipHevcSao) == 24, "FFHipHevcSao is a 24-byte record");
__constant__ uint32_t hevc_inv16[17] = { 0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4096 };

/* the packed 16-bit min / max / saturating subtract, spelled out: clang scalarises the generic vector builtins into v_cmp + v_cndmask */
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_subs_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
/* clamp(c - a, -1, 1) + 1 on both 16-bit lanes: 0..2, so that sums of them are plain dword adds (no negative lane to carry) */
__device__ __forceinline__ uint32_t sao_sign1(uint32_t c1, uint32_t a) /* c1 = c + 0x00010001 */
{
    return pk_min_i16(pk_max_i16(pk_sub_i16(c1, a), 0u), 0x00020002u);
}
/* clip_u8(c + ob - 128) on both 16-bit lanes (c + ob < 2^16: a plain add is the packed add) */
__device__ __forceinline__ uint32_t sao_add(uint32_t c, uint32_t ob)
{
    return pk_min_i16(pk_subs_u16(c + ob, 0x00800080u), 0x00ff00ffu);
}

__global__ __launch_bounds__(256) void k_hevc_sao(uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSao *blocks, int n)
{
    const int b = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (b >= n)
        return;
    const FFHipHevcSao k = blocks[b];
    const int w = __builtin_amdgcn_readfirstlane((int)k.width), h = __builtin_amdgcn_readfirstlane((int)k.height), qw = (w + 3) >> 2;
    const bool edge = __builtin_amdgcn_readfirstlane((int)k.edge) != 0;
    const int cls = __builtin_amdgcn_readfirstlane((int)k.cls);
    const uint8_t *s0 = src + __builtin_amdgcn_readfirstlane(k.src_offset);
    uint8_t *d0 = dst + __builtin_amdgcn_readfirstlane(k.dst_offset);
    int o[5];
#pragma unroll
    for (int i = 0; i < 5; i++)
        o[i] = __builtin_amdgcn_readfirstlane((int)k.offset_val[i]);
    const int o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3], o4 = o[4];
    static const int8_t dxs[4][2] = { { -1, 1 }, { 0, 0 }, { -1, 1 }, { 1, -1 } }, dys[4][2] = { { 0, 0 }, { -1, 1 }, { -1, 1 }, { -1, 1 } };
    const int eo = cls & 3;
    const ptrdiff_t a = dxs[eo][0] + dys[eo][0] * ss, bb = dxs[eo][1] + dys[eo][1] * ss;
    const int inv = (int)hevc_inv16[min(qw, 16)];

    bool packed = true;
#pragma unroll
    for (int i = 0; i < 5; i++)
        packed = packed && o[i] >= -128 && o[i] <= 127;
    /* the offset tables, + 128: edge index = sign sum + 2 -> (o1, o2, o0, o3 | o4); band index = relative band -> (o1..o4 | 0 x 4) */
    const auto ob = [](int v) { return (uint32_t)((v + 128) & 255); };
    const uint32_t tlo = edge ? ob(o1) | ob(o2) << 8 | ob(o0) << 16 | ob(o3) << 24 : ob(o1) | ob(o2) << 8 | ob(o3) << 16 | ob(o4) << 24;
    const uint32_t thi = edge ? ob(o4) | 0x80808000u : 0x80808080u;
    const bool d_al = ((reinterpret_cast<uintptr_t>(d0) | (uintptr_t)sd) & 3) == 0;
    const uint32_t rel = (uint32_t)((32 - cls) & 31) * 0x01010101u;

    for (int t = lane; t < qw * h; t += 64) {
        const int y = (t * inv) >> 16, x0 = 4 * (t - y * qw);
        const uint8_t *p = s0 + (ptrdiff_t)y * ss + x0;
        uint8_t *q = d0 + (ptrdiff_t)y * sd + x0;
        const int m = min(4, w - x0);
        if (packed && m == 4) {
            const uint32_t c = *reinterpret_cast<const uint32_t *>(p);
            const uint32_t ce = c & 0x00ff00ffu, co = (c >> 8) & 0x00ff00ffu;
            uint32_t sel;
            if (edge) {
                const uint32_t na = *reinterpret_cast<const uint32_t *>(p + a), nb = *reinterpret_cast<const uint32_t *>(p + bb);
                /* sign sums + 2 = 0..4 in each 16-bit lane */
                const uint32_t ce1 = ce + 0x00010001u, co1 = co + 0x00010001u;
                const uint32_t se = sao_sign1(ce1, na & 0x00ff00ffu) + sao_sign1(ce1, nb & 0x00ff00ffu);
                const uint32_t so = sao_sign1(co1, (na >> 8) & 0x00ff00ffu) + sao_sign1(co1, (nb >> 8) & 0x00ff00ffu);
                sel = se | so << 8;
            } else {
                const uint32_t band = (((c >> 3) & 0x1f1f1f1fu) + rel) & 0x1f1f1f1fu;       /* per byte: (band - left_class) & 31 */
                const uint32_t far = ((band & 0x1c1c1c1cu) + 0x7f7f7f7fu) & 0x80808080u;   /* 0x80 where it is not one of 0..3 */
                sel = (band & 0x03030303u) | far >> 5;                                       /* those read entries 4..7 = no offset */
            }
            const uint32_t of = __builtin_amdgcn_perm(thi, tlo, sel);
            const uint32_t re = sao_add(ce, of & 0x00ff00ffu), ro = sao_add(co, (of >> 8) & 0x00ff00ffu);
            const uint32_t out = re | ro << 8;
            if (d_al) {
                *reinterpret_cast<uint32_t *>(q) = out;
            } else {
                q[0] = (uint8_t)out; q[1] = (uint8_t)(out >> 8); q[2] = (uint8_t)(out >> 16); q[3] = (uint8_t)(out >> 24);
            }
            continue;
        }
        for (int e = 0; e < m; e++) {
            const int c = p[e];
            int off;
            if (edge) {
                const int na = p[e + a], nb = p[e + bb];
                const int sel = 2 + (c > na) - (c < na) + (c > nb) - (c < nb); /* edge_idx = { 1, 2, 0, 3, 4 } */
                off = sel == 0 ? o1 : sel == 1 ? o2 : sel == 2 ? o0 : sel == 3 ? o3 : o4;
            } else {
                const int band = ((c >> 3) - cls) & 31;                          /* 0..3: the signalled bands */
                off = band == 0 ? o1 : band == 1 ? o2 : band == 2 ? o3 : band == 3 ? o4 : 0;
            }
            q[e] = (uint8_t)clip_u8(c + off);
        }
    }
}

/* the same two filters on uint16_t samples (bd 10 / 12): band = (sample >> (bd - 5)) & 31 (h2656_sao_template.c:33), a lane per
 * two samples = one dword where the rows allow, offsets as they arrive (the decoder has scaled them by << (bd - min(bd, 10))) */
__global__ __launch_bounds__(256) void k_hevc_sao16(uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSao *blocks, int n,
                                                    int bd)
{
    const int b = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (b >= n)
        return;
    const FFHipHevcSao k = blocks[b];
    const int w = k.width, h = k.height, cls = k.cls, eo = cls & 3, maxv = (1 << bd) - 1, bshift = bd - 5;
    const bool edge = k.edge != 0;
    static const int8_t dxs[4][2] = { { -1, 1 }, { 0, 0 }, { -1, 1 }, { 1, -1 } }, dys[4][2] = { { 0, 0 }, { -1, 1 }, { -1, 1 }, { -1, 1 } };
    const ptrdiff_t sp = ss / 2, a = dxs[eo][0] + dys[eo][0] * sp, bb = dxs[eo][1] + dys[eo][1] * sp;
    const uint16_t *s0 = reinterpret_cast<const uint16_t *>(src + k.src_offset);
    uint8_t *d0 = dst + k.dst_offset;
    const int o0 = k.offset_val[0], o1 = k.offset_val[1], o2 = k.offset_val[2], o3 = k.offset_val[3], o4 = k.offset_val[4];
    for (int t = lane; t < w * h; t += 64) {
        const int y = t / w, x = t - y * w;
        const uint16_t *p = s0 + (ptrdiff_t)y * sp + x;
        const int c = p[0];
        int off;
        if (edge) {
            const int na = p[a], nb = p[bb];
            const int sel = 2 + (c > na) - (c < na) + (c > nb) - (c < nb); /* edge_idx = { 1, 2, 0, 3, 4 } */
            off = sel == 0 ? o1 : sel == 1 ? o2 : sel == 2 ? o0 : sel == 3 ? o3 : o4;
        } else {
            const int band = (((c >> bshift) & 31) - cls) & 31;           /* 0..3: the signalled bands */
            off = band == 0 ? o1 : band == 1 ? o2 : band == 2 ? o3 : band == 3 ? o4 : 0;
        }
        reinterpret_cast<uint16_t *>(d0 + (ptrdiff_t)y * sd)[x] = (uint16_t)min(max(c + off, 0), maxv);
    }
}

int ffhip_launch_hevc_sao_bd(int bd, uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSao *blocks, int n,
                             hipStream_t stream)
{
    if (bd == 8)
        return ffhip_launch_hevc_sao(dst, sd, src, ss, blocks, n, stream);
    if (n <= 0)
        return 0;
    if ((bd != 10 && bd != 12) || (((uintptr_t)dst | (uintptr_t)src | (size_t)sd | (size_t)ss) & 1)) {
        ffhip_set_error("ffhip_hevc_sao: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    hipLaunchKernelGGL(k_hevc_sao16, dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, sd, src, ss, blocks, n, bd);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_hevc_sao(uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSao *blocks, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_hevc_sao, dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, sd, src, ss, blocks, n);
    LAUNCH_CHECK();
    return 0;
}

/*
 * sao_edge_restore[0] / [1] (libavcodec/h26x/h2656_sao_template.c:81-214): after the edge filter, samples on a picture border get
 * the plain offset sao_offset_val[0] and (variant 1) samples next to slices / tiles that must not be filtered across are put
 * back.  The reference is a sequence of short loops whose later writes win; here every candidate sample — the columns 0, w-2,
 * w-1 and the rows 0, h-2, h-1 of the block — evaluates that sequence for itself and keeps the last writer, so the launch has
 * no ordering inside.  One wave per block.
 */
static_assert(sizeof(FFHipHevcSaoRestore) == 20, "FFHipHevcSaoRestore is a 20-byte record");
template <typename PIX>
__global__ __launch_bounds__(256) void k_hevc_sao_restore(uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss,
                                                          const FFHipHevcSaoRestore *blocks, int n, int bd)
{
    const int maxv = (1 << bd) - 1;
    const int b = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (b >= n)
        return;
    const FFHipHevcSaoRestore k = blocks[b];
    const int W = k.width, H = k.height, eo = k.eo & 3, off = k.offset0;
    const bool b0 = k.borders & 1, b1 = k.borders & 2, b2 = k.borders & 4, b3 = k.borders & 8;
    const bool ve0 = k.vert_edge & 1, ve1 = k.vert_edge & 2, he0 = k.horiz_edge & 1, he1 = k.horiz_edge & 2;
    const bool de0 = k.diag_edge & 1, de1 = k.diag_edge & 2, de2 = k.diag_edge & 4, de3 = k.diag_edge & 8;
    enum { HORIZ = 0, VERT = 1, D135 = 2, D45 = 3 };
    /* the running state of the reference after its border loops */
    const int init_x = (eo != VERT && b0) ? 1 : 0, w = W - ((eo != VERT && b2) ? 1 : 0);
    const int init_y = (eo != HORIZ && b1) ? 1 : 0, h = H - ((eo != HORIZ && b3) ? 1 : 0);
    const int s_ul = !de0 && eo == D135 && !b0 && !b1, s_ur = !de1 && eo == D45 && !b1 && !b2;
    const int s_lr = !de2 && eo == D135 && !b2 && !b3, s_ll = !de3 && eo == D45 && !b0 && !b3;
    const uint8_t *s0 = src + k.src_offset;
    uint8_t *d0 = dst + k.dst_offset;
    for (int t = lane; t < 3 * H + 3 * W; t += 64) {
        int x, y;
        if (t < 3 * H) {
            const int c = t / H;
            y = t - c * H;
            x = c == 0 ? 0 : c == 1 ? W - 1 : W - 2;
        } else {
            const int r = (t - 3 * H) / W;
            x = t - 3 * H - r * W;
            y = r == 0 ? 0 : r == 1 ? H - 1 : H - 2;
        }
        if (x < 0 || y < 0)
            continue;
        int kind = 0; /* 1: src + offset, 2: src */
        if (eo != VERT) {
            if (b0 && x == 0) kind = 1;
            if (b2 && x == W - 1) kind = 1;
        }
        if (eo != HORIZ) {
            if (b1 && y == 0 && x >= init_x && x < w) kind = 1;
            if (b3 && y == H - 1 && x >= init_x && x < w) kind = 1;
        }
        if (k.variant) {
            if (ve0 && eo != VERT && x == 0 && y >= init_y + s_ul && y < h - s_ll) kind = 2;
            if (ve1 && eo != VERT && x == w - 1 && y >= init_y + s_ur && y < h - s_lr) kind = 2;
            if (he0 && eo != HORIZ && y == 0 && x >= init_x + s_ul && x < w - s_ur) kind = 2;
            if (he1 && eo != HORIZ && y == h - 1 && x >= init_x + s_ll && x < w - s_lr) kind = 2;
            if (de0 && eo == D135 && x == 0 && y == 0) kind = 2;
            if (de1 && eo == D45 && x == w - 1 && y == 0) kind = 2;
            if (de2 && eo == D135 && x == w - 1 && y == h - 1) kind = 2;
            if (de3 && eo == D45 && x == 0 && y == h - 1) kind = 2;
        }
        if (kind) {
            const int v = reinterpret_cast<const PIX *>(s0 + (ptrdiff_t)y * ss)[x];
            reinterpret_cast<PIX *>(d0 + (ptrdiff_t)y * sd)[x] = (PIX)(kind == 1 ? min(max(v + off, 0), maxv) : v);
        }
    }
}

int ffhip_launch_hevc_sao_restore(uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSaoRestore *blocks, int n,
                                  hipStream_t stream)
{
    return ffhip_launch_hevc_sao_restore_bd(8, dst, sd, src, ss, blocks, n, stream);
}

int ffhip_launch_hevc_sao_restore_bd(int bd, uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSaoRestore *blocks,
                                     int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (bd == 8)
        hipLaunchKernelGGL(k_hevc_sao_restore<uint8_t>, dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, sd, src, ss, blocks, n, 8);
    else if ((bd == 10 || bd == 12) && !(((uintptr_t)dst | (uintptr_t)src | (size_t)sd | (size_t)ss) & 1))
        hipLaunchKernelGGL(k_hevc_sao_restore<uint16_t>, dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, sd, src, ss, blocks, n, bd);
    else {
        ffhip_set_error("ffhip_hevc_sao_restore: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}
