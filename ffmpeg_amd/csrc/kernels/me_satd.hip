/*
 * me_satd.hip — the exhaustive search with the SATD cost on the matrix cores (round 5).
 *
 * What is computed (hadamard8_diff8x8_c / hadamard8_diff16_c, libavcodec/me_cmp.c:514-562,933-950, as ff_me_search_esa drives them,
 * libavfilter/motion_estimation.c:60-100): for every candidate of the window, the sum over the macroblock's 8 x 8 blocks of
 * sum |H8 (cur - ref) H8^T|.  The butterfly network of the reference yields the 64 coefficients of the Sylvester-Hadamard transform in
 * some order; the absolute sum does not see the order.
 *
 * The form: the 2-D transform of an 8 x 8 block IS a product with the 64 x 64 matrix H8 (x) H8, whose entries are +-1 — a DENSE int8
 * matrix product with K = 64 pixels, M = 64 coefficients, N = candidates:  v_mfma_i32_16x16x64_i8, four M-tiles per 16 candidates.
 *   A (constant, 16 VGPRs per lane):  row m = (u, v), column k = pixel (y, x):  H[u][y] * H[v][x]
 *   B: lane (n, g) = candidate n of the tile, rows 2g and 2g + 1 of its 8 x 8 block as they lie in the window: two 8-byte LDS reads
 *      (samples - 128, so that they are int8: the window is stored with bit 7 flipped)
 *   C: BIAS - T(cur block - 128), per lane the four coefficient rows it will receive (T is linear: T(ref) - T(cur) = T(ref - cur);
 *      the constant 128 cancels)
 *   D = BIAS + T(ref - cur):  v_sad_u32 against BIAS is |coefficient| accumulated, one instruction per coefficient
 * against the packed-int16 butterflies' 640 lane-operations per candidate: 64 v_sad_u32 + 16 MFMA + 8 LDS reads per 16 candidates and
 * lane.  A lane ends with the partial sum of its four coefficient rows per M-tile; the four lane groups of a candidate meet in an LDS
 * cost array (ds_add_u32), which the wave then scans for the reference's winner (first minimum in raster order, the zero vector
 * kept unless a candidate is strictly cheaper).
 *
 * The window's LDS reads are unaligned 8-byte runs at any byte phase.  Instead of funnel shifts per read the window is staged FOUR
 * times, copy s shifted by s bytes (dword j of copy s = bytes s + 4j .. s + 4j + 3 of the row): a candidate at column cx reads copy
 * cx & 3 at dword cx >> 2 — the staging is 16 loads per lane, the search saves 16 v_alignbyte per 16 candidates.
 *
 * MFMA results have no interlock against the VALU on this part and the compiler pads only what it emits itself, so the MFMAs and
 * their v_sad_u32 are asm blocks (half a tile each: 8 + 32) with a fixed register quartet v[112:127] for the products: MFMA k + 3 is
 * issued before the sums of MFMA k, four v_sad_u32 sit between two MFMAs.
 *
 * What bounds it (tools/ubench/mfma_i8_rate.hip, profiles/r05_mfma_i8_rate.txt; four waves per SIMD): the MFMA alone issues every
 * 7.5 ns per SIMD, four v_sad_u32 alone take 8.8 ns, the two interleaved 12.6 ns — an MFMA costs the VALU port about two ordinary
 * issue slots on top of its own pipe time.  A macroblock at R = 7 is 960 v_sad_u32 + 253 MFMAs + ~400 other instructions:
 * 960 x 2.2 + 253 x 3.8 + 400 x 2.1 ns = 3.9 us per SIMD, measured 4.5.  One instruction per coefficient is the floor of this form
 * (int8 products cannot scale a second coefficient into the upper half of an accumulator, so v_sad_u16 on pairs is out of reach).
 * FFHIP_ME_SATD_PART = m / s (measure build) runs the loop with only its MFMAs / only its sums: 0.97 / 0.81 ms against 1.15 whole.
 */
#include "common.h"
#include "me_kernels.h"

typedef int ms_i4 __attribute__((ext_vector_type(4)));

struct MsHadTab { uint32_t a[4][64][4]; }; /* [M-tile][lane][dword]: the lane's 16 K-bytes of H8 (x) H8 */
constexpr int ms_par(int v) { return (v ^ (v >> 1) ^ (v >> 2)) & 1; }
constexpr MsHadTab ms_had_make()
{
    MsHadTab t = {};
    for (int mt = 0; mt < 4; mt++)
        for (int l = 0; l < 64; l++) {
            const int m = 16 * mt + (l & 15), u = m >> 3, v = m & 7, g = l >> 4;
            for (int s = 0; s < 16; s++) {
                const int y = 2 * g + (s >> 3), x = s & 7;
                const bool neg = (ms_par(u & y) ^ ms_par(v & x)) != 0;
                t.a[mt][l][s >> 2] |= (neg ? 0xFFu : 0x01u) << (8 * (s & 3));
            }
        }
    return t;
}
__device__ const MsHadTab ms_had_tab = ms_had_make();

#define MS_BIAS (1 << 20)

__device__ __forceinline__ void ms_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/*
 * 16 MFMAs (operand B q with the accumulator inputs c[4q .. 4q + 3]) and the absolute sums of their 16 x 4 results: sum q receives
 * B q's.  Product k lands in quartet k & 3 of v[112:127]; its sums are issued behind MFMA k + 3.
 */
#define MS_D0 "112:115"
#define MS_D1 "116:119"
#define MS_D2 "120:123"
#define MS_D3 "124:127"
#define MS_MF(d, a, b, c) "v_mfma_i32_16x16x64_i8 v[" d "], %[" a "], %[" b "], %[" c "]\n\t"
#define MS_S1(s, r0, r1, r2, r3) /* the first four of a sum: starts from 0 */                                       \
    "v_sad_u32 %[" s "], v" r0 ", %[k], 0\n\tv_sad_u32 %[" s "], v" r1 ", %[k], %[" s "]\n\t"                       \
    "v_sad_u32 %[" s "], v" r2 ", %[k], %[" s "]\n\tv_sad_u32 %[" s "], v" r3 ", %[k], %[" s "]\n\t"
#define MS_SN(s, r0, r1, r2, r3)                                                                                    \
    "v_sad_u32 %[" s "], v" r0 ", %[k], %[" s "]\n\tv_sad_u32 %[" s "], v" r1 ", %[k], %[" s "]\n\t"                \
    "v_sad_u32 %[" s "], v" r2 ", %[k], %[" s "]\n\tv_sad_u32 %[" s "], v" r3 ", %[k], %[" s "]\n\t"
#define MS_S1_0(s) MS_S1(s, "112", "113", "114", "115")
#define MS_SN_1(s) MS_SN(s, "116", "117", "118", "119")
#define MS_SN_2(s) MS_SN(s, "120", "121", "122", "123")
#define MS_SN_3(s) MS_SN(s, "124", "125", "126", "127")

#define MS_SN_0(s) MS_SN(s, "112", "113", "114", "115")
#define MS_INS                                                                                                                            \
    [a0] "v"(A[0]), [a1] "v"(A[1]), [a2] "v"(A[2]), [a3] "v"(A[3]), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [c0] "v"(c[0]),  \
        [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), [c8] "v"(c[8]),       \
        [c9] "v"(c[9]), [c10] "v"(c[10]), [c11] "v"(c[11]), [c12] "v"(c[12]), [c13] "v"(c[13]), [c14] "v"(c[14]), [c15] "v"(c[15]), [k] "s"(kb)
#define MS_CLOB "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
/* the 16 MFMAs in their order; X(q) = the four sums that go between */
#define MS_BODY(SA, SB, SC, SD, FB, FC, FD)                                                                                                \
    "s_nop 1\n\t"                                                                                                                          \
    MS_MF(MS_D0, "a0", "b0", "c0") MS_MF(MS_D1, "a1", "b0", "c1") MS_MF(MS_D2, "a2", "b0", "c2") MS_MF(MS_D3, "a3", "b0", "c3")              \
    MS_S1_0(SA) MS_MF(MS_D0, "a0", "b1", "c4") MS_SN_1(SA) MS_MF(MS_D1, "a1", "b1", "c5")                                                    \
    MS_SN_2(SA) MS_MF(MS_D2, "a2", "b1", "c6") MS_SN_3(SA) MS_MF(MS_D3, "a3", "b1", "c7")                                                    \
    FB(SB) MS_MF(MS_D0, "a0", "b2", "c8") MS_SN_1(SB) MS_MF(MS_D1, "a1", "b2", "c9")                                                         \
    MS_SN_2(SB) MS_MF(MS_D2, "a2", "b2", "c10") MS_SN_3(SB) MS_MF(MS_D3, "a3", "b2", "c11")                                                  \
    FC(SC) MS_MF(MS_D0, "a0", "b3", "c12") MS_SN_1(SC) MS_MF(MS_D1, "a1", "b3", "c13")                                                       \
    MS_SN_2(SC) MS_MF(MS_D2, "a2", "b3", "c14") MS_SN_3(SC) MS_MF(MS_D3, "a3", "b3", "c15")                                                  \
    FD(SD) MS_SN_1(SD) MS_SN_2(SD) MS_SN_3(SD)

/* a whole tile with one sum per B operand: the 8 x 8 macroblock form, whose four B operands are four different tiles */
__device__ __forceinline__ void ms_tile16(const ms_i4 (&A)[4], const ms_i4 &b0, const ms_i4 &b1, const ms_i4 &b2, const ms_i4 &b3,
                                          const ms_i4 (&c)[16], uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3)
{
    const int kb = MS_BIAS;
    asm volatile(MS_BODY("s0", "s1", "s2", "s3", MS_S1_0, MS_S1_0, MS_S1_0)
                 : [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [s3] "=&v"(s3) : MS_INS : MS_CLOB);
}

/*
 * Half a tile — 8 MFMAs (two B operands), 32 sums — so that the LDS reads of the other half are in flight meanwhile: the loop below
 * asks for blocks 2, 3 before the first half and for the next tile's blocks 0, 1 before the second.  FIRST: the sums start here.
 * FOUR sums, one per result register of a quartet: a v_sad_u32 that waits for the one before it issues every ~12 cycles
 * (tools/ubench/mfma_i8_rate.hip: 5.1 ns a piece from one wave), four chains issue back to back.
 */
#define MS_Q1(r0, r1, r2, r3)                                                                                                              \
    "v_sad_u32 %[s0], v" r0 ", %[k], 0\n\tv_sad_u32 %[s1], v" r1 ", %[k], 0\n\t"                                                            \
    "v_sad_u32 %[s2], v" r2 ", %[k], 0\n\tv_sad_u32 %[s3], v" r3 ", %[k], 0\n\t"
#define MS_QN(r0, r1, r2, r3)                                                                                                              \
    "v_sad_u32 %[s0], v" r0 ", %[k], %[s0]\n\tv_sad_u32 %[s1], v" r1 ", %[k], %[s1]\n\t"                                                    \
    "v_sad_u32 %[s2], v" r2 ", %[k], %[s2]\n\tv_sad_u32 %[s3], v" r3 ", %[k], %[s3]\n\t"
#define MS_Q1_0 MS_Q1("112", "113", "114", "115")
#define MS_QN_0 MS_QN("112", "113", "114", "115")
#define MS_QN_1 MS_QN("116", "117", "118", "119")
#define MS_QN_2 MS_QN("120", "121", "122", "123")
#define MS_QN_3 MS_QN("124", "125", "126", "127")
#define MS_HBODY(F0)                                                                                                                       \
    "s_nop 1\n\t"                                                                                                                          \
    MS_MF(MS_D0, "a0", "b0", "c0") MS_MF(MS_D1, "a1", "b0", "c1") MS_MF(MS_D2, "a2", "b0", "c2") MS_MF(MS_D3, "a3", "b0", "c3")              \
    F0 MS_MF(MS_D0, "a0", "b1", "c4") MS_QN_1 MS_MF(MS_D1, "a1", "b1", "c5")                                                                 \
    MS_QN_2 MS_MF(MS_D2, "a2", "b1", "c6") MS_QN_3 MS_MF(MS_D3, "a3", "b1", "c7")                                                            \
    MS_QN_0 MS_QN_1 MS_QN_2 MS_QN_3
#ifdef FFHIP_MEASURE /* FFHIP_ME_SATD_PART = m / s: the loop with only its MFMAs / only its sums (wrong results: where the time goes) */
#define MS_HBODY_M                                                                                                                         \
    "s_nop 1\n\t"                                                                                                                          \
    MS_MF(MS_D0, "a0", "b0", "c0") MS_MF(MS_D1, "a1", "b0", "c1") MS_MF(MS_D2, "a2", "b0", "c2") MS_MF(MS_D3, "a3", "b0", "c3")              \
    MS_MF(MS_D0, "a0", "b1", "c4") MS_MF(MS_D1, "a1", "b1", "c5") MS_MF(MS_D2, "a2", "b1", "c6") MS_MF(MS_D3, "a3", "b1", "c7")              \
    "s_nop 7\n\ts_nop 7\n\tv_mov_b32 %[s0], v112\n\tv_mov_b32 %[s1], v113\n\tv_mov_b32 %[s2], v114\n\tv_mov_b32 %[s3], v115\n\t"
#define MS_HBODY_S(F0) F0 MS_QN_1 MS_QN_2 MS_QN_3 MS_QN_0 MS_QN_1 MS_QN_2 MS_QN_3
#endif
template <bool FIRST, int PART = 0>
__device__ __forceinline__ void ms_half(const ms_i4 (&A)[4], const ms_i4 &b0, const ms_i4 &b1, const ms_i4 *c, uint32_t (&sum)[4])
{
    const int kb = MS_BIAS;
#define MS_HINS                                                                                                                           \
    [a0] "v"(A[0]), [a1] "v"(A[1]), [a2] "v"(A[2]), [a3] "v"(A[3]), [b0] "v"(b0), [b1] "v"(b1), [c0] "v"(c[0]), [c1] "v"(c[1]),               \
        [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), [k] "s"(kb)
#define MS_HOUT1 [s0] "=&v"(sum[0]), [s1] "=&v"(sum[1]), [s2] "=&v"(sum[2]), [s3] "=&v"(sum[3])
#define MS_HOUTN [s0] "+v"(sum[0]), [s1] "+v"(sum[1]), [s2] "+v"(sum[2]), [s3] "+v"(sum[3])
#ifdef FFHIP_MEASURE
    if (PART == 1) {
        asm volatile(MS_HBODY_M : MS_HOUT1 : MS_HINS : MS_CLOB);
        return;
    }
    if (PART == 2) {
        if (FIRST)
            asm volatile(MS_HBODY_S(MS_Q1_0) : MS_HOUT1 : MS_HINS : MS_CLOB);
        else
            asm volatile(MS_HBODY_S(MS_QN_0) : MS_HOUTN : MS_HINS : MS_CLOB);
        return;
    }
#endif
    if (FIRST)
        asm volatile(MS_HBODY(MS_Q1_0) : MS_HOUT1 : MS_HINS : MS_CLOB);
    else
        asm volatile(MS_HBODY(MS_QN_0) : MS_HOUTN : MS_HINS : MS_CLOB);
#undef MS_HINS
#undef MS_HOUT1
#undef MS_HOUTN
}

/* the lane's B operand of one 8 x 8 block: rows 2g, 2g + 1 (the caller's address is row 2g), two dwords each */
__device__ __forceinline__ ms_i4 ms_rows(const uint32_t *p, int pitchd)
{
    ms_i4 b;
    b.x = (int)p[0];
    b.y = (int)p[1];
    b.z = (int)p[pitchd];
    b.w = (int)p[pitchd + 1];
    return b;
}
__device__ __forceinline__ uint32_t ms_mul24(uint32_t a, uint32_t b) /* b: uniform */
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}

/*
 * One wave per macroblock, WPB macroblocks of a row per workgroup.  LDS of a wave, in dwords: the current block (MB * MB / 4); the
 * candidates' window offsets and their costs, one pair of arrays padded to whole tiles (the lanes past the last candidate repeat it
 * and add into the padding); four copies of the window (rows of pitchd dwords, `cstride` apart).  PD: pitchd at compile time (the
 * eight reads of a tile then share one address register) or 0.
 *
 * The offsets are a table because the four lane groups of a tile would otherwise each derive the same (row, column, copy) from the
 * candidate's index — 12 instructions per tile where the table costs them once per 64 candidates.
 */
template <int MB, int WPB, int PD, int PART = 0>
__global__ __launch_bounds__(64 * WPB) void k_me_esa_satd_mx(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride,
                                                            size_t frame_pitch, int R, int16_t *mv_out, uint32_t *cost_out, int pitchd_rt,
                                                            int cstride, int lds_per_wave)
{
    extern __shared__ __align__(16) uint8_t lds_all[];
    constexpr int LOG2 = MB == 16 ? 4 : 3;
    constexpr int NB = (MB / 8) * (MB / 8);
    const int pitchd = PD ? PD : pitchd_rt;
    const int bw = width >> LOG2, bh = height >> LOG2;
    const int wave = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int bx = blockIdx.x * WPB + wave, by = blockIdx.y, f = blockIdx.z;
    if (WPB > 1 && bx >= bw)
        return;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4;
    const int ncmax = (((2 * R + 1) * (2 * R + 1) + 63) & ~63) + 16; /* whole tiles, whole groups of four tiles, a tile of padding (the loop reads one ahead) */
    uint32_t *cblk = reinterpret_cast<uint32_t *>(lds_all + (size_t)wave * lds_per_wave);
    uint32_t *offs = cblk + MB * MB / 4;
    uint32_t *cost = offs + ncmax;
    uint32_t *win = cost + ncmax;
    const int x_mb = bx << LOG2, y_mb = by << LOG2;
    const int lim_x = (bw - 1) << LOG2, lim_y = (bh - 1) << LOG2;
    const int x0 = max(x_mb - R, 0), y0 = max(y_mb - R, 0);
    const int x1 = min(x_mb + R, lim_x), y1 = min(y_mb + R, lim_y);
    const int ncx = x1 - x0 + 1, ncy = y1 - y0 + 1, ncand = ncx * ncy;
    const int wcols = ncx + MB - 1, wrows = ncy + MB - 1;
    const int ntiles = MB == 16 ? (ncand + 15) >> 4 : ((ncand + 63) >> 6) << 2;
    const uint8_t *cf = cur + (size_t)f * frame_pitch, *rf = ref + (size_t)f * frame_pitch;

    /* idx / ncx by a multiplication: ncand * ncx < 2^20 for every R the launcher admits */
    const uint32_t magic = (uint32_t)__builtin_amdgcn_readfirstlane((int)((1u << 20) / (uint32_t)ncx + 1u));
    /* staging: everything with bit 7 flipped (sample - 128 as int8).  The loads of a pass (16 per lane) are all asked for before the
     * first is used — a wave alone pays one memory latency per pass, not four — and the tables are computed under them. */
    {
        const int dwr = (wcols + 3) >> 2;
        int lg = 2;
        while ((1 << lg) < dwr)
            lg++;
        /* a dword that would cross the picture's right edge is fetched where the row ends and shifted down: the bytes past the edge
         * belong to no candidate, they only must not be read */
        const int xlast = width - 4, total = wrows << lg;
        constexpr int DPR = MB / 4;
        uint32_t cv = 0; /* the current block: MB * MB / 4 <= 64 dwords, one per lane */
        const uint32_t ustride = (uint32_t)stride; /* offsets inside a frame fit 32 bits (the launcher checks) */
        for (int base = 0; base < total; base += 256) {
            uint32_t v[4][4];
            int sh[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = min(base + 64 * u + lane, total - 1);
                const int r = i >> lg, j = min(i & ((1 << lg) - 1), dwr - 1);
                const uint32_t rowoff = ms_mul24((uint32_t)(y0 + r), ustride);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int xb = x0 + 4 * j + s, xl = min(xb, xlast);
                    __builtin_memcpy(&v[u][s], rf + (rowoff + (uint32_t)xl), 4);
                    sh[u][s] = (8 * (xb - xl)) & 31;
                }
            }
            if (base == 0) {
                const int ic = min(lane, MB * DPR - 1);
                __builtin_memcpy(&cv, cf + (ms_mul24((uint32_t)(y_mb + ic / DPR), ustride) + (uint32_t)(x_mb + 4 * (ic % DPR))), 4);
                for (int i = lane; i < 16 * ntiles + 16; i += 64) {
                    const int idc = min(i, ncand - 1);
                    const int cy = (int)(ms_mul24((uint32_t)idc, magic) >> 20), cx = idc - (int)ms_mul24((uint32_t)cy, (uint32_t)ncx);
                    offs[i] = 4u * (ms_mul24((uint32_t)cx & 3, (uint32_t)cstride) + ms_mul24((uint32_t)cy, (uint32_t)pitchd) + ((uint32_t)cx >> 2));
                    cost[i] = 0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + 64 * u + lane;
                const int r = i >> lg, j = i & ((1 << lg) - 1);
                if (i < total && j < dwr) {
                    uint32_t *o = win + __mul24(r, pitchd) + j;
#pragma unroll
                    for (int s = 0; s < 4; s++)
                        o[s * cstride] = (v[u][s] >> sh[u][s]) ^ 0x80808080u;
                }
            }
            if (base == 0) {
                asm volatile("" : "+v"(cv)); /* used here, not before the loads above are on their way */
                if (lane < MB * DPR)
                    cblk[lane] = cv ^ 0x80808080u;
            }
        }
    }
    ms_wave_sync();

    ms_i4 A[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
        const uint4 q = *reinterpret_cast<const uint4 *>(ms_had_tab.a[mt][lane]);
        A[mt] = (ms_i4){ (int)q.x, (int)q.y, (int)q.z, (int)q.w };
    }
    /* BIAS - T(current block b), in the accumulator layout (every column of the product is the same block): the product with -A,
     * and -1 <-> +1 of a byte is ^ 0xFE */
    ms_i4 init[NB][4];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int sy = b / (MB / 8), sx = b % (MB / 8);
        const ms_i4 cb = ms_rows(cblk + (8 * sy + 2 * g) * (MB / 4) + 2 * sx, MB / 4);
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
            init[b][mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[mt] ^ (int)0xFEFEFEFEu, cb, (ms_i4){ MS_BIAS, MS_BIAS, MS_BIAS, MS_BIAS }, 0, 0, 0);
    }

    /* the lane's view: its candidate of tile t is entry 16 t + (lane & 15) of both tables, its rows start 2g below the candidate's */
    const uint32_t *po = offs + (lane & 15);
    const uint8_t *wg = reinterpret_cast<const uint8_t *>(win + 2 * g * pitchd);
    uint32_t *pc = cost + (lane & 15);
    if (MB == 16) {
        ms_i4 c[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            c[i] = init[i >> 2][i & 3];
        /* software pipeline over half tiles (the offsets table has one tile of padding behind the last) */
        const uint32_t *p = reinterpret_cast<const uint32_t *>(wg + *po);
        ms_i4 b0 = ms_rows(p, pitchd), b1 = ms_rows(p + 2, pitchd);
        for (int t = 0; t < ntiles; t++, pc += 16) {
            po += 16;
            const uint32_t on = *po;
            const uint32_t *q = p + 8 * pitchd;
            const ms_i4 b2 = ms_rows(q, pitchd), b3 = ms_rows(q + 2, pitchd);
            uint32_t sum[4];
            ms_half<true, PART>(A, b0, b1, c, sum);
            p = reinterpret_cast<const uint32_t *>(wg + on);
            b0 = ms_rows(p, pitchd);
            b1 = ms_rows(p + 2, pitchd);
            ms_half<false, PART>(A, b2, b3, c + 8, sum);
            __hip_atomic_fetch_add(pc, sum[0] + sum[1] + sum[2] + sum[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        ms_i4 c[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            c[i] = init[0][i & 3];
        for (int t = 0; t < ntiles; t += 4, po += 64, pc += 64) {
            ms_i4 b[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                b[k] = ms_rows(reinterpret_cast<const uint32_t *>(wg + po[16 * k]), pitchd);
            uint32_t s[4];
            ms_tile16(A, b[0], b[1], b[2], b[3], c, s[0], s[1], s[2], s[3]);
#pragma unroll
            for (int k = 0; k < 4; k++)
                __hip_atomic_fetch_add(pc + 16 * k, s[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    ms_wave_sync();

    /* the winner: smaller cost, then smaller raster index; the zero vector unless a candidate is strictly cheaper */
    const int ci0 = (y_mb - y0) * ncx + (x_mb - x0);
    uint32_t best = 0xFFFFFFFFu, best_ci = 0xFFFFFFFFu;
    for (int ci = lane; ci < ncand; ci += 64) {
        const uint32_t cc = cost[ci];
        if (cc < best) {
            best = cc;
            best_ci = (uint32_t)ci;
        }
    }
    unsigned long long key = ((unsigned long long)best << 32) | best_ci;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(key, s, 64);
        key = o < key ? o : key;
    }
    if (lane == 0) {
        const uint32_t cost0 = cost[ci0];
        const uint32_t mc = (uint32_t)(key >> 32), mi = (uint32_t)key;
        int mvx = x_mb, mvy = y_mb;
        uint32_t cw = cost0;
        if (mc < cost0) {
            const uint32_t my = (mi * magic) >> 20;
            mvx = x0 + (int)(mi - my * (uint32_t)ncx);
            mvy = y0 + (int)my;
            cw = mc;
        }
        const size_t b = ((size_t)f * bh + by) * bw + bx;
        mv_out[2 * b] = (int16_t)mvx;
        mv_out[2 * b + 1] = (int16_t)mvy;
        cost_out[b] = cw;
    }
}

/* 1 = launched; 0 = not this kernel's case (the caller keeps the VALU forms) */
int ffhip_launch_me_esa_satd_mx(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride, size_t frame_pitch, int nframes,
                                int mb_size, int R, int16_t *mv_out, uint32_t *cost_out, hipStream_t stream)
{
    const int lg = mb_size == 16 ? 4 : 3;
    const int bw = width >> lg, bh = height >> lg;
    const int nc = (2 * R + 1) * (2 * R + 1);
    if ((long long)nc * (2 * R + 1) >= (1 << 20))
        return 0;
    if (stride <= 0 || stride >= (1 << 24) || height >= (1 << 24) || (long long)height * stride >= (1LL << 32))
        return 0; /* the kernel addresses a frame with 32-bit offsets */
    const int pitchd = (((2 * R + mb_size + 3) >> 2) + 1) | 1;
    const int cstride = (2 * R + mb_size) * pitchd + 1;
    const size_t ncmax = (((size_t)nc + 63) & ~(size_t)63) + 16;
    const size_t lpw = (((size_t)mb_size * mb_size / 4 + 2 * ncmax + 4 * (size_t)cstride) * 4 + 15) & ~(size_t)15;
    if (lpw > 64 * 1024)
        return 0;
    const int wpb = lpw * 4 <= 64 * 1024 ? 4 : 1;
    const dim3 grid(cdiv(bw, wpb), bh, nframes), block(64 * wpb);
#define MSL(M, W, P) hipLaunchKernelGGL((k_me_esa_satd_mx<M, W, P>), grid, block, lpw * W, stream, cur, ref, width, height, stride, frame_pitch, R, \
                                        mv_out, cost_out, pitchd, cstride, (int)lpw)
    if (mb_size == 16) {
#ifdef FFHIP_MEASURE
        const char *ep = FFHIP_KNOB("FFHIP_ME_SATD_PART");
        if (ep && wpb == 4 && pitchd == 9) {
            if (ep[0] == 'm')
                hipLaunchKernelGGL((k_me_esa_satd_mx<16, 4, 9, 1>), grid, block, lpw * 4, stream, cur, ref, width, height, stride, frame_pitch, R, mv_out,
                                   cost_out, pitchd, cstride, (int)lpw);
            else
                hipLaunchKernelGGL((k_me_esa_satd_mx<16, 4, 9, 2>), grid, block, lpw * 4, stream, cur, ref, width, height, stride, frame_pitch, R, mv_out,
                                   cost_out, pitchd, cstride, (int)lpw);
            return 1;
        }
#endif
        if (wpb == 4 && pitchd == 9) MSL(16, 4, 9); /* R = 7, vf_mestimate's default search_param */
        else if (wpb == 4) MSL(16, 4, 0);
        else MSL(16, 1, 0);
    } else {
        if (wpb == 4) MSL(8, 4, 0); else MSL(8, 1, 0);
    }
#undef MSL
    return 1;
}
