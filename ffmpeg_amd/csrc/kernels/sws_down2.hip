/*
 * sws_down2.hip — the fused H+V scaler for EXACT 2:1 down-scaling with banks of up to 8 taps in both directions (bicubic /
 * bilinear 4K -> 1080p), planes and byte-interleaved U/V pairs (NV12 / NV21) in and out.
 *
 * Arithmetic: hScale8To15_c (libswscale/swscale.c:128-142), nv12ToUV_c (input.c:936), yuv2planeX_8_c / yuv2nv12cX_c
 * (output.c:468-529) — int32 sums, >> 7 and min(., 32767) for the horizontal pass, the 64 << 12 seed, >> 19 and the clip
 * to 8 bits for the vertical one.  Same results as sws_lwalk.hip / sws_scale.hip, bit for bit (tests/test_gpu_sws_fast.py).
 *
 * What exact 2:1 buys over the wide walker (sws_lwalk.hip: a source row's window goes through 9.8 KB of LDS per wave, which
 * holds the CU at a few waves):
 *  - Window positions are REGULAR: output x reads source samples 2x - 3 .. 2x + 4 (initFilter, libswscale/utils.c:519-561,
 *    folds the taps that fall outside the row onto the first / last sample: the regular bank over an edge-REPLICATED row, with
 *    its own coefficients in the columns next to either edge — ffhip_down2_virtual_bank re-expresses every bank row that way,
 *    tap by tap, else this kernel is not used).  4 adjacent outputs read 14 adjacent bytes; the SEVEN (s[2m+1], s[2m+2]) int16
 *    pairs of those bytes serve all sixteen v_dot2_i32_i16 of the row.  No position table, no LDS.
 *  - The vertical schedule is STATIC and pair-aligned: output row y reads filtered rows 2y-3 .. 2y+4 = the row pairs
 *    T(y-1) .. T(y+2), T(t) = (row 2t-1, row 2t); a step filters two new source rows, packs them into ONE new pair and keeps a
 *    ring of four: 4 dots per sample, coefficient pairs in SGPRs (one s_load_dwordx16 per four steps).
 * Per 4 output samples: 2 x 27 (horizontal, plane; 30 for a U/V pair) + 4 (pack) + 16 + 4 (vertical) VALU instructions = about 20
 * per sample — a fifth of the chip's issue rate at the HBM roof of the 5 bytes a sample moves.
 *
 * Geometry: a wave owns 64 lanes x 4 output bytes of a strip of output rows and walks down the source rows, four rows in
 * flight.  A lane reads the 16 (pair: 24) source bytes at 8g - 4 (8g - 8) of every source row and writes the dword at 4g.
 */
#include <stdlib.h>
#include <vector>

#include "common.h"
#include "sws_kernels.h"

typedef short dn_s2 __attribute__((ext_vector_type(2)));
typedef uint32_t dn_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t dn_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t dn_u16 __attribute__((ext_vector_type(16)));
typedef dn_u4 __attribute__((aligned(4))) dn_u4a;
typedef dn_u2 __attribute__((aligned(4))) dn_u2a;
typedef const uint8_t __attribute__((address_space(1))) *dn_gcp;
typedef uint8_t __attribute__((address_space(1))) *dn_gp;
typedef const dn_u4a __attribute__((address_space(1))) *dn_gc4;
typedef const dn_u2a __attribute__((address_space(1))) *dn_gc2;
typedef uint32_t __attribute__((address_space(1))) *dn_g1;
typedef const dn_u16 __attribute__((address_space(4))) *dn_cc16; /* constant address space: scalar loads */

__device__ __forceinline__ int dn_dot(uint32_t p, uint32_t c, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(dn_s2, p), __builtin_bit_cast(dn_s2, c), acc, false);
}


/*
 * The dots of a row are hand-scheduled asm blocks (as in sws_up2.hip): VOP3P v_dot2_i32_i16 with an inline 0 / the seed as the
 * first addend (the compiler's v_dot2c form costs a v_mov per chain), four chains interleaved so that every DOT result is read
 * by a non-DOT instruction >= 3 instructions after it was written (the gfx950 DOT hazard), and every value that leaves a block
 * is the result of a plain VALU instruction.
 */
/* plane: d[i] = (sum_k P[i + k] . cf[4 i + k]) >> 7 */
__device__ __forceinline__ void dn_h4_plane(int (&d)[4], const uint32_t (&P)[7], const uint32_t (&cf)[16])
{
    asm("v_dot2_i32_i16 %0, %4, %11, 0\n\t"
        "v_dot2_i32_i16 %1, %5, %15, 0\n\t"
        "v_dot2_i32_i16 %2, %6, %19, 0\n\t"
        "v_dot2_i32_i16 %3, %7, %23, 0\n\t"
        "v_dot2_i32_i16 %0, %5, %12, %0\n\t"
        "v_dot2_i32_i16 %1, %6, %16, %1\n\t"
        "v_dot2_i32_i16 %2, %7, %20, %2\n\t"
        "v_dot2_i32_i16 %3, %8, %24, %3\n\t"
        "v_dot2_i32_i16 %0, %6, %13, %0\n\t"
        "v_dot2_i32_i16 %1, %7, %17, %1\n\t"
        "v_dot2_i32_i16 %2, %8, %21, %2\n\t"
        "v_dot2_i32_i16 %3, %9, %25, %3\n\t"
        "v_dot2_i32_i16 %0, %7, %14, %0\n\t"
        "v_dot2_i32_i16 %1, %8, %18, %1\n\t"
        "v_dot2_i32_i16 %2, %9, %22, %2\n\t"
        "v_dot2_i32_i16 %3, %10, %26, %3\n\t"
        "v_ashrrev_i32 %0, 7, %0\n\t"
        "v_ashrrev_i32 %1, 7, %1\n\t"
        "v_ashrrev_i32 %2, 7, %2\n\t"
        "v_ashrrev_i32 %3, 7, %3"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]), "v"(P[4]), "v"(P[5]), "v"(P[6]),
          "v"(cf[0]), "v"(cf[1]), "v"(cf[2]), "v"(cf[3]), "v"(cf[4]), "v"(cf[5]), "v"(cf[6]), "v"(cf[7]),
          "v"(cf[8]), "v"(cf[9]), "v"(cf[10]), "v"(cf[11]), "v"(cf[12]), "v"(cf[13]), "v"(cf[14]), "v"(cf[15]));
}
/* pair: d[0] = A[0..3] . cf[0..3], d[1] = B[0..3] . cf[0..3], d[2] = A[1..4] . cf[4..7], d[3] = B[1..4] . cf[4..7], each >> 7 */
__device__ __forceinline__ void dn_h4_pair(int (&d)[4], const uint32_t (&A)[5], const uint32_t (&B)[5], const uint32_t (&cf)[8])
{
    asm("v_dot2_i32_i16 %0, %4, %14, 0\n\t"
        "v_dot2_i32_i16 %1, %9, %14, 0\n\t"
        "v_dot2_i32_i16 %2, %5, %18, 0\n\t"
        "v_dot2_i32_i16 %3, %10, %18, 0\n\t"
        "v_dot2_i32_i16 %0, %5, %15, %0\n\t"
        "v_dot2_i32_i16 %1, %10, %15, %1\n\t"
        "v_dot2_i32_i16 %2, %6, %19, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %19, %3\n\t"
        "v_dot2_i32_i16 %0, %6, %16, %0\n\t"
        "v_dot2_i32_i16 %1, %11, %16, %1\n\t"
        "v_dot2_i32_i16 %2, %7, %20, %2\n\t"
        "v_dot2_i32_i16 %3, %12, %20, %3\n\t"
        "v_dot2_i32_i16 %0, %7, %17, %0\n\t"
        "v_dot2_i32_i16 %1, %12, %17, %1\n\t"
        "v_dot2_i32_i16 %2, %8, %21, %2\n\t"
        "v_dot2_i32_i16 %3, %13, %21, %3\n\t"
        "v_ashrrev_i32 %0, 7, %0\n\t"
        "v_ashrrev_i32 %1, 7, %1\n\t"
        "v_ashrrev_i32 %2, 7, %2\n\t"
        "v_ashrrev_i32 %3, 7, %3"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]),
          "v"(cf[0]), "v"(cf[1]), "v"(cf[2]), "v"(cf[3]), "v"(cf[4]), "v"(cf[5]), "v"(cf[6]), "v"(cf[7]));
}
/* the same two blocks with the shift in an SGPR: samples above 8 bits shift by depth - 1 (hScale16To15_c, swscale.c:99-126) */
__device__ __forceinline__ void dn_h4_plane_s(int (&d)[4], const uint32_t (&P)[7], const uint32_t (&cf)[16], int sh)
{
    asm("v_dot2_i32_i16 %0, %4, %11, 0\n\t"
        "v_dot2_i32_i16 %1, %5, %15, 0\n\t"
        "v_dot2_i32_i16 %2, %6, %19, 0\n\t"
        "v_dot2_i32_i16 %3, %7, %23, 0\n\t"
        "v_dot2_i32_i16 %0, %5, %12, %0\n\t"
        "v_dot2_i32_i16 %1, %6, %16, %1\n\t"
        "v_dot2_i32_i16 %2, %7, %20, %2\n\t"
        "v_dot2_i32_i16 %3, %8, %24, %3\n\t"
        "v_dot2_i32_i16 %0, %6, %13, %0\n\t"
        "v_dot2_i32_i16 %1, %7, %17, %1\n\t"
        "v_dot2_i32_i16 %2, %8, %21, %2\n\t"
        "v_dot2_i32_i16 %3, %9, %25, %3\n\t"
        "v_dot2_i32_i16 %0, %7, %14, %0\n\t"
        "v_dot2_i32_i16 %1, %8, %18, %1\n\t"
        "v_dot2_i32_i16 %2, %9, %22, %2\n\t"
        "v_dot2_i32_i16 %3, %10, %26, %3\n\t"
        "v_ashrrev_i32 %0, %27, %0\n\t"
        "v_ashrrev_i32 %1, %27, %1\n\t"
        "v_ashrrev_i32 %2, %27, %2\n\t"
        "v_ashrrev_i32 %3, %27, %3"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]), "v"(P[4]), "v"(P[5]), "v"(P[6]),
          "v"(cf[0]), "v"(cf[1]), "v"(cf[2]), "v"(cf[3]), "v"(cf[4]), "v"(cf[5]), "v"(cf[6]), "v"(cf[7]),
          "v"(cf[8]), "v"(cf[9]), "v"(cf[10]), "v"(cf[11]), "v"(cf[12]), "v"(cf[13]), "v"(cf[14]), "v"(cf[15]), "s"(sh));
}
/* pair: d[0] = A[0..3] . cf[0..3], d[1] = B[0..3] . cf[0..3], d[2] = A[1..4] . cf[4..7], d[3] = B[1..4] . cf[4..7], each >> 7 */
__device__ __forceinline__ void dn_h4_pair_s(int (&d)[4], const uint32_t (&A)[5], const uint32_t (&B)[5], const uint32_t (&cf)[8], int sh)
{
    asm("v_dot2_i32_i16 %0, %4, %14, 0\n\t"
        "v_dot2_i32_i16 %1, %9, %14, 0\n\t"
        "v_dot2_i32_i16 %2, %5, %18, 0\n\t"
        "v_dot2_i32_i16 %3, %10, %18, 0\n\t"
        "v_dot2_i32_i16 %0, %5, %15, %0\n\t"
        "v_dot2_i32_i16 %1, %10, %15, %1\n\t"
        "v_dot2_i32_i16 %2, %6, %19, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %19, %3\n\t"
        "v_dot2_i32_i16 %0, %6, %16, %0\n\t"
        "v_dot2_i32_i16 %1, %11, %16, %1\n\t"
        "v_dot2_i32_i16 %2, %7, %20, %2\n\t"
        "v_dot2_i32_i16 %3, %12, %20, %3\n\t"
        "v_dot2_i32_i16 %0, %7, %17, %0\n\t"
        "v_dot2_i32_i16 %1, %12, %17, %1\n\t"
        "v_dot2_i32_i16 %2, %8, %21, %2\n\t"
        "v_dot2_i32_i16 %3, %13, %21, %3\n\t"
        "v_ashrrev_i32 %0, %22, %0\n\t"
        "v_ashrrev_i32 %1, %22, %1\n\t"
        "v_ashrrev_i32 %2, %22, %2\n\t"
        "v_ashrrev_i32 %3, %22, %3"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]),
          "v"(cf[0]), "v"(cf[1]), "v"(cf[2]), "v"(cf[3]), "v"(cf[4]), "v"(cf[5]), "v"(cf[6]), "v"(cf[7]), "s"(sh));
}
/* one output row of 4 samples: t[i] = kround + sum_k R_k[i] . c_k, bytes clip_u8(t[i] >> 19) packed in sample order */
__device__ __forceinline__ uint32_t dn_v4(const uint32_t (&R0)[4], const uint32_t (&R1)[4], const uint32_t (&R2)[4], const uint32_t (&R3)[4],
                                          uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int kround)
{
    uint32_t out;
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %1, %5, %21, %25\n\t"
        "v_dot2_i32_i16 %2, %6, %21, %25\n\t"
        "v_dot2_i32_i16 %3, %7, %21, %25\n\t"
        "v_dot2_i32_i16 %4, %8, %21, %25\n\t"
        "v_dot2_i32_i16 %1, %9, %22, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %22, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %22, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %22, %4\n\t"
        "v_dot2_i32_i16 %1, %13, %23, %1\n\t"
        "v_dot2_i32_i16 %2, %14, %23, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %23, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %23, %4\n\t"
        "v_dot2_i32_i16 %1, %17, %24, %1\n\t"
        "v_dot2_i32_i16 %2, %18, %24, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %24, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %24, %4\n\t"
        "v_ashr_pk_u8_i32 %0, %1, %2, 19\n\t"
        "s_nop 1\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 19 op_sel:[0,0,0,1]"
        : "=&v"(out), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(R0[0]), "v"(R0[1]), "v"(R0[2]), "v"(R0[3]), "v"(R1[0]), "v"(R1[1]), "v"(R1[2]), "v"(R1[3]),
          "v"(R2[0]), "v"(R2[1]), "v"(R2[2]), "v"(R2[3]), "v"(R3[0]), "v"(R3[1]), "v"(R3[2]), "v"(R3[3]),
          "s"(c0), "s"(c1), "s"(c2), "s"(c3), "v"(kround));
    return out;
}

/* the same row above 8 bits (yuv2planeX_10_c_template, output.c:341-360): t[i] >> (27 - depth) clipped to depth bits, two dwords */
__device__ __forceinline__ void dn_v4h(uint32_t &o0, uint32_t &o1, const uint32_t (&R0)[4], const uint32_t (&R1)[4], const uint32_t (&R2)[4],
                                       const uint32_t (&R3)[4], uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int kround, int sh, uint32_t maxpk)
{
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %2, %6, %22, %26\n\t"
        "v_dot2_i32_i16 %3, %7, %22, %26\n\t"
        "v_dot2_i32_i16 %4, %8, %22, %26\n\t"
        "v_dot2_i32_i16 %5, %9, %22, %26\n\t"
        "v_dot2_i32_i16 %2, %10, %23, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %23, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %23, %4\n\t"
        "v_dot2_i32_i16 %5, %13, %23, %5\n\t"
        "v_dot2_i32_i16 %2, %14, %24, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %24, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %24, %4\n\t"
        "v_dot2_i32_i16 %5, %17, %24, %5\n\t"
        "v_dot2_i32_i16 %2, %18, %25, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %25, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %25, %4\n\t"
        "v_dot2_i32_i16 %5, %21, %25, %5\n\t"
        "v_ashrrev_i32 %2, %27, %2\n\t"
        "v_ashrrev_i32 %3, %27, %3\n\t"
        "v_ashrrev_i32 %4, %27, %4\n\t"
        "v_ashrrev_i32 %5, %27, %5\n\t"
        "v_cvt_pk_i16_i32 %0, %2, %3\n\t"
        "v_cvt_pk_i16_i32 %1, %4, %5\n\t"
        "v_pk_max_i16 %0, %0, 0\n\t"
        "v_pk_max_i16 %1, %1, 0\n\t"
        "v_pk_min_i16 %0, %0, %28\n\t"
        "v_pk_min_i16 %1, %1, %28"
        : "=&v"(o0), "=&v"(o1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(R0[0]), "v"(R0[1]), "v"(R0[2]), "v"(R0[3]), "v"(R1[0]), "v"(R1[1]), "v"(R1[2]), "v"(R1[3]),
          "v"(R2[0]), "v"(R2[1]), "v"(R2[2]), "v"(R2[3]), "v"(R3[0]), "v"(R3[1]), "v"(R3[2]), "v"(R3[3]),
          "s"(c0), "s"(c1), "s"(c2), "s"(c3), "v"(kround), "v"(sh), "v"(maxpk));
}

/* ff_dither_8x8_128 (libswscale/output.c:70-79): the ordered dither of an 8-bit target fed from a deeper source */
__constant__ __attribute__((aligned(8))) uint8_t dn_dither[8][8] = {
    {  36, 68,  60, 92,  34, 66,  58, 90, }, { 100,  4, 124, 28,  98,  2, 122, 26, }, {  52, 84,  44, 76,  50, 82,  42, 74, },
    { 116, 20, 108, 12, 114, 18, 106, 10, }, {  32, 64,  56, 88,  38, 70,  62, 94, }, {  96,  0, 120, 24, 102,  6, 126, 30, },
    {  48, 80,  40, 72,  54, 86,  46, 78, }, { 112, 16, 104,  8, 118, 22, 110, 14, },
};
/* a row of an 8-bit target fed from a deeper source (yuv2planeX_8_c / yuv2nv12cX_c, output.c:468-529): every sample its own seed
 * dither << 12, >> 19, clip to 8 bits, four bytes in sample order */
__device__ __forceinline__ uint32_t dn_v4d(const uint32_t (&R0)[4], const uint32_t (&R1)[4], const uint32_t (&R2)[4], const uint32_t (&R3)[4],
                                           uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, const int (&sd)[4])
{
    uint32_t out;
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %1, %5, %21, %25\n\t"
        "v_dot2_i32_i16 %2, %6, %21, %26\n\t"
        "v_dot2_i32_i16 %3, %7, %21, %27\n\t"
        "v_dot2_i32_i16 %4, %8, %21, %28\n\t"
        "v_dot2_i32_i16 %1, %9, %22, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %22, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %22, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %22, %4\n\t"
        "v_dot2_i32_i16 %1, %13, %23, %1\n\t"
        "v_dot2_i32_i16 %2, %14, %23, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %23, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %23, %4\n\t"
        "v_dot2_i32_i16 %1, %17, %24, %1\n\t"
        "v_dot2_i32_i16 %2, %18, %24, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %24, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %24, %4\n\t"
        "v_ashr_pk_u8_i32 %0, %1, %2, 19\n\t"
        "s_nop 1\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 19 op_sel:[0,0,0,1]"
        : "=&v"(out), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(R0[0]), "v"(R0[1]), "v"(R0[2]), "v"(R0[3]), "v"(R1[0]), "v"(R1[1]), "v"(R1[2]), "v"(R1[3]),
          "v"(R2[0]), "v"(R2[1]), "v"(R2[2]), "v"(R2[3]), "v"(R3[0]), "v"(R3[1]), "v"(R3[2]), "v"(R3[3]),
          "s"(c0), "s"(c1), "s"(c2), "s"(c3), "v"(sd[0]), "v"(sd[1]), "v"(sd[2]), "v"(sd[3]));
    return out;
}

/* the same row, 8-bit pipeline, NOT clipped: t[i] >> 19 as int16 pairs — the luma plane of a packed-RGB target's first stage
 * (yuv2rgb_X reads the sums unclipped, libswscale/output.c:1814-1835; sws_y16rgb.hip is the second stage) */
__device__ __forceinline__ void dn_v4y(uint32_t &o0, uint32_t &o1, const uint32_t (&R0)[4], const uint32_t (&R1)[4], const uint32_t (&R2)[4],
                                       const uint32_t (&R3)[4], uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int kround)
{
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %2, %6, %22, %26\n\t"
        "v_dot2_i32_i16 %3, %7, %22, %26\n\t"
        "v_dot2_i32_i16 %4, %8, %22, %26\n\t"
        "v_dot2_i32_i16 %5, %9, %22, %26\n\t"
        "v_dot2_i32_i16 %2, %10, %23, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %23, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %23, %4\n\t"
        "v_dot2_i32_i16 %5, %13, %23, %5\n\t"
        "v_dot2_i32_i16 %2, %14, %24, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %24, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %24, %4\n\t"
        "v_dot2_i32_i16 %5, %17, %24, %5\n\t"
        "v_dot2_i32_i16 %2, %18, %25, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %25, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %25, %4\n\t"
        "v_dot2_i32_i16 %5, %21, %25, %5\n\t"
        "v_ashrrev_i32 %2, 19, %2\n\t"
        "v_ashrrev_i32 %3, 19, %3\n\t"
        "v_ashrrev_i32 %4, 19, %4\n\t"
        "v_ashrrev_i32 %5, 19, %5\n\t"
        "v_cvt_pk_i16_i32 %0, %2, %3\n\t"
        "v_cvt_pk_i16_i32 %1, %4, %5"
        : "=&v"(o0), "=&v"(o1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(R0[0]), "v"(R0[1]), "v"(R0[2]), "v"(R0[3]), "v"(R1[0]), "v"(R1[1]), "v"(R1[2]), "v"(R1[3]),
          "v"(R2[0]), "v"(R2[1]), "v"(R2[2]), "v"(R2[3]), "v"(R3[0]), "v"(R3[1]), "v"(R3[2]), "v"(R3[3]),
          "s"(c0), "s"(c1), "s"(c2), "s"(c3), "v"(kround));
}

/* HB: samples above 8 bits (little-endian uint16, 9..14 bits, P01x's in the high bits).  A lane then reads 32 source bytes per row
 * (plane: the 16 samples from 8g - 4, whose seven (s[2m+1], s[2m+2]) pairs are one v_alignbyte each) or 40 (pair: the ten (u, v)
 * columns from 4g - 3, pairs by v_perm as at 8 bits) and writes 8 destination bytes per row. */
/* OUT 1: the instantiation that carries the rarer output stages — HB 0: a job may write the int16 luma of a packed-RGB target (J.y16); HB 1: the
 * 8-bit target with the ordered dither.  OUT 0 is straight-line (round 6: uniform branches around a row's arithmetic made the compiler copy
 * the rows in flight at every join, and wait for them: docs/KERNELS.md R6.8) */
template <int PAIR, int HB = 0, int OUT = 0>
__device__ __forceinline__ void dn2_unit(const FFHipDn2Job &J, int frame, int gbase, int strip, int lane)
{
    constexpr int NQ = HB ? (PAIR ? 10 : 8) : (PAIR ? 6 : 4), NCF = PAIR ? 8 : 16;
    const int graw = gbase + lane;
    const bool act = graw < J.ngroups;
    const int g = min(graw, J.ngroups - 1);
    const bool lb = g == 0, rb = g == J.ngroups - 1;
    const bool border = gbase == 0 || gbase + 64 >= J.ngroups; /* wave-uniform */
    /* the first / last lane of a row loads its span inside the row and rebuilds the replicated bytes */
    const uint32_t soff = HB ? (uint32_t)(lb ? 0 : 16 * g - (PAIR ? 12 : 8) - (rb ? (PAIR ? 12 : 8) : 0))
                             : (uint32_t)(lb ? 0 : 8 * g - (PAIR ? 8 : 4) - (rb ? (PAIR ? 8 : 4) : 0));
    const uint32_t doff = (HB ? 8u : 4u) * (uint32_t)g;
    const int hsh = HB ? J.hb_sdepth - 1 : 7, vsh = HB ? 27 - J.hb_ddepth : 19;
    const uint32_t maxpk = HB ? ((1u << J.hb_ddepth) - 1) * 0x00010001u : 0;
    const int smsb = HB ? (J.hb_smsb ? 16 - J.hb_sdepth : 0) : 0, dmsb = HB ? (J.hb_dmsb ? 16 - J.hb_ddepth : 0) : 0;

    uint32_t cf[NCF];
    {
        const dn_u4 *p = reinterpret_cast<const dn_u4 *>(J.hfv) + (size_t)g * (NCF / 4);
#pragma unroll
        for (int i = 0; i < NCF / 4; i++) {
            const dn_u4 v = p[i];
            cf[4 * i] = v.x; cf[4 * i + 1] = v.y; cf[4 * i + 2] = v.z; cf[4 * i + 3] = v.w;
        }
    }
    const uint32_t par = PAIR ? (J.swap ? 0x00010001u : 0u) : 0u;
    const uint32_t selA = 0x0c040c02u + par, selB = 0x0c050c03u - par; /* pair: channel samples 2 bytes apart */

    const int S = J.steps_per_strip;                 /* a multiple of 4 */
    const int a = strip * S, b = min(a + S, J.dstH); /* this strip's output rows */
    const uint8_t *sbase = J.src + (size_t)frame * J.sfp;
    uint8_t *dr = J.dst + (size_t)frame * J.dfp + (ptrdiff_t)a * J.dstride;
    const ptrdiff_t sstride = J.sstride, dstride = J.dstride;
    const int srcH = J.srcH;
    int pr = 2 * a - 3;                              /* next source row to fetch (unclamped) */
    const uint8_t *pf = sbase + (ptrdiff_t)min(max(pr, 0), srcH - 1) * sstride;
    asm("" : "+s"(pf), "+s"(dr));

    struct Raw { uint32_t q[NQ]; };
    auto load_next = [&](Raw &o) {
        uint32_t off = soff;
        asm volatile("" : "+v"(off)); /* keeps `uniform base + zext(lane offset)` next to the access: saddr addressing */
        const dn_u4 w = *(dn_gc4)((dn_gcp)pf + off);
        o.q[0] = w.x; o.q[1] = w.y; o.q[2] = w.z; o.q[3] = w.w;
        if (HB) {
            const dn_u4 x = *(dn_gc4)((dn_gcp)pf + off + 16);
            o.q[4] = x.x; o.q[5] = x.y; o.q[6] = x.z; o.q[7] = x.w;
            if (PAIR) {
                const dn_u2 e = *(dn_gc2)((dn_gcp)pf + off + 32);
                o.q[8 % NQ] = e.x; o.q[9 % NQ] = e.y;
            }
        } else if (PAIR) {
            const dn_u2 e = *(dn_gc2)((dn_gcp)pf + off + 16);
            o.q[4] = e.x; o.q[5] = e.y;
        }
        pr++;
        pf += (pr >= 1 && pr <= srcH - 1) ? sstride : 0; /* rows above / below the plane replicate the edge row */
        asm("" : "+s"(pf));
    };

    /* horizontal pass of one source row: this lane's 4 samples, >> 7 */
    auto hpass = [&](const Raw &w, int (&h)[4]) {
        uint32_t v[NQ];
#pragma unroll
        for (int i = 0; i < NQ; i++)
            v[i] = w.q[i];
        if (HB) {
            typedef unsigned short dn_h2 __attribute__((ext_vector_type(2)));
            if (border) {
                /* the first / last lane of a row loaded its span two (plane) or three (pair) dwords further inside */
                constexpr int SHIFT = PAIR ? 3 : 2;
                const uint32_t f0 = PAIR ? w.q[0] : __builtin_amdgcn_perm(w.q[0], w.q[0], 0x01000100u);
                const uint32_t fl = PAIR ? w.q[NQ - 1] : __builtin_amdgcn_perm(w.q[NQ - 1], w.q[NQ - 1], 0x03020302u);
#pragma unroll
                for (int i = 0; i < NQ; i++) {
                    const uint32_t vl = i < SHIFT ? f0 : w.q[(i - SHIFT + NQ) % NQ];
                    const uint32_t vr = i + SHIFT < NQ ? w.q[(i + SHIFT) % NQ] : fl;
                    v[i] = lb ? vl : rb ? vr : w.q[i];
                }
            }
            if (smsb) {
#pragma unroll
                for (int i = 0; i < NQ; i++)
                    v[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(dn_h2, v[i]) >> (unsigned short)smsb);
            }
            if constexpr (PAIR != 0) {
                /* columns c0..c9 from 4g - 3: pair m of a channel = (c[2m], c[2m+1]) halves = columns 4g - 3 + 2m, + 1 */
                uint32_t A[5], B[5];
#pragma unroll
                for (int m = 0; m < 5; m++) {
                    A[m] = __builtin_amdgcn_perm(v[(2 * m + 1) % NQ], v[(2 * m) % NQ], 0x05040100u);
                    B[m] = __builtin_amdgcn_perm(v[(2 * m + 1) % NQ], v[(2 * m) % NQ], 0x07060302u);
                }
                dn_h4_pair_s(h, A, B, cf, hsh);
            } else {
                /* dwords (s0,s1) .. (s14,s15) from 8g - 4: pair m = (s[2m+1], s[2m+2]) */
                uint32_t P[7];
#pragma unroll
                for (int m = 0; m < 7; m++)
                    P[m] = __builtin_amdgcn_alignbyte(v[(m + 1) % NQ], v[m % NQ], 2);
                dn_h4_plane_s(h, P, cf, hsh);
            }
            return;
        }
        if (border) {
            if (PAIR) {
                const uint32_t f0 = __builtin_amdgcn_perm(w.q[0], w.q[0], 0x01000100u), f5 = __builtin_amdgcn_perm(w.q[5], w.q[5], 0x03020302u);
                v[0] = lb ? f0 : rb ? w.q[2] : w.q[0];
                v[1] = lb ? f0 : rb ? w.q[3] : w.q[1];
                v[2] = lb ? w.q[0] : rb ? w.q[4] : w.q[2];
                v[3] = lb ? w.q[1] : rb ? w.q[5] : w.q[3];
                v[4] = lb ? w.q[2] : rb ? f5 : w.q[4];
                v[5] = lb ? w.q[3] : rb ? f5 : w.q[5];
            } else {
                const uint32_t f0 = __builtin_amdgcn_perm(w.q[0], w.q[0], 0x00000000u), f3 = __builtin_amdgcn_perm(w.q[3], w.q[3], 0x03030303u);
                v[0] = lb ? f0 : rb ? w.q[1] : w.q[0];
                v[1] = lb ? w.q[0] : rb ? w.q[2] : w.q[1];
                v[2] = lb ? w.q[1] : rb ? w.q[3] : w.q[2];
                v[3] = lb ? w.q[2] : rb ? f3 : w.q[3];
            }
        }
        if constexpr (PAIR != 0) {
            uint32_t A[5], B[5];
#pragma unroll
            for (int m = 0; m < 5; m++) {
                A[m] = __builtin_amdgcn_perm(v[m + 1], v[m], selA);
                B[m] = __builtin_amdgcn_perm(v[m + 1], v[m], selB);
            }
            dn_h4_pair(h, A, B, cf);
        } else {
            uint32_t P[7];
#pragma unroll
            for (int m = 0; m < 3; m++) {
                P[2 * m] = __builtin_amdgcn_perm(v[m + 1], v[m], 0x0c020c01u);
                P[2 * m + 1] = __builtin_amdgcn_perm(v[m + 1], v[m], 0x0c040c03u);
            }
            P[6] = __builtin_amdgcn_perm(v[3], v[3], 0x0c020c01u);
            dn_h4_plane(h, P, cf);
        }
    };
    /* two source rows -> one row pair (int16-saturated: equals min(., 32767) + truncation because no sum of the bank can fall
     * below -32768, host-checked) */
    auto hpair = [&](const Raw &w0, const Raw &w1, uint32_t (&T)[4]) {
        int h0[4], h1[4];
        hpass(w0, h0);
        hpass(w1, h1);
#pragma unroll
        for (int i = 0; i < 4; i++)
            T[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(h0[i], h1[i]));
    };

    if (!HB && J.v1) { /* uniform */
        /* no vertical filter: the chroma planes of a packed-RGB target's first stage when the source has a chroma line per output line
         * (4K 4:2:0 -> 1080p RGB).  The vertical bank is then one tap of 4096 on the row itself and yuv2rgb_X's (U * 4096 + (1 << 18)) >> 19
         * (libswscale/output.c:1814-1835) is (U + 64) >> 7 on the horizontal sum: four rows in flight, a dword of four samples out per row
         * (pair: u0 v0 u1 v1) */
        pr = a;
        pf = sbase + (ptrdiff_t)min(a, srcH - 1) * sstride;
        asm("" : "+s"(pf));
        Raw rb4[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            load_next(rb4[k]);
        for (int y = a; y < b; y += 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (y + k < b) { /* uniform */
                    int h[4];
                    hpass(rb4[k], h);
                    load_next(rb4[k]);
                    uint32_t out, off = doff;
                    asm("v_ashr_pk_u8_i32 %0, %1, %2, 7\n\t"
                        "v_ashr_pk_u8_i32 %0, %3, %4, 7 op_sel:[0,0,0,1]"
                        : "=&v"(out) : "v"(h[0] + 64), "v"(h[1] + 64), "v"(h[2] + 64), "v"(h[3] + 64));
                    asm volatile("" : "+v"(off));
                    if (act)
                        *(dn_g1)((dn_gp)dr + off) = out;
                    dr += dstride;
                    asm("" : "+s"(dr));
                }
            }
        }
        return;
    }

    Raw buf[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        load_next(buf[k]);
    /* rows 2a-3 .. 2a+2: the pairs T(a-1), T(a), T(a+1) -> slots 3, 0, 1 (a % 4 == 0) */
    uint32_t ring[4][4];
    hpair(buf[0], buf[1], ring[3]);
    load_next(buf[0]); load_next(buf[1]);
    hpair(buf[2], buf[3], ring[0]);
    load_next(buf[2]); load_next(buf[3]);
    hpair(buf[0], buf[1], ring[1]);
    load_next(buf[0]); load_next(buf[1]);

    int kround = HB ? 1 << (vsh - 1) : 64 << 12;
    asm volatile("" : "+v"(kround));
    const uint32_t *vt = J.vfv;
    for (int y = a; y < b; y += 4) {
        const dn_u16 c16 = *(dn_cc16)(vt + 4 * y);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            { /* every row of the trip is computed; the store alone looks at the strip's end */
                const bool live = act && y + k < b;
                Raw &w0 = buf[(2 * k + 2) & 3], &w1 = buf[(2 * k + 3) & 3];
                hpair(w0, w1, ring[(k + 2) & 3]);
                load_next(w0); load_next(w1);
                const uint32_t c0 = c16[4 * k], c1 = c16[4 * k + 1], c2 = c16[4 * k + 2], c3 = c16[4 * k + 3];
                uint32_t off = doff;
                if constexpr (HB != 0 && OUT != 0) {
                    /* an 8-bit target (round 5: a 10-bit decoder's 4K frames for a 1080p 8-bit consumer): the ordered dither's entry of
                     * every sample — (x + offset) & 7 with offset 3 for the V channel / plane (yuv2nv12cX_c, vscale.c's chroma call) */
                    const uint2 drow = *reinterpret_cast<const uint2 *>(dn_dither[(y + k) & 7]);
                    int sdv[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int idx = (PAIR ? 2 * g + (i >> 1) + ((i & 1) ? 3 : 0) : 4 * g + i + J.dither_off) & 7;
                        sdv[i] = (int)((((idx >= 4 ? drow.y : drow.x) >> (8 * (idx & 3))) & 255u) << 12);
                    }
                    const uint32_t out = dn_v4d(ring[(k + 3) & 3], ring[k], ring[(k + 1) & 3], ring[(k + 2) & 3], c0, c1, c2, c3, sdv);
                    uint32_t off8 = 4u * (uint32_t)g;
                    asm volatile("" : "+v"(off8));
                    if (live)
                        *(dn_g1)((dn_gp)dr + off8) = out;
                } else if constexpr (HB != 0) {
                    typedef unsigned short dn_h2 __attribute__((ext_vector_type(2)));
                    typedef dn_u2 __attribute__((address_space(1))) *dn_g2;
                    uint32_t o0, o1;
                    dn_v4h(o0, o1, ring[(k + 3) & 3], ring[k], ring[(k + 1) & 3], ring[(k + 2) & 3], c0, c1, c2, c3, kround, vsh, maxpk);
                    o0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(dn_h2, o0) << (unsigned short)dmsb);
                    o1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(dn_h2, o1) << (unsigned short)dmsb);
                    asm volatile("" : "+v"(off));
                    if (live) {
                        dn_u2 st; st.x = o0; st.y = o1;
                        *(dn_g2)((dn_gp)dr + off) = st;
                    }
                } else if (!PAIR && OUT && J.y16) { /* uniform */
                    typedef dn_u2 __attribute__((address_space(1))) *dn_g2y;
                    uint32_t o0, o1;
                    dn_v4y(o0, o1, ring[(k + 3) & 3], ring[k], ring[(k + 1) & 3], ring[(k + 2) & 3], c0, c1, c2, c3, kround);
                    off *= 2; /* four int16 per lane */
                    asm volatile("" : "+v"(off));
                    if (live) {
                        dn_u2 st; st.x = o0; st.y = o1;
                        *(dn_g2y)((dn_gp)dr + off) = st;
                    }
                } else {
                    const uint32_t out = dn_v4(ring[(k + 3) & 3], ring[k], ring[(k + 1) & 3], ring[(k + 2) & 3], c0, c1, c2, c3, kround);
                    asm volatile("" : "+v"(off));
                    if (live)
                        *(dn_g1)((dn_gp)dr + off) = out;
                }
                dr += dstride;
                asm("" : "+s"(dr));
            }
        }
    }
}

/* the vertical sums of one output row, >> 19, NOT clipped, as four ints: the luma of a packed-RGB target (yuv2rgb_X reads it so) */
__device__ __forceinline__ void dn_v4i(int (&t)[4], const uint32_t (&R0)[4], const uint32_t (&R1)[4], const uint32_t (&R2)[4], const uint32_t (&R3)[4],
                                       uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int kround)
{
    asm("v_dot2_i32_i16 %0, %4, %20, %24\n\t"
        "v_dot2_i32_i16 %1, %5, %20, %24\n\t"
        "v_dot2_i32_i16 %2, %6, %20, %24\n\t"
        "v_dot2_i32_i16 %3, %7, %20, %24\n\t"
        "v_dot2_i32_i16 %0, %8, %21, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %21, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %21, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %21, %3\n\t"
        "v_dot2_i32_i16 %0, %12, %22, %0\n\t"
        "v_dot2_i32_i16 %1, %13, %22, %1\n\t"
        "v_dot2_i32_i16 %2, %14, %22, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %22, %3\n\t"
        "v_dot2_i32_i16 %0, %16, %23, %0\n\t"
        "v_dot2_i32_i16 %1, %17, %23, %1\n\t"
        "v_dot2_i32_i16 %2, %18, %23, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %23, %3\n\t"
        "v_ashrrev_i32 %0, 19, %0\n\t"
        "v_ashrrev_i32 %1, 19, %1\n\t"
        "v_ashrrev_i32 %2, 19, %2\n\t"
        "v_ashrrev_i32 %3, 19, %3"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
        : "v"(R0[0]), "v"(R0[1]), "v"(R0[2]), "v"(R0[3]), "v"(R1[0]), "v"(R1[1]), "v"(R1[2]), "v"(R1[3]),
          "v"(R2[0]), "v"(R2[1]), "v"(R2[2]), "v"(R2[3]), "v"(R3[0]), "v"(R3[1]), "v"(R3[2]), "v"(R3[3]),
          "s"(c0), "s"(c1), "s"(c2), "s"(c3), "v"(kround));
}
__device__ __forceinline__ int dn_mad24(int a, int b, int c) /* b: uniform */
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
/* one dword of four bytes clip_u8(x >> 16) */
__device__ __forceinline__ uint32_t dn_pk4_16(int a, int b, int c, int d)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 16 op_sel:[0,0,0,1]"
        : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}

/*
 * k_sws_down2_rgb — exact 2:1 from NV12 / NV21 into packed RGB, FUSED (round 5, the last step): the two-stage form (R5.7) moves 1.65x the
 * algorithmic bytes and both of its kernels stream, so what is left is the intermediate itself.  A lane's four luma outputs and its two
 * chroma pairs cover the SAME four pixels (a pair job's group is two chroma columns = four pixels), so one lane has all of a pixel quad:
 * the luma walk of dn2_unit (static vertical schedule, ring of four row pairs), one chroma row per output row through the horizontal pass
 * alone ((U + 64) >> 7: the vertical chroma bank is one tap), the tables' closed form with the chroma terms from an LDS table
 * (sws_y16rgb.hip's), 12 or 16 bytes per lane and row — lanes side by side, a row segment of 768 / 1024 contiguous bytes per store
 * instruction.
 */
template <int LAY, bool PL> /* PL: planar chroma (yuv420p): csrc / csrc2 are the U and V planes, 12 bytes of each per lane and row */
__device__ __forceinline__ void dn2rgb_unit(const FFHipDn2RgbArgs &J, int frame, int gbase, int strip, int lane, const uint2 *lut)
{
    constexpr int BPG = LAY < 2 ? 12 : 16; /* destination bytes per group of four pixels */
    const int graw = gbase + lane;
    const bool act = graw < J.ngroups;
    const int g = min(graw, J.ngroups - 1);
    const bool lb = g == 0, rb = g == J.ngroups - 1;
    const bool border = gbase == 0 || gbase + 64 >= J.ngroups; /* wave-uniform */
    const uint32_t soff = (uint32_t)(lb ? 0 : 8 * g - 4 - (rb ? 4 : 0));
    const uint32_t coff = PL ? (uint32_t)(lb ? 0 : 4 * g - 4 - (rb ? 4 : 0)) : (uint32_t)(lb ? 0 : 8 * g - 8 - (rb ? 8 : 0));
    uint32_t cf[16], cc[8];
    {
        const dn_u4 *p = reinterpret_cast<const dn_u4 *>(J.hfv_l) + (size_t)g * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const dn_u4 v = p[i];
            cf[4 * i] = v.x; cf[4 * i + 1] = v.y; cf[4 * i + 2] = v.z; cf[4 * i + 3] = v.w;
        }
        const dn_u4 *q = reinterpret_cast<const dn_u4 *>(J.hfv_c) + (size_t)g * 2;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const dn_u4 v = q[i];
            cc[4 * i] = v.x; cc[4 * i + 1] = v.y; cc[4 * i + 2] = v.z; cc[4 * i + 3] = v.w;
        }
    }
    const uint32_t par = J.swap ? 0x00010001u : 0u;
    const uint32_t selA = 0x0c040c02u + par, selB = 0x0c050c03u - par; /* channel samples 2 bytes apart */

    const int S = J.steps_per_strip; /* a multiple of 4 */
    const int a = strip * S, b = min(a + S, J.dstH);
    const uint8_t *sbase = J.ysrc + (size_t)frame * J.ysfp;
    const uint8_t *cbase = J.csrc + (size_t)frame * J.csfp;
    const ptrdiff_t c2 = PL ? J.csrc2 - J.csrc : 0; /* the V plane from the U plane: same stride and frame pitch (the caller sees to it) */
    uint8_t *dr = J.dst + (size_t)frame * J.dfp + (ptrdiff_t)a * J.dstride;
    const ptrdiff_t sstride = J.ysstride, cstride = J.csstride, dstride = J.dstride;
    const int srcH = J.srcH, chrH = J.chrH;
    int pr = 2 * a - 3; /* next luma row to fetch (unclamped) */
    const uint8_t *pf = sbase + (ptrdiff_t)min(max(pr, 0), srcH - 1) * sstride;
    int cr = a;         /* next chroma row to fetch */
    const uint8_t *pc = cbase + (ptrdiff_t)min(a, chrH - 1) * cstride;
    asm("" : "+s"(pf), "+s"(pc), "+s"(dr));

    struct Raw { uint32_t q[4]; };
    struct RawC { uint32_t q[6]; };
    auto load_next = [&](Raw &o) {
        uint32_t off = soff;
        asm volatile("" : "+v"(off));
        const dn_u4 w = *(dn_gc4)((dn_gcp)pf + off);
        o.q[0] = w.x; o.q[1] = w.y; o.q[2] = w.z; o.q[3] = w.w;
        pr++;
        pf += (pr >= 1 && pr <= srcH - 1) ? sstride : 0;
        asm("" : "+s"(pf));
    };
    auto load_chroma = [&](RawC &o) {
        uint32_t off = coff;
        asm volatile("" : "+v"(off));
        if (PL) {
            typedef uint32_t dn_u3 __attribute__((ext_vector_type(3)));
            typedef dn_u3 __attribute__((aligned(4))) dn_u3a;
            typedef const dn_u3a __attribute__((address_space(1))) *dn_gc3;
            const dn_u3 u = *(dn_gc3)((dn_gcp)pc + off), v = *(dn_gc3)((dn_gcp)(pc + c2) + off);
            o.q[0] = u.x; o.q[1] = u.y; o.q[2] = u.z; o.q[3] = v.x; o.q[4] = v.y; o.q[5] = v.z;
        } else {
            const dn_u4 w = *(dn_gc4)((dn_gcp)pc + off);
            const dn_u2 e = *(dn_gc2)((dn_gcp)pc + off + 16);
            o.q[0] = w.x; o.q[1] = w.y; o.q[2] = w.z; o.q[3] = w.w; o.q[4] = e.x; o.q[5] = e.y;
        }
        cr++;
        pc += cr <= chrH - 1 ? cstride : 0;
        asm("" : "+s"(pc));
    };
    auto hpass = [&](const Raw &w, int (&h)[4]) {
        uint32_t v[4] = { w.q[0], w.q[1], w.q[2], w.q[3] };
        if (border) {
            const uint32_t f0 = __builtin_amdgcn_perm(w.q[0], w.q[0], 0x00000000u), f3 = __builtin_amdgcn_perm(w.q[3], w.q[3], 0x03030303u);
            v[0] = lb ? f0 : rb ? w.q[1] : w.q[0];
            v[1] = lb ? w.q[0] : rb ? w.q[2] : w.q[1];
            v[2] = lb ? w.q[1] : rb ? w.q[3] : w.q[2];
            v[3] = lb ? w.q[2] : rb ? f3 : w.q[3];
        }
        uint32_t P[7];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            P[2 * m] = __builtin_amdgcn_perm(v[m + 1], v[m], 0x0c020c01u);
            P[2 * m + 1] = __builtin_amdgcn_perm(v[m + 1], v[m], 0x0c040c03u);
        }
        P[6] = __builtin_amdgcn_perm(v[3], v[3], 0x0c020c01u);
        dn_h4_plane(h, P, cf);
    };
    auto hpass_c = [&](const RawC &w, int (&h)[4]) {
        uint32_t A[5], B[5];
        if (PL) {
            /* a plane's samples 4g - 4 .. 4g + 7 in three dwords: pair m = (s[2m + 1], s[2m + 2]), as the luma's */
            uint32_t x[2][3] = { { w.q[0], w.q[1], w.q[2] }, { w.q[3], w.q[4], w.q[5] } };
            if (border) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const uint32_t q0 = w.q[3 * c], q1 = w.q[3 * c + 1], q2 = w.q[3 * c + 2];
                    const uint32_t f0 = __builtin_amdgcn_perm(q0, q0, 0x00000000u), f2 = __builtin_amdgcn_perm(q2, q2, 0x03030303u);
                    x[c][0] = lb ? f0 : rb ? q1 : q0;
                    x[c][1] = lb ? q0 : rb ? q2 : q1;
                    x[c][2] = lb ? q1 : rb ? f2 : q2;
                }
            }
#pragma unroll
            for (int c = 0; c < 2; c++) {
                uint32_t *P = c ? B : A;
                P[0] = __builtin_amdgcn_perm(x[c][1], x[c][0], 0x0c020c01u);
                P[1] = __builtin_amdgcn_perm(x[c][1], x[c][0], 0x0c040c03u);
                P[2] = __builtin_amdgcn_perm(x[c][2], x[c][1], 0x0c020c01u);
                P[3] = __builtin_amdgcn_perm(x[c][2], x[c][1], 0x0c040c03u);
                P[4] = __builtin_amdgcn_perm(x[c][2], x[c][2], 0x0c020c01u);
            }
        } else {
            uint32_t v[6] = { w.q[0], w.q[1], w.q[2], w.q[3], w.q[4], w.q[5] };
            if (border) {
                const uint32_t f0 = __builtin_amdgcn_perm(w.q[0], w.q[0], 0x01000100u), f5 = __builtin_amdgcn_perm(w.q[5], w.q[5], 0x03020302u);
                v[0] = lb ? f0 : rb ? w.q[2] : w.q[0];
                v[1] = lb ? f0 : rb ? w.q[3] : w.q[1];
                v[2] = lb ? w.q[0] : rb ? w.q[4] : w.q[2];
                v[3] = lb ? w.q[1] : rb ? w.q[5] : w.q[3];
                v[4] = lb ? w.q[2] : rb ? f5 : w.q[4];
                v[5] = lb ? w.q[3] : rb ? f5 : w.q[5];
            }
#pragma unroll
            for (int m = 0; m < 5; m++) {
                A[m] = __builtin_amdgcn_perm(v[m + 1], v[m], selA);
                B[m] = __builtin_amdgcn_perm(v[m + 1], v[m], selB);
            }
        }
        dn_h4_pair(h, A, B, cc);
    };
    auto hpair = [&](const Raw &w0, const Raw &w1, uint32_t (&T)[4]) {
        int h0[4], h1[4];
        hpass(w0, h0);
        hpass(w1, h1);
#pragma unroll
        for (int i = 0; i < 4; i++)
            T[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(h0[i], h1[i]));
    };

    Raw buf[4];
    RawC cbuf[2];
#pragma unroll
    for (int k = 0; k < 4; k++)
        load_next(buf[k]);
    load_chroma(cbuf[0]);
    load_chroma(cbuf[1]);
    uint32_t ring[4][4];
    hpair(buf[0], buf[1], ring[3]);
    load_next(buf[0]); load_next(buf[1]);
    hpair(buf[2], buf[3], ring[0]);
    load_next(buf[2]); load_next(buf[3]);
    hpair(buf[0], buf[1], ring[1]);
    load_next(buf[0]); load_next(buf[1]);

    int kround = 64 << 12;
    asm volatile("" : "+v"(kround));
    const int cy = __builtin_amdgcn_readfirstlane(J.k.cy);
    const char *lutb = reinterpret_cast<const char *>(lut);
    const uint32_t *vt = J.vfv;
    const uint32_t doff = (uint32_t)BPG * (uint32_t)g;
    for (int y = a; y < b; y += 4) {
        const dn_u16 c16 = *(dn_cc16)(vt + 4 * y);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (y + k < b) { /* uniform */
                Raw &w0 = buf[(2 * k + 2) & 3], &w1 = buf[(2 * k + 3) & 3];
                hpair(w0, w1, ring[(k + 2) & 3]);
                load_next(w0); load_next(w1);
                int hc[4];
                hpass_c(cbuf[k & 1], hc);
                load_chroma(cbuf[k & 1]);
                int Y[4];
                dn_v4i(Y, ring[(k + 3) & 3], ring[k], ring[(k + 1) & 3], ring[(k + 2) & 3], c16[4 * k], c16[4 * k + 1], c16[4 * k + 2], c16[4 * k + 3], kround);
                /* (u0, v0, u1, v1): clip_u8((h + 64) >> 7), the tables' index */
                int c0[2], c1[2], c2[2];
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    const int ui = min(max((hc[2 * m] + 64) >> 7, 0), 255), vi = min(max((hc[2 * m + 1] + 64) >> 7, 0), 255);
                    const uint2 tu = *reinterpret_cast<const uint2 *>(lutb + (ui << 3));
                    const uint2 tv = *reinterpret_cast<const uint2 *>(lutb + 2048 + (vi << 3));
                    constexpr bool BGR = LAY == 1 || LAY == 4 || LAY == 5;
                    c0[m] = (int)(BGR ? tu.x : tv.x);
                    c1[m] = (int)(tu.y + tv.y);
                    c2[m] = (int)(BGR ? tv.x : tu.x);
                }
                int val[12];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    val[3 * p] = dn_mad24(Y[p], cy, c0[p >> 1]);
                    val[3 * p + 1] = dn_mad24(Y[p], cy, c1[p >> 1]);
                    val[3 * p + 2] = dn_mad24(Y[p], cy, c2[p >> 1]);
                }
                uint32_t off = doff;
                asm volatile("" : "+v"(off));
                if (LAY >= 2) {
                    int alpha = 255 << 16;
                    asm("" : "+v"(alpha));
                    dn_u4 o;
                    uint32_t w[4];
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        w[p] = (LAY == 2 || LAY == 4) ? dn_pk4_16(alpha, val[3 * p], val[3 * p + 1], val[3 * p + 2])
                                                      : dn_pk4_16(val[3 * p], val[3 * p + 1], val[3 * p + 2], alpha);
                    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
                    if (act)
                        __builtin_nontemporal_store(o, (dn_u4a __attribute__((address_space(1))) *)((dn_gp)dr + off));
                } else {
                    typedef uint32_t dn_u3 __attribute__((ext_vector_type(3)));
                    typedef dn_u3 __attribute__((aligned(4))) dn_u3a;
                    dn_u3 o;
                    o.x = dn_pk4_16(val[0], val[1], val[2], val[3]);
                    o.y = dn_pk4_16(val[4], val[5], val[6], val[7]);
                    o.z = dn_pk4_16(val[8], val[9], val[10], val[11]);
                    if (act)
                        __builtin_nontemporal_store(o, (dn_u3a __attribute__((address_space(1))) *)((dn_gp)dr + off));
                }
                dr += dstride;
                asm("" : "+s"(dr));
            }
        }
    }
}

template <int LAY, bool PL>
__global__ __launch_bounds__(256) void k_sws_down2_rgb(FFHipDn2RgbArgs A)
{
    __shared__ uint2 lut[512]; /* [U] = { b(U), gu(U) }, [256 + V] = { r(V), gv(V) }: the chroma terms, cy-scaled, rounding in (sws_y16rgb.hip) */
    {
        const int t = (int)threadIdx.x;
        const FFHipYuv2RgbK Kt = A.k;
        lut[t] = make_uint2((uint32_t)(__mul24(Kt.off_b + (__mul24(t, Kt.cbu) >> 16), Kt.cy) + Kt.kb),
                            (uint32_t)(__mul24(Kt.off_g + (__mul24(t, Kt.cgu) >> 16), Kt.cy) + Kt.kb));
        lut[256 + t] = make_uint2((uint32_t)(__mul24(Kt.off_r + (__mul24(t, Kt.crv) >> 16), Kt.cy) + Kt.kb),
                                  (uint32_t)__mul24(__mul24(t, Kt.cgv) >> 16, Kt.cy));
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    uint32_t blk = blockIdx.x;
    if (A.xcd) {
        const uint32_t nb = gridDim.x, x = blk & 7u, sl = blk >> 3, q = nb >> 3, r = nb & 7u;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + sl;
    }
    const uint32_t gw = blk * 4u + (uint32_t)wave;
    const uint32_t upf = (uint32_t)A.ncb * (uint32_t)A.nstrips;
    if (gw >= upf * (uint32_t)A.nframes)
        return;
    const int frame = (int)(gw / upf);
    const int u = (int)(gw - (uint32_t)frame * upf);
    const int strip = u / A.ncb, cb = u - strip * A.ncb;
    dn2rgb_unit<LAY, PL>(A, frame, cb * 64, strip, lane, lut);
}

template <int HB, int OUT>
__global__ __launch_bounds__(256) void k_sws_down2(FFHipDn2Args A)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    uint32_t blk = blockIdx.x;
    if (A.xcd) {
        /* workgroup b runs on XCD b % 8 (observed, not promised: speed only): every XCD gets one contiguous eighth of the units,
         * so that the waves sharing source lines (adjacent column blocks, the halo rows of adjacent strips) meet in one L2 */
        const uint32_t nb = gridDim.x, x = blk & 7u, sl = blk >> 3, q = nb >> 3, r = nb & 7u;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + sl;
    }
    const uint32_t gw = blk * 4u + (uint32_t)wave;
    if (gw >= (uint32_t)A.units_per_frame * (uint32_t)A.nframes)
        return;
    const int frame = (int)(gw / (uint32_t)A.units_per_frame);
    const int u = (int)(gw - (uint32_t)frame * (uint32_t)A.units_per_frame);
    int j = 0;
    if (A.njobs > 1 && u >= A.job[1].unit_begin) j = 1;
    if (A.njobs > 2 && u >= A.job[2].unit_begin) j = 2;
    const FFHipDn2Job &J = A.job[j];
    const int local = u - J.unit_begin;
    const int strip = local / J.ncb, cb = local - strip * J.ncb;
    if (J.pair)
        dn2_unit<1, HB, OUT>(J, frame, cb * 64, strip, lane);
    else
        dn2_unit<0, HB, OUT>(J, frame, cb * 64, strip, lane);
}

/* ================================================================================================== */
/* host side */

/*
 * Re-express a bank of an exact 2:1 down-scale (at most 8 taps) as coefficients on the REGULAR windows of the edge-replicated
 * row: output x reads samples clamp(2x - 3 + k), k = 0..7.  Every non-zero tap of the bank row must sit on one of those
 * samples; taps the reference folded onto the edge sample land on one of the replicas.  Output: n_dst x 4 dwords,
 * (c0, c1) .. (c6, c7) as int16 pairs.  Returns 0 when the bank is not of this shape.
 */
int ffhip_down2_virtual_bank(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, std::vector<uint32_t> *out)
{
    if (n_src != 2 * n_dst || fsize < 1 || fsize > 16)
        return 0;
    out->assign((size_t)n_dst * 4, 0);
    for (int x = 0; x < n_dst; x++) {
        const int s0 = 2 * x - 3;
        int16_t v[8] = { 0 };
        bool used[8] = { false };
        for (int i = 0; i < fsize; i++) {
            const int16_t c = filter[(size_t)x * fsize + i];
            if (!c)
                continue;
            const int p = pos[x] + i;
            if (p < 0 || p >= n_src)
                return 0;
            int k = 0;
            for (; k < 8; k++) {
                int q = s0 + k;
                q = q < 0 ? 0 : q >= n_src ? n_src - 1 : q;
                if (q == p && !used[k])
                    break;
            }
            if (k == 8)
                return 0;
            used[k] = true;
            v[k] = c;
        }
        for (int k = 0; k < 4; k++)
            (*out)[4 * (size_t)x + k] = (uint16_t)v[2 * k] | ((uint32_t)(uint16_t)v[2 * k + 1] << 16);
    }
    return 1;
}

/* strips of about `want` output rows (a multiple of 4: the row loop is unrolled four times), evened out over the plane */
void ffhip_down2_plan_job(FFHipDn2Job *j, int want)
{
    const int n = cdiv(j->dstH, want);
    j->steps_per_strip = cdiv(cdiv(j->dstH, n), 4) * 4;
    j->nstrips = cdiv(j->dstH, j->steps_per_strip);
    j->ncb = cdiv(j->ngroups, 64);
}

int ffhip_launch_down2(FFHipDn2Args &A, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    int u = 0;
    for (int i = 0; i < A.njobs; i++) {
        A.job[i].unit_begin = u;
        u += A.job[i].ncb * A.job[i].nstrips;
    }
    A.units_per_frame = u;
    const long long waves = (long long)u * A.nframes;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    const dim3 grid((unsigned)((waves + 3) / 4));
    bool y16 = false;
    for (int i = 0; i < A.njobs; i++)
        y16 = y16 || A.job[i].y16;
    if (A.job[0].hb_sdepth && A.job[0].hb_ddepth == 8)
        hipLaunchKernelGGL((k_sws_down2<1, 1>), grid, dim3(256), 0, stream, A);
    else if (A.job[0].hb_sdepth)
        hipLaunchKernelGGL((k_sws_down2<1, 0>), grid, dim3(256), 0, stream, A);
    else if (y16)
        hipLaunchKernelGGL((k_sws_down2<0, 1>), grid, dim3(256), 0, stream, A);
    else
        hipLaunchKernelGGL((k_sws_down2<0, 0>), grid, dim3(256), 0, stream, A);
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_down2_rgb(FFHipDn2RgbArgs &A, int want_rows, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    if (A.ngroups < 3 || A.dstH <= 0) {
        ffhip_set_error("ffhip_sws: the fused exact-2:1 RGB kernel takes widths from 12 (got %d groups)", A.ngroups);
        return FFHIP_EINVAL;
    }
    const int n = cdiv(A.dstH, want_rows);
    A.steps_per_strip = cdiv(cdiv(A.dstH, n), 4) * 4;
    A.nstrips = cdiv(A.dstH, A.steps_per_strip);
    A.ncb = cdiv(A.ngroups, 64);
    const long long waves = (long long)A.ncb * A.nstrips * A.nframes;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define DR_L(L)                                                                                        \
    case L:                                                                                            \
        if (A.csrc2) hipLaunchKernelGGL((k_sws_down2_rgb<L, true>), grid, block, 0, stream, A);        \
        else hipLaunchKernelGGL((k_sws_down2_rgb<L, false>), grid, block, 0, stream, A);               \
        break;
    switch (A.lay) {
    DR_L(0) DR_L(1) DR_L(2) DR_L(3) DR_L(4) DR_L(5)
    default:
        ffhip_set_error("ffhip_sws: packed layout %d is not one of the RGB writer's", A.lay);
        return FFHIP_EINVAL;
    }
#undef DR_L
    LAUNCH_CHECK();
    return 0;
}
