/*
 * sws_up32.hip — the fused H+V scaler for EXACT 3:2 up-scaling (720p -> 1080p, 1080p -> 1620p, 1440p -> 4K ...) of 9..14-bit samples
 * with 4-tap banks in both directions (bicubic, bilinear, point), planes and interleaved (u, v) planes of words (P010 / P012) in and
 * out (round 6).  The commonest up-scale that is not 2x ran on the 16-bit column walker (sws_walk16.hip: 0.35 of HBM).
 *
 * Arithmetic (the reference's, bit for bit — the same as sws_walk16.hip / k_sws_up2<., ., 1>, tests/test_gpu_sws_hbd.py):
 *   hScale16To15_c        libswscale/swscale.c:99-126    val = sum src[pos + j] * filter[j];  dst = FFMIN(val >> (depth - 1), 32767)
 *   yuv2planeX_10_c       libswscale/output.c:341-360    val = (1 << (26 - bits)) + sum line[j][i] * filter[j];  av_clip_uintp2(val >> (27 - bits), bits)
 *   yuv2p01xlX / cX       libswscale/output.c:478-529    the same, stored << (16 - bits)
 *
 * At exactly 3:2 everything is regular with period (2 in, 3 out):
 *  - output x = 3k + j reads source samples 2k - 2 + j .. 2k + 1 + j (initFilter, libswscale/utils.c:519-561, folds the taps that fall
 *    outside the row onto the edge sample: the regular bank over an edge-REPLICATED row with its own coefficients next to either edge —
 *    ffhip_u32_virtual_bank re-expresses every bank row that way, tap by tap, else this kernel is not used).  A lane owns 6 outputs
 *    = 2 periods = 4 source samples: 16 source bytes at the dword-aligned offset 8g - 4; two samples in a dword ARE the (s[k], s[k+1])
 *    operand of v_dot2_i32_i16 for the windows that start on an even sample (j = 0, 2), one v_alignbit away for j = 1 (three per row).
 *    An interleaved pair: 3 columns x 2 channels per lane from 6 (u, v) dwords, the channels' pairs split with v_perm_b32.  (The bank is
 *    NOT periodic — 2/3 is not a 16.16 number, initFilter's positions drift — so every column keeps its own coefficients, 12 VGPRs; the
 *    first version had 12 outputs per lane: 136 VGPRs, three waves per SIMD, 0.35 of HBM — no faster than the walker.)
 *  - the vertical schedule is static with period (2 source rows, 3 output rows): with P(q) = (row q, row q + 1) of horizontally
 *    filtered samples in a ring of three, source row r completes P(r - 1) and every output that ends on r reads P(r - 3), P(r - 1):
 *    an odd r = 2m + 1 is due rows 3m - 1 and 3m, an even r = 2m + 2 row 3m + 1.  Six source rows per loop trip make every ring index a
 *    constant; the rows' coefficient pairs are wave-uniform (scalar loads).
 * Dots in hand-scheduled blocks of four chains (VOP3P form: no v_mov per chain).  Algorithmic bytes: source in + destination out, 2 bytes per sample: 2.89 B per output sample.
 */
#include <type_traits>

#include "common.h"
#include "sws_kernels.h"

typedef short u3_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u3_h2 __attribute__((ext_vector_type(2)));
typedef uint32_t u3_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u3_u4 __attribute__((ext_vector_type(4)));
typedef u3_u4 __attribute__((aligned(4))) u3_u4a;
typedef u3_u2 __attribute__((aligned(4))) u3_u2a;
typedef uint32_t u3_u3 __attribute__((ext_vector_type(3)));
typedef u3_u3 __attribute__((aligned(4))) u3_u3a;
typedef const uint8_t __attribute__((address_space(1))) *u3_gcp;
typedef uint8_t __attribute__((address_space(1))) *u3_gp;
typedef const u3_u4a __attribute__((address_space(1))) *u3_gc4;
typedef const u3_u2a __attribute__((address_space(1))) *u3_gc2;
typedef u3_u4a __attribute__((address_space(1))) *u3_g4;
typedef u3_u2a __attribute__((address_space(1))) *u3_g2;
typedef const uint32_t __attribute__((address_space(1))) *u3_gc1;
typedef u3_u3a __attribute__((address_space(1))) *u3_g3;
typedef const uint32_t __attribute__((address_space(4))) *u3_cc; /* constant address space: scalar loads */

__device__ __forceinline__ int u3_dot(uint32_t p, uint32_t c, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(u3_s2, p), __builtin_bit_cast(u3_s2, c), acc, false);
}


/*
 * Hand-scheduled dot products, as in sws_up2.hip / sws_walk16.hip: hipcc selects the accumulate-in-place VOP2 form v_dot2c_i32_i16 for the
 * builtin, which costs a v_mov per chain to seed the accumulator (825 v_mov for 720 dots in this kernel's first version); the VOP3P form
 * takes the seed as a third source.  Hazards inside an asm block are ours (gfx950: a DOT result may feed the same opcode as src2 at
 * once, any other VALU only after 3 wait states): six chains are interleaved, every result is shifted five instructions after its
 * last DOT and leaves the block as the result of a plain VALU instruction.
 */
/* six horizontal samples: d[i] = (pa[i] . ca[i] + pb[i] . cb[i]) >> sh; cf = (ca, cb) per sample */
__device__ __forceinline__ void u3_h6(int (&d)[6], const uint32_t (&pa)[6], const uint32_t (&pb)[6], const uint32_t (&cf)[6][2], int sh)
{
    asm("v_dot2_i32_i16 %0, %6, %18, 0\n\t"
        "v_dot2_i32_i16 %1, %7, %19, 0\n\t"
        "v_dot2_i32_i16 %2, %8, %20, 0\n\t"
        "v_dot2_i32_i16 %3, %9, %21, 0\n\t"
        "v_dot2_i32_i16 %4, %10, %22, 0\n\t"
        "v_dot2_i32_i16 %5, %11, %23, 0\n\t"
        "v_dot2_i32_i16 %0, %12, %24, %0\n\t"
        "v_dot2_i32_i16 %1, %13, %25, %1\n\t"
        "v_dot2_i32_i16 %2, %14, %26, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %27, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %28, %4\n\t"
        "v_dot2_i32_i16 %5, %17, %29, %5\n\t"
        "v_ashrrev_i32 %0, %30, %0\n\t"
        "v_ashrrev_i32 %1, %30, %1\n\t"
        "v_ashrrev_i32 %2, %30, %2\n\t"
        "v_ashrrev_i32 %3, %30, %3\n\t"
        "v_ashrrev_i32 %4, %30, %4\n\t"
        "v_ashrrev_i32 %5, %30, %5"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pb[4]), "v"(pb[5]),
          "v"(cf[0][0]), "v"(cf[1][0]), "v"(cf[2][0]), "v"(cf[3][0]), "v"(cf[4][0]), "v"(cf[5][0]),
          "v"(cf[0][1]), "v"(cf[1][1]), "v"(cf[2][1]), "v"(cf[3][1]), "v"(cf[4][1]), "v"(cf[5][1]), "s"(sh));
}
/* six output samples as three dwords: t[i] = seed + pa[i] . f01 + pb[i] . f23, clipped to 0 .. 2^depth - 1 after >> sh (v_cvt_pk_i16_i32
 * saturates to int16: the clip range lies inside), << msb (P01x; the shift in both halves of `msb`) */
__device__ __forceinline__ void u3_v6(uint32_t (&w)[3], const uint32_t (&pa)[6], const uint32_t (&pb)[6], uint32_t f01, uint32_t f23, int seed, int sh,
                                      uint32_t maxpk, uint32_t msb)
{
    int t0, t1, t2, t3, t4, t5;
    asm("v_dot2_i32_i16 %3, %9, %21, %23\n\t"
        "v_dot2_i32_i16 %4, %10, %21, %23\n\t"
        "v_dot2_i32_i16 %5, %11, %21, %23\n\t"
        "v_dot2_i32_i16 %6, %12, %21, %23\n\t"
        "v_dot2_i32_i16 %7, %13, %21, %23\n\t"
        "v_dot2_i32_i16 %8, %14, %21, %23\n\t"
        "v_dot2_i32_i16 %3, %15, %22, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %22, %4\n\t"
        "v_dot2_i32_i16 %5, %17, %22, %5\n\t"
        "v_dot2_i32_i16 %6, %18, %22, %6\n\t"
        "v_dot2_i32_i16 %7, %19, %22, %7\n\t"
        "v_dot2_i32_i16 %8, %20, %22, %8\n\t"
        "v_ashrrev_i32 %3, %24, %3\n\t"
        "v_ashrrev_i32 %4, %24, %4\n\t"
        "v_ashrrev_i32 %5, %24, %5\n\t"
        "v_ashrrev_i32 %6, %24, %6\n\t"
        "v_ashrrev_i32 %7, %24, %7\n\t"
        "v_ashrrev_i32 %8, %24, %8\n\t"
        "v_cvt_pk_i16_i32 %0, %3, %4\n\t"
        "v_cvt_pk_i16_i32 %1, %5, %6\n\t"
        "v_cvt_pk_i16_i32 %2, %7, %8\n\t"
        "v_pk_max_i16 %0, %0, 0\n\t"
        "v_pk_max_i16 %1, %1, 0\n\t"
        "v_pk_max_i16 %2, %2, 0\n\t"
        "v_pk_min_i16 %0, %0, %25\n\t"
        "v_pk_min_i16 %1, %1, %25\n\t"
        "v_pk_min_i16 %2, %2, %25\n\t"
        "v_pk_lshlrev_b16 %0, %26, %0\n\t"
        "v_pk_lshlrev_b16 %1, %26, %1\n\t"
        "v_pk_lshlrev_b16 %2, %26, %2"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pb[4]), "v"(pb[5]),
          "s"(f01), "s"(f23), "v"(seed), "s"(sh), "s"(maxpk), "s"(msb));
}


/* the same blocks with four chains (eight outputs per lane = two of them): every result is shifted three instructions after its last DOT */
__device__ __forceinline__ void u3_h4(int *d, const uint32_t *pa, const uint32_t *pb, const uint32_t (*cf)[2], int sh)
{
    asm("v_dot2_i32_i16 %0, %4, %12, 0\n\t"
        "v_dot2_i32_i16 %1, %5, %13, 0\n\t"
        "v_dot2_i32_i16 %2, %6, %14, 0\n\t"
        "v_dot2_i32_i16 %3, %7, %15, 0\n\t"
        "v_dot2_i32_i16 %0, %8, %16, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %17, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %18, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %19, %3\n\t"
        "v_ashrrev_i32 %0, %20, %0\n\t"
        "v_ashrrev_i32 %1, %20, %1\n\t"
        "v_ashrrev_i32 %2, %20, %2\n\t"
        "v_ashrrev_i32 %3, %20, %3"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(cf[0][0]), "v"(cf[1][0]), "v"(cf[2][0]),
          "v"(cf[3][0]), "v"(cf[0][1]), "v"(cf[1][1]), "v"(cf[2][1]), "v"(cf[3][1]), "s"(sh));
}
__device__ __forceinline__ void u3_v4(uint32_t *w, const uint32_t *pa, const uint32_t *pb, uint32_t f01, uint32_t f23, int seed, int sh, uint32_t maxpk,
                                      uint32_t msb)
{
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %2, %6, %14, %16\n\t"
        "v_dot2_i32_i16 %3, %7, %14, %16\n\t"
        "v_dot2_i32_i16 %4, %8, %14, %16\n\t"
        "v_dot2_i32_i16 %5, %9, %14, %16\n\t"
        "v_dot2_i32_i16 %2, %10, %15, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %15, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %15, %4\n\t"
        "v_dot2_i32_i16 %5, %13, %15, %5\n\t"
        "v_ashrrev_i32 %2, %17, %2\n\t"
        "v_ashrrev_i32 %3, %17, %3\n\t"
        "v_ashrrev_i32 %4, %17, %4\n\t"
        "v_ashrrev_i32 %5, %17, %5\n\t"
        "v_cvt_pk_i16_i32 %0, %2, %3\n\t"
        "v_cvt_pk_i16_i32 %1, %4, %5\n\t"
        "v_pk_max_i16 %0, %0, 0\n\t"
        "v_pk_max_i16 %1, %1, 0\n\t"
        "v_pk_min_i16 %0, %0, %18\n\t"
        "v_pk_min_i16 %1, %1, %18\n\t"
        "v_pk_lshlrev_b16 %0, %19, %0\n\t"
        "v_pk_lshlrev_b16 %1, %19, %1"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "s"(f01), "s"(f23), "v"(seed), "s"(sh),
          "s"(maxpk), "s"(msb));
}

/* where the window of output j of a period starts, from the period's first source sample: floor(((2j + 1) PIN - POUT) / (2 POUT)) - 1 */
template <int PIN, int POUT>
__host__ __device__ constexpr int u3_off(int j)
{
    const int n = (2 * j + 1) * PIN - POUT;
    return (n >= 0 ? n / (2 * POUT) : -((-n + 2 * POUT - 1) / (2 * POUT))) - 1;
}

template <int PIN, int POUT, int PAIR>
__device__ __forceinline__ void u32_unit(const FFHipU32Args &A, const FFHipU32Job &J, int frame, int gbase, int strip, int lane)
{
    /* a lane owns NO = 2 POUT outputs: two periods of a plane (2 PIN source samples + 4 of halo = PIN + 2 dwords at byte 4 PIN g - 4), one
     * period x two channels of a pair (PIN columns + 4 of halo = PIN + 4 dwords at byte 4 PIN g - 8) */
    constexpr int NO = 2 * POUT, NW = PAIR ? PIN + 4 : PIN + 2;
    const int graw = gbase + lane;
    const bool act = graw < J.ngroups;
    const int g = min(graw, J.ngroups - 1);
    const bool lb = g == 0, rb = g == J.ngroups - 1;
    const bool border = gbase == 0 || gbase + 64 >= J.ngroups; /* wave-uniform */
    /* the first / last lane of a row loads one dword (pair: two) further inside and rebuilds the replicated ones */
    const uint32_t soff = PAIR ? (uint32_t)(lb ? 0 : 4 * PIN * g - 8 - (rb ? 8 : 0)) : (uint32_t)(lb ? 0 : 4 * PIN * g - 4 - (rb ? 4 : 0));
    const int hsh = A.sdepth - 1, smsb = A.smsb ? 16 - A.sdepth : 0;
    const uint32_t dmsb = (uint32_t)(A.dmsb ? 16 - A.ddepth : 0) * 0x00010001u;
    const int vsh = 27 - A.ddepth, vseed = 1 << (26 - A.ddepth);
    const uint32_t maxpk = (uint32_t)((1 << A.ddepth) - 1) * 0x00010001u;
    uint32_t cf[NO][2];
    {
        /* plane: outputs NO g .. NO g + NO - 1; pair: columns POUT g .. + POUT - 1, both channels of a column share its coefficients */
        const uint32_t *p = J.hfv + (size_t)g * (PAIR ? NO : 2 * NO);
#pragma unroll
        for (int j = 0; j < NO; j++)
#pragma unroll
            for (int k = 0; k < 2; k++)
                cf[j][k] = p[2 * (PAIR ? j >> 1 : j) + k];
    }
    const int a = strip * J.strip_rows, b = min(a + J.strip_rows, J.dstH); /* this strip's output rows; a is a multiple of 3 POUT */
    const uint8_t *sbase = J.src + (size_t)frame * J.sfp;
    uint8_t *dbase = J.dst + (size_t)frame * J.dfp;
    const ptrdiff_t sstride = J.sstride, dstride = J.dstride;
    const int srcH = J.srcH;

    auto load_row = [&](int r, uint32_t (&w)[NW]) {
        const uint8_t *p = sbase + (ptrdiff_t)min(max(r, 0), srcH - 1) * sstride; /* rows above / below the plane replicate the edge row */
        const u3_u4 v = *(u3_gc4)((u3_gcp)p + soff);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        if (NW == 5) {
            w[NW - 1] = *(u3_gc1)((u3_gcp)p + soff + 16);
        } else if (NW == 6) {
            const u3_u2 e = *(u3_gc2)((u3_gcp)p + soff + 16);
            w[NW - 2] = e.x; w[NW - 1] = e.y;
        } else if (NW == 7) {
            const u3_u2 e = *(u3_gc2)((u3_gcp)p + soff + 16);
            w[4] = e.x; w[5 % NW] = e.y;
            w[NW - 1] = *(u3_gc1)((u3_gcp)p + soff + 24);
        }
    };
    /* the horizontal pass of one source row: this lane's NO samples (plane: x0 .. ; pair: u0 v0 u1 v1 ..) */
    auto hpass = [&](const uint32_t (&raw)[NW], int (&h)[NO]) {
        uint32_t w[NW];
#pragma unroll
        for (int i = 0; i < NW; i++)
            w[i] = raw[i];
        if (border) {
            if (PAIR) {
#pragma unroll
                for (int i = 0; i < NW; i++)
                    w[i] = lb ? raw[i < 2 ? 0 : i - 2] : rb ? raw[i + 2 < NW ? i + 2 : NW - 1] : raw[i];
            } else {
                const uint32_t f0 = __builtin_amdgcn_perm(raw[0], raw[0], 0x01000100u), fl = __builtin_amdgcn_perm(raw[NW - 1], raw[NW - 1], 0x03020302u);
#pragma unroll
                for (int i = 0; i < NW; i++)
                    w[i] = lb ? (i ? raw[i - 1] : f0) : rb ? (i + 1 < NW ? raw[i + 1] : fl) : raw[i];
            }
        }
        if (smsb) { /* uniform: P01x keeps its samples in the high bits */
#pragma unroll
            for (int i = 0; i < NW; i++)
                w[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u3_h2, w[i]) >> (unsigned short)smsb);
        }
        uint32_t pa[NO], pb[NO];
        if (PAIR) {
            /* column j of the lane's period reads columns off(j) + 2 .. + 3 more from the lane's base: the channels' pairs (c, c + 1), (c + 2, c + 3) */
#pragma unroll
            for (int j = 0; j < POUT; j++) {
                constexpr int dummy = 0; (void)dummy;
                const int c0 = u3_off<PIN, POUT>(j) + 2;
                pa[2 * j] = __builtin_amdgcn_perm(w[c0 + 1], w[c0], 0x05040100u);
                pa[2 * j + 1] = __builtin_amdgcn_perm(w[c0 + 1], w[c0], 0x07060302u);
                pb[2 * j] = __builtin_amdgcn_perm(w[c0 + 3], w[c0 + 2], 0x05040100u);
                pb[2 * j + 1] = __builtin_amdgcn_perm(w[c0 + 3], w[c0 + 2], 0x07060302u);
            }
        } else {
            /* output i = POUT k + j of the lane reads samples s .. s + 3, s = PIN k + off(j) + 2 from the lane's base: an even s takes the
             * dwords w[s / 2], w[s / 2 + 1], an odd one the pairs that start on an odd sample */
            uint32_t o[NW - 1];
#pragma unroll
            for (int k = 0; k < NW - 1; k++)
                o[k] = __builtin_amdgcn_alignbit(w[k + 1], w[k], 16);
#pragma unroll
            for (int i = 0; i < NO; i++) {
                const int s0 = PIN * (i / POUT) + u3_off<PIN, POUT>(i % POUT) + 2;
                pa[i] = (s0 & 1) ? o[s0 >> 1] : w[s0 >> 1];
                pb[i] = (s0 & 1) ? o[(s0 >> 1) + 1] : w[(s0 >> 1) + 1];
            }
        }
        if (NO == 6) {
            u3_h6(reinterpret_cast<int (&)[6]>(h), reinterpret_cast<const uint32_t (&)[6]>(pa), reinterpret_cast<const uint32_t (&)[6]>(pb),
                  reinterpret_cast<const uint32_t (&)[6][2]>(cf), hsh);
        } else {
#pragma unroll
            for (int q = 0; q < NO / 4; q++)
                u3_h4(h + 4 * q, pa + 4 * q, pb + 4 * q, cf + 4 * q, hsh);
        }
    };

    /* source rows in trips of T = lcm(3, PIN) from rbase = PIN (a / POUT) - T (a multiple of T); its last two rows only fill the ring.
     * Straight-line code: every due row is computed, the store alone looks at the strip's bounds (branches around the arithmetic made the
     * compiler copy the rows in flight at every join — and wait for them); the walk ends at uniform exits after the row the strip's last
     * output ends on.  Output y = POUT m + j ends on source row PIN m + off(j) + 3 and reads P(that - 3), P(that - 1). */
    constexpr int T = PIN % 3 ? 3 * PIN : PIN;
    const int rbase0 = PIN * (a / POUT) - T, r_last = PIN * (b / POUT - 1) + u3_off<PIN, POUT>(POUT - 1) + 3, dstH = J.dstH;
    uint32_t ring[3][NO];
    int hprev[NO];
#pragma unroll
    for (int c = 0; c < NO; c++)
        hprev[c] = 0;
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int c = 0; c < NO; c++)
            ring[s][c] = 0;
    constexpr int D = 3; /* source rows in flight */
    uint32_t nxt[D][NW];
#pragma unroll
    for (int i = 0; i < D; i++)
        load_row(rbase0 + T - 2 + i, nxt[(T - 2 + i) % D]);
    const u3_cc vt = (u3_cc)J.vfv;
    const uint32_t doff = (uint32_t)(2 * NO) * (uint32_t)g;

    auto emit = [&](int y, const uint32_t (&p0)[NO], const uint32_t (&p1)[NO]) {
        const int yc = min(max(y, 0), dstH - 1);
        const uint32_t c0 = vt[2 * yc], c1 = vt[2 * yc + 1];
        uint32_t o[NO / 2];
        if (NO == 6) {
            u3_v6(reinterpret_cast<uint32_t (&)[3]>(o), reinterpret_cast<const uint32_t (&)[6]>(p0), reinterpret_cast<const uint32_t (&)[6]>(p1), c0, c1,
                  vseed, vsh, maxpk, dmsb);
        } else {
#pragma unroll
            for (int q = 0; q < NO / 4; q++)
                u3_v4(o + 2 * q, p0 + 4 * q, p1 + 4 * q, c0, c1, vseed, vsh, maxpk, dmsb);
        }
        if (act && y >= a && y < b) {
            u3_gp d = (u3_gp)(dbase + (ptrdiff_t)y * dstride) + doff;
            if (NO == 6) {
                *(u3_g3)d = (u3_u3){ o[0], o[1], o[2] };
            } else {
                *(u3_g4)d = (u3_u4){ o[0], o[1], o[2], o[3 % (NO / 2)] };
            }
        }
    };
    auto step = [&](int rbase, auto uc, auto ec) {
        constexpr int u = decltype(uc)::value;
        constexpr bool EMIT = decltype(ec)::value;
        const int r = rbase + u;
        uint32_t cur[NW];
#pragma unroll
        for (int i = 0; i < NW; i++)
            cur[i] = nxt[u % D][i];
        load_row(r + D, nxt[u % D]);
        int h[NO];
        hpass(cur, h);
        /* P(r - 1) = (row r - 1, row r): v_cvt_pk_i16_i32 saturates = min(., 32767) (no sum of an admitted bank falls below -32768) */
#pragma unroll
        for (int c = 0; c < NO; c++) {
            ring[(u + 2) % 3][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[c], h[c]));
            hprev[c] = h[c];
        }
        if (EMIT) {
            /* the outputs that end on this row, in ascending order: j with (u - off(j) - 3) a multiple of PIN (rbase is one) */
#pragma unroll
            for (int j = 0; j < POUT; j++) {
                const int e = u - u3_off<PIN, POUT>(j) - 3; /* = PIN m - rbase */
                if (((e % PIN) + PIN) % PIN == 0)
                    emit(POUT * ((rbase + e) / PIN) + j, ring[u % 3], ring[(u + 2) % 3]);
            }
        }
    };
    step(rbase0, std::integral_constant<int, T - 2>(), std::false_type());
    step(rbase0, std::integral_constant<int, T - 1>(), std::false_type());
    for (int rbase = rbase0 + T; ; rbase += T) {
        step(rbase, std::integral_constant<int, 0>(), std::true_type());
        if (rbase + 1 > r_last) return;
        step(rbase, std::integral_constant<int, 1>(), std::true_type());
        if (rbase + 2 > r_last) return;
        step(rbase, std::integral_constant<int, 2>(), std::true_type());
        if (rbase + 3 > r_last) return;
        if constexpr (T == 6) {
            step(rbase, std::integral_constant<int, 3>(), std::true_type());
            if (rbase + 4 > r_last) return;
            step(rbase, std::integral_constant<int, 4>(), std::true_type());
            if (rbase + 5 > r_last) return;
            step(rbase, std::integral_constant<int, 5>(), std::true_type());
            if (rbase + 6 > r_last) return;
        }
    }
}

/* ================================================================================================== */
/*
 * The 8-bit twin (round 6): the same periods for planes of bytes and byte-interleaved U/V pairs (NV12 in and out) — hScale8To15_c
 * (libswscale/swscale.c:128-142), nv12ToUV_c (input.c:936), yuv2planeX_8_c / yuv2nv12cX_c (output.c:468-529) as in sws_up2.hip.  A lane owns
 * NO = 4 POUT output bytes — four periods of a plane, two periods x two channels of a pair — from the NWD dwords at the dword-aligned byte
 * 4 PIN g - 4 (the windows start two samples / columns before the lane's first one); every window is a FIXED byte position in them: the
 * int16 pairs come out of v_perm_b32 with static selectors.  720p -> 1080p and 1080p -> 1440p ran on the general 4-tap column walker at
 * 0.35 - 0.40 of HBM.
 */
template <int Q, int STEP, int N>
__device__ __forceinline__ uint32_t u3_bpair(const uint32_t (&w)[N])
{
    constexpr int i = Q >> 2, r = Q & 3;
    constexpr uint32_t sel = 0x0c000c00u | (uint32_t)(r + STEP) << 16 | (uint32_t)r;
    return __builtin_amdgcn_perm(w[i + 1 > N - 1 ? N - 1 : i + 1], w[i], sel);
}
/* four output bytes: t[i] = seed + pa[i] . f01 + pb[i] . f23, clip_u8(t[i] >> 19) packed in sample order */
__device__ __forceinline__ uint32_t u3_v4b(const uint32_t *pa, const uint32_t *pb, uint32_t f01, uint32_t f23, int seed)
{
    uint32_t w;
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %1, %5, %13, %15\n\t"
        "v_dot2_i32_i16 %2, %6, %13, %15\n\t"
        "v_dot2_i32_i16 %3, %7, %13, %15\n\t"
        "v_dot2_i32_i16 %4, %8, %13, %15\n\t"
        "v_dot2_i32_i16 %1, %9, %14, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %14, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %14, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %14, %4\n\t"
        "s_nop 1\n\t"
        "v_ashr_pk_u8_i32 %0, %1, %2, 19\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 19 op_sel:[0,0,0,1]"
        : "=&v"(w), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "s"(f01), "s"(f23), "v"(seed));
    return w;
}

template <int PIN, int POUT, int PAIR>
__device__ __forceinline__ void u32b_unit(const FFHipU32Job &J, int frame, int gbase, int strip, int lane)
{
    constexpr int NO = 4 * POUT, NWD = PIN + 2; /* 4 PIN source bytes + 8 around them */
    const int graw = gbase + lane;
    const bool act = graw < J.ngroups;
    const int g = min(graw, J.ngroups - 1);
    const bool lb = g == 0, rb = g == J.ngroups - 1;
    const bool border = gbase == 0 || gbase + 64 >= J.ngroups; /* wave-uniform */
    /* the first / last lane of a row loads its dwords one further inside and rebuilds the replicated ones */
    const uint32_t soff = (uint32_t)(lb ? 0 : 4 * PIN * g - 4 - (rb ? 4 : 0));
    uint32_t cf[NO][2];
    {
        /* plane: outputs NO g .. ; pair: columns (NO / 2) g .., both channels of a column share its coefficients */
        const uint32_t *p = J.hfv + (size_t)g * (PAIR ? NO : 2 * NO);
#pragma unroll
        for (int j = 0; j < NO; j++)
#pragma unroll
            for (int k = 0; k < 2; k++)
                cf[j][k] = p[2 * (PAIR ? j >> 1 : j) + k];
    }
    const int a = strip * J.strip_rows, b = min(a + J.strip_rows, J.dstH);
    const uint8_t *sbase = J.src + (size_t)frame * J.sfp;
    uint8_t *dbase = J.dst + (size_t)frame * J.dfp;
    const ptrdiff_t sstride = J.sstride, dstride = J.dstride;
    const int srcH = J.srcH;

    auto load_row = [&](int r, uint32_t (&w)[NWD]) {
        const uint8_t *p = sbase + (ptrdiff_t)min(max(r, 0), srcH - 1) * sstride;
        const u3_u4 v = *(u3_gc4)((u3_gcp)p + soff);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        if (NWD == 5)
            w[NWD - 1] = *(u3_gc1)((u3_gcp)p + soff + 16);
    };
    auto hpass = [&](const uint32_t (&raw)[NWD], int (&h)[NO]) {
        uint32_t w[NWD];
#pragma unroll
        for (int i = 0; i < NWD; i++)
            w[i] = raw[i];
        if (border) {
            const uint32_t f0 = PAIR ? __builtin_amdgcn_perm(raw[0], raw[0], 0x01000100u) : __builtin_amdgcn_perm(raw[0], raw[0], 0x00000000u);
            const uint32_t fl = PAIR ? __builtin_amdgcn_perm(raw[NWD - 1], raw[NWD - 1], 0x03020302u) : __builtin_amdgcn_perm(raw[NWD - 1], raw[NWD - 1], 0x03030303u);
#pragma unroll
            for (int i = 0; i < NWD; i++)
                w[i] = lb ? (i ? raw[i - 1] : f0) : rb ? (i + 1 < NWD ? raw[i + 1] : fl) : raw[i];
        }
        /* window start of output e, in bytes from the loaded base:
         *   plane: e = POUT k + j -> PIN k + off(j) + 4;   pair: e = 2 col + ch, col = POUT k + j -> 2 (PIN k + off(j) + 2) + ch */
#define U3B_S(e) (PAIR ? 2 * (PIN * (((e) >> 1) / POUT) + u3_off<PIN, POUT>(((e) >> 1) % POUT) + 2) + ((e) & 1) : PIN * ((e) / POUT) + u3_off<PIN, POUT>((e) % POUT) + 4)
#define U3B_Q(e0)                                                                                                                        \
        {                                                                                                                                \
            constexpr int ST = PAIR ? 2 : 1;                                                                                             \
            const uint32_t pa[4] = { u3_bpair<U3B_S(e0), ST>(w), u3_bpair<U3B_S(e0 + 1), ST>(w), u3_bpair<U3B_S(e0 + 2), ST>(w), u3_bpair<U3B_S(e0 + 3), ST>(w) }; \
            const uint32_t pb[4] = { u3_bpair<U3B_S(e0) + 2 * ST, ST>(w), u3_bpair<U3B_S(e0 + 1) + 2 * ST, ST>(w), u3_bpair<U3B_S(e0 + 2) + 2 * ST, ST>(w),    \
                                     u3_bpair<U3B_S(e0 + 3) + 2 * ST, ST>(w) };                                                          \
            u3_h4(h + (e0), pa, pb, cf + (e0), 7);                                                                                       \
        }
        U3B_Q(0) U3B_Q(4) U3B_Q(8)
        if constexpr (NO == 16)
            U3B_Q(12 % NO)
#undef U3B_Q
#undef U3B_S
    };

    constexpr int T = PIN % 3 ? 3 * PIN : PIN;
    const int rbase0 = PIN * (a / POUT) - T, r_last = PIN * (b / POUT - 1) + u3_off<PIN, POUT>(POUT - 1) + 3, dstH = J.dstH;
    uint32_t ring[3][NO];
    int hprev[NO];
#pragma unroll
    for (int c = 0; c < NO; c++)
        hprev[c] = 0;
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int c = 0; c < NO; c++)
            ring[s][c] = 0;
    constexpr int D = T % 2 ? 3 : 2; /* source rows in flight (two keep the 3:2 lanes at 128 registers: four waves per SIMD) */
    uint32_t nxt[D][NWD];
#pragma unroll
    for (int i = 0; i < D; i++)
        load_row(rbase0 + T - 2 + i, nxt[(T - 2 + i) % D]);
    const u3_cc vt = (u3_cc)J.vfv;
    const uint32_t doff = (uint32_t)NO * (uint32_t)g;

    auto emit = [&](int y, const uint32_t (&p0)[NO], const uint32_t (&p1)[NO]) {
        const int yc = min(max(y, 0), dstH - 1);
        const uint32_t c0 = vt[2 * yc], c1 = vt[2 * yc + 1];
        uint32_t o[NO / 4];
#pragma unroll
        for (int q = 0; q < NO / 4; q++)
            o[q] = u3_v4b(p0 + 4 * q, p1 + 4 * q, c0, c1, 64 << 12);
        if (act && y >= a && y < b) {
            u3_gp d = (u3_gp)(dbase + (ptrdiff_t)y * dstride) + doff;
            if (NO == 12) {
                *(u3_g3)d = (u3_u3){ o[0], o[1], o[2] };
            } else {
                *(u3_g4)d = (u3_u4){ o[0], o[1], o[2], o[3 % (NO / 4)] };
            }
        }
    };
    auto step = [&](int rbase, auto uc, auto ec) {
        constexpr int u = decltype(uc)::value;
        constexpr bool EMIT = decltype(ec)::value;
        const int r = rbase + u;
        uint32_t cur[NWD];
#pragma unroll
        for (int i = 0; i < NWD; i++)
            cur[i] = nxt[u % D][i];
        load_row(r + D, nxt[u % D]);
        int h[NO];
        hpass(cur, h);
#pragma unroll
        for (int c = 0; c < NO; c++) {
            ring[(u + 2) % 3][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[c], h[c]));
            hprev[c] = h[c];
        }
        if (EMIT) {
#pragma unroll
            for (int j = 0; j < POUT; j++) {
                const int e = u - u3_off<PIN, POUT>(j) - 3;
                if (((e % PIN) + PIN) % PIN == 0)
                    emit(POUT * ((rbase + e) / PIN) + j, ring[u % 3], ring[(u + 2) % 3]);
            }
        }
    };
    step(rbase0, std::integral_constant<int, T - 2>(), std::false_type());
    step(rbase0, std::integral_constant<int, T - 1>(), std::false_type());
    for (int rbase = rbase0 + T; ; rbase += T) {
        step(rbase, std::integral_constant<int, 0>(), std::true_type());
        if (rbase + 1 > r_last) return;
        step(rbase, std::integral_constant<int, 1>(), std::true_type());
        if (rbase + 2 > r_last) return;
        step(rbase, std::integral_constant<int, 2>(), std::true_type());
        if (rbase + 3 > r_last) return;
        if constexpr (T == 6) {
            step(rbase, std::integral_constant<int, 3>(), std::true_type());
            if (rbase + 4 > r_last) return;
            step(rbase, std::integral_constant<int, 4>(), std::true_type());
            if (rbase + 5 > r_last) return;
            step(rbase, std::integral_constant<int, 5>(), std::true_type());
            if (rbase + 6 > r_last) return;
        }
    }
}

template <int R43> /* a kernel per period: the (3 in, 4 out) lanes hold a third more registers */
__global__ __launch_bounds__(256) void k_sws_up32b(FFHipU32Args A)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave;
    if (gw >= (uint32_t)A.units_per_frame * (uint32_t)A.nframes)
        return;
    const int frame = (int)(gw / (uint32_t)A.units_per_frame);
    const int u = (int)(gw - (uint32_t)frame * (uint32_t)A.units_per_frame);
    int j = 0;
    if (A.njobs > 1 && u >= A.job[1].unit_begin) j = 1;
    if (A.njobs > 2 && u >= A.job[2].unit_begin) j = 2;
    const FFHipU32Job &J = A.job[j];
    const int local = u - J.unit_begin;
    const int strip = local / J.ncb, cb = local - strip * J.ncb;
    if (J.pair)
        u32b_unit<R43 ? 3 : 2, R43 ? 4 : 3, 1>(J, frame, cb * 64, strip, lane);
    else
        u32b_unit<R43 ? 3 : 2, R43 ? 4 : 3, 0>(J, frame, cb * 64, strip, lane);
}

__global__ __launch_bounds__(256) void k_sws_up32(FFHipU32Args A)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave;
    if (gw >= (uint32_t)A.units_per_frame * (uint32_t)A.nframes)
        return;
    const int frame = (int)(gw / (uint32_t)A.units_per_frame);
    const int u = (int)(gw - (uint32_t)frame * (uint32_t)A.units_per_frame);
    int j = 0;
    if (A.njobs > 1 && u >= A.job[1].unit_begin) j = 1;
    if (A.njobs > 2 && u >= A.job[2].unit_begin) j = 2;
    const FFHipU32Job &J = A.job[j];
    const int local = u - J.unit_begin;
    const int strip = local / J.ncb, cb = local - strip * J.ncb;
    if (A.ratio43) { /* 4:3 (1080p -> 1440p): period (3 in, 4 out) */
        if (J.pair)
            u32_unit<3, 4, 1>(A, J, frame, cb * 64, strip, lane);
        else
            u32_unit<3, 4, 0>(A, J, frame, cb * 64, strip, lane);
    } else if (J.pair) {
        u32_unit<2, 3, 1>(A, J, frame, cb * 64, strip, lane);
    } else {
        u32_unit<2, 3, 0>(A, J, frame, cb * 64, strip, lane);
    }
}

/* ================================================================================================== */
/* host side */

/*
 * Re-express a bank of an exact 3:2 up-scale (at most 4 taps) as coefficients on the REGULAR windows of the edge-replicated row:
 * output x reads samples clamp(2 (x / 3) - 2 + x % 3 + k), k = 0..3.  Every non-zero tap of the bank row must sit on one of those
 * samples; taps the reference folded onto the edge sample land on one of the replicas.  Output: n_dst x 2 dwords, (c0, c1) (c2, c3) as
 * int16 pairs.  Returns 0 when the bank is not of this shape.
 */
int ffhip_u32_virtual_bank(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, int pin, int pout, std::vector<uint32_t> *out)
{
    if ((long)pin * n_dst != (long)pout * n_src || fsize < 1 || fsize > 16 || (n_dst % pout) || !((pin == 2 && pout == 3) || (pin == 3 && pout == 4)))
        return 0;
    out->assign((size_t)n_dst * 2, 0);
    for (int x = 0; x < n_dst; x++) {
        const int s0 = pin * (x / pout) + (pin == 2 ? u3_off<2, 3>(x % pout) : u3_off<3, 4>(x % pout));
        int16_t v[4] = { 0 };
        bool used[4] = { false };
        for (int i = 0; i < fsize; i++) {
            const int16_t c = filter[(size_t)x * fsize + i];
            if (!c)
                continue;
            const int p = pos[x] + i;
            if (p < 0 || p >= n_src)
                return 0;
            int k = 0;
            for (; k < 4; k++) {
                int q = s0 + k;
                q = q < 0 ? 0 : q >= n_src ? n_src - 1 : q;
                if (q == p && !used[k])
                    break;
            }
            if (k == 4)
                return 0;
            used[k] = true;
            v[k] = c;
        }
        for (int k = 0; k < 2; k++)
            (*out)[(size_t)2 * x + k] = (uint16_t)v[2 * k] | ((uint32_t)(uint16_t)v[2 * k + 1] << 16);
    }
    return 1;
}

int ffhip_launch_up32(FFHipU32Args &A, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    /* strips of 72 output rows, shorter until the launch has the waves the chip keeps resident (a strip re-filters three source rows) */
    for (int want = 72; ; want >>= 1) {
        long long u = 0;
        for (int i = 0; i < A.njobs; i++) {
            FFHipU32Job &j = A.job[i];
            const int pout = A.ratio43 ? 4 : 3;
            if (j.ngroups < 3 || j.dstH <= 0 || (j.dstH % pout)) {
                ffhip_set_error("ffhip_sws: the exact-3:2 up-scaler takes rows of three groups or more and a multiple of three output rows");
                return FFHIP_EINVAL;
            }
            const int n = cdiv(j.dstH, want);
            j.strip_rows = cdiv(cdiv(j.dstH, n), 3 * pout) * 3 * pout;
            j.nstrips = cdiv(j.dstH, j.strip_rows);
            j.ncb = cdiv(j.ngroups, 64);
            u += (long long)j.ncb * j.nstrips;
        }
        if (u * A.nframes >= 8192 || want <= 12)
            break;
    }
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
    if (A.bytes && A.ratio43)
        hipLaunchKernelGGL(k_sws_up32b<1>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, A);
    else if (A.bytes)
        hipLaunchKernelGGL(k_sws_up32b<0>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, A);
    else
        hipLaunchKernelGGL(k_sws_up32, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, A);
    LAUNCH_CHECK();
    return 0;
}
