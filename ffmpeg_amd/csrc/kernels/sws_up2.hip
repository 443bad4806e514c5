/*
 * sws_up2.hip — the fused H+V scaler for EXACT 2x up-scaling with 4-tap banks in both directions (bicubic / bilinear /
 * point 1080p -> 4K: BASELINE config "swscale bicubic 1080p->4K nv12, 256-frame batch"), planes and byte-interleaved
 * U/V pairs (NV12 / NV21) in and out.
 *
 * Arithmetic: hScale8To15_c (libswscale/swscale.c:128-142), nv12ToUV_c (input.c:936), yuv2planeX_8_c / yuv2nv12cX_c
 * (output.c:468-529) — int32 sums, >> 7 and min(., 32767) for the horizontal pass, the 64 << 12 seed, >> 19 and the clip
 * to 8 bits for the vertical one.  Same results as sws_colwalk.hip / sws_scale.hip, bit for bit (tests/test_gpu_sws_fast.py).
 *
 * What exact 2x buys over the general column walker (sws_colwalk.hip, 7.0 VALU instructions per output sample, VALU-bound):
 *
 *  - Window positions are REGULAR: output x = 2j + ph reads source samples j - 2 + ph .. j + 1 + ph.  initFilter()
 *    (libswscale/utils.c:519-561) folds the taps that fall outside the row onto the first / last sample, i.e. the bank
 *    is the regular one over an edge-REPLICATED row — with its own, renormalised coefficients in the three columns next
 *    to either edge.  The host re-expresses every bank row as four coefficients on the regular window of the replicated
 *    row (ffhip_up2_virtual_bank, verified tap by tap, else this kernel is not used), so a lane needs no position
 *    table and no byte selectors of its own: 8 adjacent output columns read 8 adjacent source bytes, and the SEVEN
 *    distinct (s[k], s[k+1]) int16 pairs of those bytes serve all sixteen v_dot2_i32_i16 of the row (12 v_perm_b32 per 8
 *    samples in the general kernel).  Replication is a 2-instruction fix-up that only the waves at a row's ends execute.
 *  - The vertical schedule is STATIC: after source row r both output rows 2r-3 and 2r-2 are due, and both read rows
 *    r-3 .. r (clamped = replicated).  No `need` bookkeeping, no v_readlane: the rows' coefficient pairs arrive through
 *    scalar loads (one s_load_dwordx8 per two steps) and sit in SGPRs, the one scalar operand a VOP3P instruction takes.
 *  - The two halves of an output dword are written by v_ashr_pk_u8_i32 and v_ashr_pk_u8_i32 op_sel:[0,0,0,1] (the
 *    high-half form keeps the low half): no merging v_perm.  Dots, shifts and packs of a row are one hand-scheduled asm
 *    block in which every DOT result is consumed >= 3 instructions after it was written (the gfx950 DOT hazard).
 *  - Source rows are consumed in place (the prefetch of a buffer slot is issued after its row was filtered): no copies.
 *
 * Per 16 output samples: 39 (horizontal, once per source row) + 2 x 20 (vertical) VALU instructions = 4.9 per sample.
 *
 * Geometry: a wave owns 64 lanes x 8 output bytes of a strip of rows and walks down the source rows.  When a row does
 * not fill a whole number of waves (3840 columns = 7.5 x 512) the column blocks at the ragged right end are shared by 2 or
 * 4 FRAMES of the batch (the same strip and columns of frames f, f+1: identical schedule, the frame pitch is part of the
 * lane offset), so no lane idles there; every other wave stays inside one frame — 512 contiguous bytes per store
 * instruction (measured with the arithmetic taken out: 256-byte pieces cost 7 % of the bandwidth).
 */
#include <stdlib.h>
#include <type_traits>
#include <vector>

#include "common.h"
#include "sws_kernels.h"

typedef uint32_t up_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t up_u3 __attribute__((ext_vector_type(3)));
typedef uint32_t up_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t up_u8 __attribute__((ext_vector_type(8)));
typedef up_u3 __attribute__((aligned(4))) up_u3a;
typedef const uint8_t __attribute__((address_space(1))) *up_gcp;
typedef uint8_t __attribute__((address_space(1))) *up_gp;
typedef const up_u3a __attribute__((address_space(1))) *up_gc3;
typedef up_u2 __attribute__((address_space(1))) *up_g2;
typedef up_u4 __attribute__((aligned(4))) up_u4a;
typedef up_u2 __attribute__((aligned(4))) up_u2a;
typedef const up_u4a __attribute__((address_space(1))) *up_gc4;
typedef const up_u2a __attribute__((address_space(1))) *up_gc2;
typedef up_u4a __attribute__((address_space(1))) *up_g4;
typedef const up_u8 __attribute__((address_space(4))) *up_cc8; /* constant address space: scalar loads */

struct UpRaw { uint32_t q[3]; };
struct UpRawH { uint32_t q[6]; }; /* 16-bit samples: 8 of a plane (4 dwords) or 6 (u, v) columns of an interleaved pair (6 dwords) */

/*
 * Four horizontal samples: d[i] = (pa[i] . ca[i] + pb[i] . cb[i]) >> 7.  The two DOT chains per sample are interleaved
 * four wide; each result is shifted three instructions after its last DOT wrote it, and leaves the block as the
 * result of a plain VALU instruction.
 */
/* the same for samples above 8 bits: >> (depth - 1) (hScale16To15_c, libswscale/swscale.c:99-126), the shift in an SGPR */
__device__ __forceinline__ void up_h4s(int (&d)[4], uint32_t pa0, uint32_t pa1, uint32_t pa2, uint32_t pa3, uint32_t pb0,
                                       uint32_t pb1, uint32_t pb2, uint32_t pb3, const uint32_t *cf, int sh)
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
        : "v"(pa0), "v"(pa1), "v"(pa2), "v"(pa3), "v"(pb0), "v"(pb1), "v"(pb2), "v"(pb3),
          "v"(cf[0]), "v"(cf[2]), "v"(cf[4]), "v"(cf[6]), "v"(cf[1]), "v"(cf[3]), "v"(cf[5]), "v"(cf[7]), "s"(sh));
}

__device__ __forceinline__ void up_h4(int (&d)[4], uint32_t pa0, uint32_t pa1, uint32_t pa2, uint32_t pa3, uint32_t pb0,
                                      uint32_t pb1, uint32_t pb2, uint32_t pb3, const uint32_t *cf)
{
    asm("v_dot2_i32_i16 %0, %4, %12, 0\n\t"
        "v_dot2_i32_i16 %1, %5, %13, 0\n\t"
        "v_dot2_i32_i16 %2, %6, %14, 0\n\t"
        "v_dot2_i32_i16 %3, %7, %15, 0\n\t"
        "v_dot2_i32_i16 %0, %8, %16, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %17, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %18, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %19, %3\n\t"
        "v_ashrrev_i32 %0, 7, %0\n\t"
        "v_ashrrev_i32 %1, 7, %1\n\t"
        "v_ashrrev_i32 %2, 7, %2\n\t"
        "v_ashrrev_i32 %3, 7, %3"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(pa0), "v"(pa1), "v"(pa2), "v"(pa3), "v"(pb0), "v"(pb1), "v"(pb2), "v"(pb3),
          "v"(cf[0]), "v"(cf[2]), "v"(cf[4]), "v"(cf[6]), "v"(cf[1]), "v"(cf[3]), "v"(cf[5]), "v"(cf[7]));
}

/* the same four samples with the bank in SGPRs (round 6, SC): even columns (e0, e1), odd columns (o0, o1) — one scalar operand per DOT */
__device__ __forceinline__ void up_h4c(int (&d)[4], uint32_t pa0, uint32_t pa1, uint32_t pa2, uint32_t pa3, uint32_t pb0,
                                       uint32_t pb1, uint32_t pb2, uint32_t pb3, uint32_t e0, uint32_t e1, uint32_t o0, uint32_t o1)
{
    asm("v_dot2_i32_i16 %0, %4, %12, 0\n\t"
        "v_dot2_i32_i16 %1, %5, %14, 0\n\t"
        "v_dot2_i32_i16 %2, %6, %12, 0\n\t"
        "v_dot2_i32_i16 %3, %7, %14, 0\n\t"
        "v_dot2_i32_i16 %0, %8, %13, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %15, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %13, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %15, %3\n\t"
        "v_ashrrev_i32 %0, 7, %0\n\t"
        "v_ashrrev_i32 %1, 7, %1\n\t"
        "v_ashrrev_i32 %2, 7, %2\n\t"
        "v_ashrrev_i32 %3, 7, %3"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(pa0), "v"(pa1), "v"(pa2), "v"(pa3), "v"(pb0), "v"(pb1), "v"(pb2), "v"(pb3), "s"(e0), "s"(e1), "s"(o0), "s"(o1));
}
/* one sample of a row's end columns (their own coefficients): the compiler's DOT, hazards and all */
__device__ __forceinline__ int up_h1(uint32_t pa, uint32_t pb, uint32_t c0, uint32_t c1)
{
    typedef short up_s2 __attribute__((ext_vector_type(2)));
    int r = __builtin_amdgcn_sdot2(__builtin_bit_cast(up_s2, pa), __builtin_bit_cast(up_s2, c0), 0, false);
    r = __builtin_amdgcn_sdot2(__builtin_bit_cast(up_s2, pb), __builtin_bit_cast(up_s2, c1), r, false);
    return r >> 7;
}

/*
 * One output row of 8 samples: t[i] = kround + pa[i] . f01 + pb[i] . f23, bytes clip_u8(t[i] >> 19) packed in sample order;
 * the upper halves of the two dwords are written by the op_sel form of v_ashr_pk_u8_i32 (it keeps the lower half: verified on
 * the hardware by the parity tests, a v_perm_b32 merge gives the same bytes).
 */
__device__ __forceinline__ void up_v8(uint32_t &w0, uint32_t &w1, const uint32_t (&pa)[8], const uint32_t (&pb)[8], uint32_t f01,
                                      uint32_t f23, int kround)
{
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_dot2_i32_i16 %2, %10, %26, %28\n\t"
        "v_dot2_i32_i16 %3, %11, %26, %28\n\t"
        "v_dot2_i32_i16 %4, %12, %26, %28\n\t"
        "v_dot2_i32_i16 %5, %13, %26, %28\n\t"
        "v_dot2_i32_i16 %6, %14, %26, %28\n\t"
        "v_dot2_i32_i16 %7, %15, %26, %28\n\t"
        "v_dot2_i32_i16 %8, %16, %26, %28\n\t"
        "v_dot2_i32_i16 %9, %17, %26, %28\n\t"
        "v_dot2_i32_i16 %2, %18, %27, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %27, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %27, %4\n\t"
        "v_dot2_i32_i16 %5, %21, %27, %5\n\t"
        "v_dot2_i32_i16 %6, %22, %27, %6\n\t"
        "v_dot2_i32_i16 %7, %23, %27, %7\n\t"
        "v_dot2_i32_i16 %8, %24, %27, %8\n\t"
        "v_dot2_i32_i16 %9, %25, %27, %9\n\t"
        "v_ashr_pk_u8_i32 %0, %2, %3, 19\n\t"
        "v_ashr_pk_u8_i32 %1, %6, %7, 19\n\t"
        "v_ashr_pk_u8_i32 %0, %4, %5, 19 op_sel:[0,0,0,1]\n\t"
        "v_ashr_pk_u8_i32 %1, %8, %9, 19 op_sel:[0,0,0,1]"
        : "=&v"(w0), "=&v"(w1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pa[6]), "v"(pa[7]),
          "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pb[4]), "v"(pb[5]), "v"(pb[6]), "v"(pb[7]),
          "s"(f01), "s"(f23), "v"(kround));
}

/*
 * The same row for targets above 8 bits (yuv2planeX_10_c_template, libswscale/output.c:341-360): t[i] >> (27 - depth), clipped to
 * depth bits, two samples per dword: v_cvt_pk_i16_i32 saturates to int16 (depth <= 14: the clip range lies inside), v_pk_max_i16 / v_pk_min_i16
 * clip the pair to 0 .. 2^depth - 1.
 */
__device__ __forceinline__ void up_v8h(uint32_t (&w)[4], const uint32_t (&pa)[8], const uint32_t (&pb)[8], uint32_t f01, uint32_t f23,
                                       int kround, int sh, uint32_t maxpk)
{
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_dot2_i32_i16 %4, %12, %28, %30\n\t"
        "v_dot2_i32_i16 %5, %13, %28, %30\n\t"
        "v_dot2_i32_i16 %6, %14, %28, %30\n\t"
        "v_dot2_i32_i16 %7, %15, %28, %30\n\t"
        "v_dot2_i32_i16 %8, %16, %28, %30\n\t"
        "v_dot2_i32_i16 %9, %17, %28, %30\n\t"
        "v_dot2_i32_i16 %10, %18, %28, %30\n\t"
        "v_dot2_i32_i16 %11, %19, %28, %30\n\t"
        "v_dot2_i32_i16 %4, %20, %29, %4\n\t"
        "v_dot2_i32_i16 %5, %21, %29, %5\n\t"
        "v_dot2_i32_i16 %6, %22, %29, %6\n\t"
        "v_dot2_i32_i16 %7, %23, %29, %7\n\t"
        "v_dot2_i32_i16 %8, %24, %29, %8\n\t"
        "v_dot2_i32_i16 %9, %25, %29, %9\n\t"
        "v_dot2_i32_i16 %10, %26, %29, %10\n\t"
        "v_dot2_i32_i16 %11, %27, %29, %11\n\t"
        "v_ashrrev_i32 %4, %31, %4\n\t"
        "v_ashrrev_i32 %5, %31, %5\n\t"
        "v_ashrrev_i32 %6, %31, %6\n\t"
        "v_ashrrev_i32 %7, %31, %7\n\t"
        "v_ashrrev_i32 %8, %31, %8\n\t"
        "v_ashrrev_i32 %9, %31, %9\n\t"
        "v_ashrrev_i32 %10, %31, %10\n\t"
        "v_ashrrev_i32 %11, %31, %11\n\t"
        "v_cvt_pk_i16_i32 %0, %4, %5\n\t"
        "v_cvt_pk_i16_i32 %1, %6, %7\n\t"
        "v_cvt_pk_i16_i32 %2, %8, %9\n\t"
        "v_cvt_pk_i16_i32 %3, %10, %11\n\t"
        "v_pk_max_i16 %0, %0, 0\n\t"
        "v_pk_max_i16 %1, %1, 0\n\t"
        "v_pk_max_i16 %2, %2, 0\n\t"
        "v_pk_max_i16 %3, %3, 0\n\t"
        "v_pk_min_i16 %0, %0, %32\n\t"
        "v_pk_min_i16 %1, %1, %32\n\t"
        "v_pk_min_i16 %2, %2, %32\n\t"
        "v_pk_min_i16 %3, %3, %32"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6),
          "=&v"(t7)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pa[6]), "v"(pa[7]),
          "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pb[4]), "v"(pb[5]), "v"(pb[6]), "v"(pb[7]),
          "s"(f01), "s"(f23), "v"(kround), "v"(sh), "v"(maxpk));
}

/*
 * One unit = one wave.  PAIR 0: a plane, 8 output columns per lane.  PAIR 1: a byte-interleaved U/V pair, 4 + 4 output
 * samples per lane.  Either way a lane reads the 12 source bytes at 4g - 4 of every source row and writes the 8
 * destination bytes at 8g of two destination rows per source row (g = the lane's group in the row).
 * D = source rows in flight (3 or 6); the ring of vertical pairs has 3 slots, so D | 6 keeps every index static.
 * HB: samples above 8 bits (little-endian uint16, 9..14 bits, P01x's in the high bits): the same walk with 16 source bytes (plane:
 * 8 samples from 4g - 2) or 24 (pair: 6 (u, v) columns from 2g - 2) per lane and row, 16 destination bytes per lane and row; the
 * seven (s[k], s[k+1]) pairs of a plane are its dwords and three v_alignbyte, a pair's come from v_perm as at 8 bits.
 */
template <int PAIR, int D, int VAR, int HB = 0, int RC = 0, int SC = 0>
__device__ __forceinline__ void up2_unit(const FFHipUp2Job &J, int frame0, int fshift, int gbase, int strip, int lane, int nframes)
{
    /* measurement-only variants (wrong output; tools/sweep_sws.py): 16 never stores, 32 re-reads one source row (48 = both: the
     * arithmetic alone), 64 moves the bytes without arithmetic (the memory pattern alone) */
    constexpr bool DBG_NOST = VAR & 16, DBG_ROW0 = VAR & 32, DBG_COPY = VAR & 64;
    constexpr bool NTS = VAR & 1; /* measured variant: non-temporal stores */
    const int lpf = 64 >> fshift;                 /* lanes per frame */
    const int fsub = lane >> (6 - fshift);
    const int gl = lane & (lpf - 1);
    const int graw = gbase + gl, fraw = frame0 + fsub;
    const bool act = graw < J.ngroups && fraw < nframes;
    const int g = min(graw, J.ngroups - 1);
    const int fs = fraw < nframes ? fsub : 0;     /* idle lanes shadow valid ones */
    const bool lb = g == 0, rb = g == J.ngroups - 1;
    const bool border = gbase == 0 || gbase + lpf >= J.ngroups; /* wave-uniform */

    const uint32_t soff = (uint32_t)fs * (uint32_t)J.sfp +
                          (uint32_t)(!HB ? (lb ? 0 : rb ? 4 * g - 8 : 4 * g - 4)
                                     : PAIR ? (lb ? 0 : rb ? 8 * g - 16 : 8 * g - 8) : (lb ? 0 : rb ? 8 * g - 8 : 8 * g - 4));
    const uint32_t doff = (uint32_t)fs * (uint32_t)J.dfp + (HB ? 16u : 8u) * (uint32_t)g;
    /* above 8 bits: horizontal shift depth - 1, vertical shift 27 - depth with its rounding seed, the clip, P01x's alignment */
    const int hsh = HB ? J.hb_sdepth - 1 : 7, vsh = HB ? 27 - J.hb_ddepth : 19;
    const uint32_t maxpk = HB ? ((1u << J.hb_ddepth) - 1) * 0x00010001u : 0;
    const int smsb = HB ? (J.hb_smsb ? 16 - J.hb_sdepth : 0) : 0, dmsb = HB ? (J.hb_dmsb ? 16 - J.hb_ddepth : 0) : 0;

    /* ---- horizontal coefficients of this lane's columns (virtual bank: regular windows of the replicated row) ---- */
    /* SC (round 6): the bank in SGPRs — an even and an odd column's dwords, and the three columns at either end of a row, which only the
     * first / last lane of the row's first / last wave recompute: 16 VGPRs fewer (68 -> 52: eight waves per SIMD) */
    constexpr int NCF = SC ? 4 : PAIR ? 8 : 16;
    uint32_t cf[NCF];
    if (!SC) {
        const up_u4 *p = reinterpret_cast<const up_u4 *>(J.hfv) + (size_t)g * (NCF / 4);
#pragma unroll
        for (int i = 0; i < NCF / 4; i++) {
            const up_u4 v = p[i];
            cf[4 * i] = v.x; cf[4 * i + 1] = v.y; cf[4 * i + 2] = v.z; cf[4 * i + 3] = v.w;
        }
    }
    const uint32_t cE0 = J.hco[0], cE1 = J.hco[1], cO0 = J.hco[2], cO1 = J.hco[3];
    /* byte selectors: (s[k], s[k+1]) as an int16 pair.  Plane: adjacent bytes, sample b_k = byte k + 2 of the span.
     * Pair: bytes 2k (+1 for the channel that sits at the odd bytes) and 2k + 2. */
    const uint32_t par = PAIR ? (J.swap ? 0x00010001u : 0u) : 0u;
    const uint32_t sA0 = 0x0c020c00u + par, sA1 = 0x0c040c02u + par, sA2 = 0x0c060c04u + par; /* first output channel  */
    const uint32_t sB0 = 0x0c030c01u - par, sB1 = 0x0c050c03u - par, sB2 = 0x0c070c05u - par; /* second output channel */
    const uint32_t fixl = PAIR ? 0x01000100u : 0x00000000u, fixr = PAIR ? 0x03020302u : 0x03030303u;

    /* ---- row addressing: scalar running pointers + 32-bit lane offsets ---- */
    const int S = J.steps_per_strip;
    const int a = 1 + strip * S, b = min(a + S, J.srcH + 2); /* this strip's steps: step r emits rows 2r-3 and 2r-2 */
    const uint8_t *sbase = J.src + (size_t)frame0 * J.sfp;
    uint8_t *dbase = J.dst + (size_t)frame0 * J.dfp;
    const ptrdiff_t sstride = J.sstride, dstride = J.dstride;
    const int srcH = J.srcH, dstH = 2 * J.srcH;
    int pr = a - 3;                                               /* next source row to fetch (unclamped) */
    const uint8_t *pf = sbase + (ptrdiff_t)min(max(pr, 0), srcH - 1) * sstride;
    uint8_t *dr = dbase + (ptrdiff_t)(2 * a - 3) * dstride;      /* row 2a-3 (row -1 of the first strip is never stored) */
    asm("" : "+s"(pf), "+s"(dr));

    typedef typename std::conditional<HB != 0, UpRawH, UpRaw>::type Raw;
    auto load_next = [&](Raw &o) {
        uint32_t off = soff;
        asm volatile("" : "+v"(off)); /* keeps `uniform base + zext(lane offset)` next to the access: saddr addressing */
        if (HB) {
            const up_u4 w = *(up_gc4)((up_gcp)pf + off);
            o.q[0] = w.x; o.q[1] = w.y; o.q[2] = w.z; o.q[3] = w.w;
            if (PAIR) {
                const up_u2 x = *(up_gc2)((up_gcp)pf + off + 16);
                o.q[4] = x.x; o.q[5] = x.y;
            }
        } else {
            const up_u3 w = *(up_gc3)((up_gcp)pf + off);
            o.q[0] = w.x; o.q[1] = w.y; o.q[2] = w.z;
        }
        pr++;
        if (!DBG_ROW0)
            pf += (pr >= 1 && pr <= srcH - 1) ? sstride : 0; /* rows above / below the plane replicate the edge row */
        asm("" : "+s"(pf));
    };

    uint32_t ring[3][8];
    int hprev[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        hprev[i] = 0;
    int kround = HB ? 1 << (vsh - 1) : 64 << 12;
    asm volatile("" : "+v"(kround));

    /* horizontal pass of one source row: 8 samples, appended to the ring as (h[r-1], h[r]) pairs (int16-saturated:
     * equals min(., 32767) + truncation because no sum of the bank can fall below -32768, host-checked) */
    auto hpass = [&](const Raw &w, uint32_t (&Pnew)[8]) {
        int h[8];
        if (HB) {
            uint32_t v[6];
            constexpr int NQ = PAIR ? 6 : 4;
#pragma unroll
            for (int i = 0; i < NQ; i++)
                v[i] = w.q[i];
            if (border) {
                /* edge replication: the first / last lane of a row loaded its span one (plane) or two (pair) dwords further inside */
                if (PAIR) {
                    const uint32_t q0 = w.q[0], q1 = w.q[1], q2 = w.q[2], q3 = w.q[3], q4 = w.q[4], q5 = w.q[5];
                    v[0] = lb ? q0 : rb ? q2 : q0; v[1] = lb ? q0 : rb ? q3 : q1; v[2] = lb ? q0 : rb ? q4 : q2;
                    v[3] = lb ? q1 : rb ? q5 : q3; v[4] = lb ? q2 : rb ? q5 : q4; v[5] = lb ? q3 : rb ? q5 : q5;
                } else {
                    const uint32_t q0 = w.q[0], q1 = w.q[1], q2 = w.q[2], q3 = w.q[3];
                    const uint32_t f0 = __builtin_amdgcn_perm(q0, q0, 0x01000100u), f3 = __builtin_amdgcn_perm(q3, q3, 0x03020302u);
                    v[0] = lb ? f0 : rb ? q1 : q0; v[1] = lb ? q0 : rb ? q2 : q1;
                    v[2] = lb ? q1 : rb ? q3 : q2; v[3] = lb ? q2 : rb ? f3 : q3;
                }
            }
            if (smsb) { /* P01x: the samples sit in the high bits */
                typedef unsigned short up_h2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int i = 0; i < NQ; i++)
                    v[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(up_h2, v[i]) >> (unsigned short)smsb);
            }
            if (PAIR) {
                /* columns c0..c5 = (u, v) dwords; pair k of a channel = (c_k, c_k+1) halves */
                uint32_t a[5], b[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    a[k] = __builtin_amdgcn_perm(v[k + 1], v[k], 0x05040100u);
                    b[k] = __builtin_amdgcn_perm(v[k + 1], v[k], 0x07060302u);
                }
                int ha[4], hb[4];
                up_h4s(ha, a[0], a[1], a[1], a[2], a[2], a[3], a[3], a[4], cf, hsh);
                up_h4s(hb, b[0], b[1], b[1], b[2], b[2], b[3], b[3], b[4], cf, hsh);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    h[2 * i] = ha[i];
                    h[2 * i + 1] = hb[i];
                }
            } else {
                /* samples s0..s7 = the four dwords; the odd pairs are one funnel shift each */
                const uint32_t p0 = v[0], p2 = v[1], p4 = v[2], p6 = v[3];
                const uint32_t p1 = __builtin_amdgcn_alignbyte(v[1], v[0], 2), p3 = __builtin_amdgcn_alignbyte(v[2], v[1], 2);
                const uint32_t p5 = __builtin_amdgcn_alignbyte(v[3], v[2], 2);
                int hl[4], hh[4];
                up_h4s(hl, p0, p1, p1, p2, p2, p3, p3, p4, cf, hsh);
                up_h4s(hh, p2, p3, p3, p4, p4, p5, p5, p6, cf + 8, hsh);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    h[i] = hl[i];
                    h[4 + i] = hh[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                Pnew[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[i], h[i]));
                hprev[i] = h[i];
            }
            return;
        }
        uint32_t v0 = w.q[0], v1 = w.q[1], v2 = w.q[2];
        if (border) {
            /* edge replication: the first / last lane of a row loaded its span one dword further inside */
            const uint32_t f0 = __builtin_amdgcn_perm(w.q[0], w.q[0], fixl), f2 = __builtin_amdgcn_perm(w.q[2], w.q[2], fixr);
            v0 = lb ? f0 : rb ? w.q[1] : w.q[0];
            v1 = lb ? w.q[0] : rb ? w.q[2] : w.q[1];
            v2 = lb ? w.q[1] : rb ? f2 : w.q[2];
        }
        if (PAIR) {
            /* channel samples u0..u5 at bytes 0,2,..,10 (or 1,3,..,11); pair k = (u_k, u_k+1) */
            const uint32_t a0 = __builtin_amdgcn_perm(v1, v0, sA0), a1 = __builtin_amdgcn_perm(v1, v0, sA1);
            const uint32_t a2 = __builtin_amdgcn_perm(v1, v0, sA2), a3 = __builtin_amdgcn_perm(v2, v1, sA1);
            const uint32_t a4 = __builtin_amdgcn_perm(v2, v1, sA2);
            const uint32_t b0 = __builtin_amdgcn_perm(v1, v0, sB0), b1 = __builtin_amdgcn_perm(v1, v0, sB1);
            const uint32_t b2 = __builtin_amdgcn_perm(v1, v0, sB2), b3 = __builtin_amdgcn_perm(v2, v1, sB1);
            const uint32_t b4 = __builtin_amdgcn_perm(v2, v1, sB2);
            int ha[4], hb[4];
            /* output column c of a channel: window offset o = (c >> 1) + (c & 1) -> pairs o and o + 2 */
            if (SC) {
                up_h4c(ha, a0, a1, a1, a2, a2, a3, a3, a4, cE0, cE1, cO0, cO1);
                up_h4c(hb, b0, b1, b1, b2, b2, b3, b3, b4, cE0, cE1, cO0, cO1);
                if (border) { /* wave-uniform: columns 0..2 of the row's first lane, the last three of its last lane */
                    const uint32_t *L = J.hco + 4, *R = J.hco + 10;
                    const int l0 = up_h1(a0, a2, L[0], L[1]), l1 = up_h1(a1, a3, L[2], L[3]), l2 = up_h1(a1, a3, L[4], L[5]);
                    const int m0 = up_h1(b0, b2, L[0], L[1]), m1 = up_h1(b1, b3, L[2], L[3]), m2 = up_h1(b1, b3, L[4], L[5]);
                    const int r1 = up_h1(a1, a3, R[0], R[1]), r2 = up_h1(a1, a3, R[2], R[3]), r3 = up_h1(a2, a4, R[4], R[5]);
                    const int q1 = up_h1(b1, b3, R[0], R[1]), q2 = up_h1(b1, b3, R[2], R[3]), q3 = up_h1(b2, b4, R[4], R[5]);
                    ha[0] = lb ? l0 : ha[0]; ha[1] = lb ? l1 : rb ? r1 : ha[1]; ha[2] = lb ? l2 : rb ? r2 : ha[2]; ha[3] = rb ? r3 : ha[3];
                    hb[0] = lb ? m0 : hb[0]; hb[1] = lb ? m1 : rb ? q1 : hb[1]; hb[2] = lb ? m2 : rb ? q2 : hb[2]; hb[3] = rb ? q3 : hb[3];
                }
            } else {
                up_h4(ha, a0, a1, a1, a2, a2, a3, a3, a4, cf);
                up_h4(hb, b0, b1, b1, b2, b2, b3, b3, b4, cf);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                h[2 * i] = ha[i];
                h[2 * i + 1] = hb[i];
            }
        } else {
            /* samples b0..b7 = bytes 2..9 of the span */
            const uint32_t p0 = __builtin_amdgcn_perm(v1, v0, 0x0c030c02u), p1 = __builtin_amdgcn_perm(v1, v0, 0x0c040c03u);
            const uint32_t p2 = __builtin_amdgcn_perm(v1, v0, 0x0c050c04u), p3 = __builtin_amdgcn_perm(v1, v0, 0x0c060c05u);
            const uint32_t p4 = __builtin_amdgcn_perm(v1, v0, 0x0c070c06u), p5 = __builtin_amdgcn_perm(v2, v1, 0x0c040c03u);
            const uint32_t p6 = __builtin_amdgcn_perm(v2, v1, 0x0c050c04u);
            int hl[4], hh[4];
            if (SC) {
                up_h4c(hl, p0, p1, p1, p2, p2, p3, p3, p4, cE0, cE1, cO0, cO1);
                up_h4c(hh, p2, p3, p3, p4, p4, p5, p5, p6, cE0, cE1, cO0, cO1);
                if (border) { /* wave-uniform: columns 0..2 of the row's first lane, 5..7 of its last lane */
                    const uint32_t *L = J.hco + 4, *R = J.hco + 10;
                    const int l0 = up_h1(p0, p2, L[0], L[1]), l1 = up_h1(p1, p3, L[2], L[3]), l2 = up_h1(p1, p3, L[4], L[5]);
                    const int r5 = up_h1(p3, p5, R[0], R[1]), r6 = up_h1(p3, p5, R[2], R[3]), r7 = up_h1(p4, p6, R[4], R[5]);
                    hl[0] = lb ? l0 : hl[0]; hl[1] = lb ? l1 : hl[1]; hl[2] = lb ? l2 : hl[2];
                    hh[1] = rb ? r5 : hh[1]; hh[2] = rb ? r6 : hh[2]; hh[3] = rb ? r7 : hh[3];
                }
            } else {
                up_h4(hl, p0, p1, p1, p2, p2, p3, p3, p4, cf);
                up_h4(hh, p2, p3, p3, p4, p4, p5, p5, p6, cf + 8);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                h[i] = hl[i];
                h[4 + i] = hh[i];
            }
        }
        if (RC) {
            /* the reference converts the range of its int16 line buffer: min(h, 32767) is what that buffer holds, the product fits
             * 32 bits, and the pack below saturates at 32767 as ...ToJpeg's FFMIN does (the other direction never gets there; no
             * result falls below -32768: host-checked) */
            const int rcc = J.rc_coeff, rco = J.rc_offset;
#pragma unroll
            for (int i = 0; i < 8; i++)
                h[i] = (__mul24(min(h[i], 32767), rcc) + rco) >> 14; /* 16 x 17 bits: v_mad_i32_i24, full rate */
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            Pnew[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[i], h[i]));
            hprev[i] = h[i];
        }
    };

    Raw buf[D];
#pragma unroll
    for (int k = 0; k < D; k++)
        load_next(buf[k]);
    /* rows a-3, a-2, a-1: fill the ring */
#pragma unroll
    for (int k = 0; k < 3; k++) {
        hpass(buf[k % D], ring[k]);
        load_next(buf[k % D]);
    }

    /* vertical coefficient pairs: row y of the virtual bank at dwords 2 (y + 1), 2 (y + 1) + 1; a step reads rows
     * 2r-3, 2r-2 = 4 consecutive dwords from 4r - 4; two steps per scalar load */
    const uint32_t *vt = J.vfv;
    for (int r = a; r < b; r += 6) {
        up_u8 c8[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
            c8[q] = *(up_cc8)(vt + 4 * (r + 2 * q) - 4);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            Raw &w = buf[(k + 3) % D];
            if constexpr (HB != 0) {
                /* straight-line (round 6): every row of the trip is computed, the stores alone look at the strip's end — a uniform branch
                 * around the row made the compiler copy the rows in flight at its join and wait for them (docs/KERNELS.md R6.8) */
                const bool live = act && r + k < b;
                hpass(w, ring[k % 3]);
                const up_u8 cc = c8[k >> 1];
                const uint32_t fb01 = k & 1 ? cc.s4 : cc.s0, fb23 = k & 1 ? cc.s5 : cc.s1;
                const uint32_t fa01 = k & 1 ? cc.s6 : cc.s2, fa23 = k & 1 ? cc.s7 : cc.s3;
                const int y = 2 * (r + k) - 3;
                typedef unsigned short up_h2 __attribute__((ext_vector_type(2)));
                uint32_t o0[4], o1[4];
                up_v8h(o0, ring[(k + 1) % 3], ring[k % 3], fb01, fb23, kround, vsh, maxpk);
                uint32_t off = doff;
                asm volatile("" : "+v"(off));
#pragma unroll
                for (int i = 0; i < 4; i++)
                    o0[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(up_h2, o0[i]) << (unsigned short)dmsb);
                if (live && y >= 0) {
                    up_u4 st; st.x = o0[0]; st.y = o0[1]; st.z = o0[2]; st.w = o0[3];
                    if (NTS) __builtin_nontemporal_store(st, (up_g4)((up_gp)dr + off));
                    else *(up_g4)((up_gp)dr + off) = st;
                }
                up_v8h(o1, ring[(k + 1) % 3], ring[k % 3], fa01, fa23, kround, vsh, maxpk);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    o1[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(up_h2, o1[i]) << (unsigned short)dmsb);
                if (live && y + 1 < dstH) {
                    up_u4 st; st.x = o1[0]; st.y = o1[1]; st.z = o1[2]; st.w = o1[3];
                    if (NTS) __builtin_nontemporal_store(st, (up_g4)((up_gp)(dr + dstride) + off));
                    else *(up_g4)((up_gp)(dr + dstride) + off) = st;
                }
                dr += 2 * dstride;
                asm("" : "+s"(dr));
                load_next(w);
                continue;
            }
            if (r + k < b) { /* uniform (the straight-line form measured the same here: tools/ab_headline.py) */
                if (!DBG_COPY)
                    hpass(w, ring[k % 3]);
                const up_u8 cc = c8[k >> 1];
                const uint32_t fb01 = k & 1 ? cc.s4 : cc.s0, fb23 = k & 1 ? cc.s5 : cc.s1;
                const uint32_t fa01 = k & 1 ? cc.s6 : cc.s2, fa23 = k & 1 ? cc.s7 : cc.s3;
                const int y = 2 * (r + k) - 3;
                uint32_t w0, w1, a0, a1;
                if (DBG_COPY) { w0 = w.q[0] + fb01; w1 = w.q[1] + fb23; a0 = w.q[1] + fa01; a1 = w.q[2] + fa23; }
                else up_v8(w0, w1, ring[(k + 1) % 3], ring[k % 3], fb01, fb23, kround);
                uint32_t off = doff;
                asm volatile("" : "+v"(off));
                if (act && y >= 0 && (!DBG_NOST || (w0 == 0x12345678u && w1 == 0x9abcdef0u))) {
                    up_u2 s; s.x = w0; s.y = w1;
                    if (NTS) __builtin_nontemporal_store(s, (up_g2)((up_gp)dr + off));
                    else *(up_g2)((up_gp)dr + off) = s;
                }
                if (!DBG_COPY)
                    up_v8(a0, a1, ring[(k + 1) % 3], ring[k % 3], fa01, fa23, kround);
                if (act && y + 1 < dstH && (!DBG_NOST || (a0 == 0x12345678u && a1 == 0x9abcdef0u))) {
                    up_u2 s; s.x = a0; s.y = a1;
                    if (NTS) __builtin_nontemporal_store(s, (up_g2)((up_gp)(dr + dstride) + off));
                    else *(up_g2)((up_gp)(dr + dstride) + off) = s;
                }
                dr += 2 * dstride;
                asm("" : "+s"(dr));
            }
            load_next(w);
        }
    }
}

template <int D, int VAR, int HB = 0, int RC = 0, int SC = 0> /* VAR: 0 the product; 16 / 48 / 64 measurement only (see up2_unit); HB: samples above 8 bits; RC: range conversion; SC: the horizontal bank in SGPRs */
__global__ __launch_bounds__(256) void k_sws_up2(FFHipUp2Args A)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    uint32_t blk = blockIdx.x;
    if (A.xcd == 1) {
        /* workgroup b runs on XCD b % 8 (observed, not promised: speed only).  Give every XCD one contiguous eighth of the
         * units, in order, so that the waves sharing source lines (adjacent column blocks, the 3 halo rows of adjacent
         * strips) meet in one L2. */
        const uint32_t nb = gridDim.x, x = blk & 7u, sl = blk >> 3, q = nb >> 3, r = nb & 7u;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + sl;
    } else if (A.xcd > 1) {
        /* measured variant (round 6): XCD-contiguous CHUNKS of 1 << (xcd - 1) workgroups dealt round-robin — every XCD still walks
         * contiguous units, but the eight XCDs' fronts stay within eight chunks of each other instead of an eighth of the batch apart */
        const uint32_t lg = (uint32_t)A.xcd - 1u, C = 1u << lg, nb = gridDim.x, full = nb & ~(8u * C - 1u);
        if (blk < full) {
            const uint32_t x = blk & 7u, sl = blk >> 3;
            blk = (((sl >> lg) << 3) + x) * C + (sl & (C - 1u));
        }
    }
    const uint32_t gw = blk * 4u + (uint32_t)wave;
    if (gw >= (uint32_t)A.units_per_pack * (uint32_t)A.npacks)
        return;
    const int pack = (int)(gw / (uint32_t)A.units_per_pack);
    const int u = (int)(gw - (uint32_t)pack * (uint32_t)A.units_per_pack);
    int j = 0;
    if (A.njobs > 1 && u >= A.job[1].unit_begin) j = 1;
    if (A.njobs > 2 && u >= A.job[2].unit_begin) j = 2;
    const FFHipUp2Job &J = A.job[j];
    const int local = u - J.unit_begin;
    /* units of a (pack, strip): the full 64-lane column blocks of each of the pack's frames, then the blocks at the ragged
     * right end of the rows, which the pack's frames share lane block by lane block */
    const int strip = local / J.upj, idx = local - strip * J.upj;
    const int nf = J.nfull << A.fshift;
    int frame0 = pack << A.fshift, fsh = 0, gbase;
    if (idx < nf) {
        const int fsub = idx / J.nfull;
        frame0 += fsub;
        gbase = (idx - fsub * J.nfull) * 64;
        if (frame0 >= A.nframes)
            return;
    } else {
        fsh = A.fshift;
        gbase = J.nfull * 64 + (idx - nf) * (64 >> fsh);
    }
    if (J.pair)
        up2_unit<1, D, VAR, HB, RC, SC>(J, frame0, fsh, gbase, strip, lane, A.nframes);
    else
        up2_unit<0, D, VAR, HB, RC, SC>(J, frame0, fsh, gbase, strip, lane, A.nframes);
}

/* ================================================================================================== */
/* host side */

/*
 * Re-express a 4-tap bank of an exact 2x up-scale as coefficients on the REGULAR windows of the edge-replicated row:
 * output x reads samples clamp(s0 + k), s0 = (x >> 1) - 2 + (x & 1), k = 0..3.  Every non-zero tap of the bank row must
 * sit on one of those samples; taps the reference folded onto the edge sample land on one of the replicas.  Output: n_dst x 2
 * dwords, (c0, c1) (c2, c3) as int16 pairs.  Returns 0 when the bank is not of this shape.
 */
int ffhip_up2_virtual_bank(const int16_t *filter, const int32_t *pos, int n_dst, int n_src, std::vector<uint32_t> *out)
{
    if (n_dst != 2 * n_src || n_src < 4)
        return 0;
    out->assign((size_t)n_dst * 2, 0);
    for (int x = 0; x < n_dst; x++) {
        const int s0 = (x >> 1) - 2 + (x & 1);
        int16_t v[4] = { 0, 0, 0, 0 };
        bool used[4] = { false, false, false, false };
        for (int i = 0; i < 4; i++) {
            const int16_t c = filter[(size_t)x * 4 + i];
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
        (*out)[2 * (size_t)x] = (uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16);
        (*out)[2 * (size_t)x + 1] = (uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
    }
    return 1;
}

/* the virtual horizontal bank as scalars: 1 when every column but the three next to either end carries the two dwords of its parity
 * (out[0..3] = an even and an odd column, [4..9] columns 0..2, [10..15] the last three) */
int ffhip_up2_hco(const std::vector<uint32_t> &v, uint32_t out[16])
{
    const int n = (int)(v.size() / 2);
    if (n < 16 || (n & 1))
        return 0;
    for (int x = 3; x < n - 3; x++)
        if (v[2 * (size_t)x] != v[2 * (size_t)(4 + (x & 1))] || v[2 * (size_t)x + 1] != v[2 * (size_t)(4 + (x & 1)) + 1])
            return 0;
    for (int i = 0; i < 4; i++)
        out[i] = v[8 + i];
    for (int i = 0; i < 6; i++) {
        out[4 + i] = v[i];
        out[10 + i] = v[2 * (size_t)(n - 3) + i];
    }
    return 1;
}

/* strips of `want` steps (a multiple of 6: the row loop is unrolled six times), evened out over the plane */
void ffhip_up2_plan_job(FFHipUp2Job *j, int lanes_per_frame, int want)
{
    const int steps = j->srcH + 1;
    int n = cdiv(steps, want);
    int s = cdiv(cdiv(steps, n), 6) * 6;
    j->steps_per_strip = s;
    j->nstrips = cdiv(steps, s);
    if (lanes_per_frame == 64) {
        j->nfull = 0; /* one frame per wave throughout: every block is a "tail" block of 64 lanes */
        j->upj = cdiv(j->ngroups, 64);
    } else {
        j->nfull = j->ngroups / 64;
        j->upj = j->nfull * (64 / lanes_per_frame) + cdiv(j->ngroups - j->nfull * 64, lanes_per_frame);
    }
}

int ffhip_launch_up2(FFHipUp2Args &A, int depth, int var, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    int u = 0;
    for (int i = 0; i < A.njobs; i++) {
        A.job[i].unit_begin = u;
        u += A.job[i].upj * A.job[i].nstrips;
    }
    A.units_per_pack = u;
    A.npacks = (A.nframes + (1 << A.fshift) - 1) >> A.fshift;
    const long long waves = (long long)u * A.npacks;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (A.job[0].hb_sdepth) { /* samples above 8 bits: plain or (var & 1) non-temporal stores */
        if (depth == 3 && !(var & 1))      hipLaunchKernelGGL((k_sws_up2<3, 0, 1>), grid, block, 0, stream, A);
        else if (depth == 3)               hipLaunchKernelGGL((k_sws_up2<3, 1, 1>), grid, block, 0, stream, A);
        else if (!(var & 1))               hipLaunchKernelGGL((k_sws_up2<6, 0, 1>), grid, block, 0, stream, A);
        else                               hipLaunchKernelGGL((k_sws_up2<6, 1, 1>), grid, block, 0, stream, A);
        LAUNCH_CHECK();
        return 0;
    }
    if (A.job[0].rc_coeff) { /* range conversion between the passes: rounds 4-5's form, or (var 3) the bank in SGPRs + non-temporal stores */
        bool scr = var == 3;
        for (int i = 0; i < A.njobs; i++)
            scr = scr && A.job[i].hco_ok;
        if (scr && depth == 3)             hipLaunchKernelGGL((k_sws_up2<3, 1, 0, 1, 1>), grid, block, 0, stream, A);
        else if (scr)                      hipLaunchKernelGGL((k_sws_up2<6, 1, 0, 1, 1>), grid, block, 0, stream, A);
        else if (depth == 3)               hipLaunchKernelGGL((k_sws_up2<3, 0, 0, 1>), grid, block, 0, stream, A);
        else                               hipLaunchKernelGGL((k_sws_up2<6, 0, 0, 1>), grid, block, 0, stream, A);
        LAUNCH_CHECK();
        return 0;
    }
    bool sc = true;
    for (int i = 0; i < A.njobs; i++)
        sc = sc && A.job[i].hco_ok;
    if (var == 2 || var == 3) { /* measured variants (round 6): the bank in SGPRs (2), with non-temporal stores as well (3); depth as asked */
        if (!sc)
            var &= 1;
        else {
            if (var == 2 && depth == 3) hipLaunchKernelGGL((k_sws_up2<3, 0, 0, 0, 1>), grid, block, 0, stream, A);
            else if (var == 2)          hipLaunchKernelGGL((k_sws_up2<6, 0, 0, 0, 1>), grid, block, 0, stream, A);
            else if (depth == 3)        hipLaunchKernelGGL((k_sws_up2<3, 1, 0, 0, 1>), grid, block, 0, stream, A);
            else                        hipLaunchKernelGGL((k_sws_up2<6, 1, 0, 0, 1>), grid, block, 0, stream, A);
            LAUNCH_CHECK();
            return 0;
        }
    }
#define UP2_LAUNCH(DD, VV) hipLaunchKernelGGL((k_sws_up2<DD, VV>), grid, block, 0, stream, A)
#define UP2_CASE(VV) case VV: if (depth == 3) UP2_LAUNCH(3, VV); else UP2_LAUNCH(6, VV); break
    switch (var) {
    UP2_CASE(0);
    UP2_CASE(1); /* non-temporal stores */
    UP2_CASE(16); UP2_CASE(48); UP2_CASE(64); /* measurement only */
    default:
        ffhip_set_error("ffhip_sws: exact-2x kernel variant %d is not built", var);
        return FFHIP_EINVAL;
    }
#undef UP2_CASE
#undef UP2_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
