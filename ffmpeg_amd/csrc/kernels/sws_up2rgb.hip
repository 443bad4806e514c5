/*
 * sws_up2rgb.hip — the fused scaler + packed-RGB writer for EXACT 2x up-scaling of 4:2:0 (yuv420p, NV12, NV21) with 4-tap banks on all four
 * axes (bicubic / bilinear 1080p -> 4K rgb24): hScale8To15_c (libswscale/swscale.c:128-142) on Y, U and V, then
 * yuv2rgb_X_c_template / yuv2rgb_write (libswscale/output.c:1789-1840, 1663-1787) as yuv2packedX reaches them (vscale.c:126-170).
 * Same bytes as k_sws_colwalk_rgb / k_scale_rgb (tests/test_gpu_sws_fast.py), which stay the kernels of every other geometry.
 *
 * What exact 2x buys over the general column walker with RGB output (k_sws_colwalk_rgb: 25 VALU instructions per pixel, VALU-bound):
 *
 *  - REGULAR windows on all four axes, as in sws_up2.hip: luma column x reads source samples (x >> 1) - 2 + (x & 1) .. + 3 of the
 *    edge-replicated row, chroma column the same at half the width; luma row y the rows (y >> 1) - 2 + (y & 1) .. + 3.  A packed-RGB
 *    target has a chroma line per output line (chrDstH == dstH, libswscale/utils.c:1560-1575), so 4:2:0 chroma goes up FOUR times
 *    vertically: row y reads chroma rows ((y + 2) >> 2) - 2 .. + 1, four consecutive output rows share a window.  The host re-expresses
 *    every bank row on those windows (ffhip_upn_virtual_bank; a bank that does not fit keeps the context on the walker).  No position
 *    tables, no per-lane byte selectors, no `need` bookkeeping, no v_readlane: horizontal coefficients sit in VGPRs, the four vertical
 *    coefficient dwords of an output row arrive by scalar loads (one s_load_dwordx8 per source row = two output rows).
 *  - A STATIC schedule: luma source row r completes output rows 2r-3 and 2r-2; chroma row c completes 4c-6 .. 4c-3, i.e. the chroma
 *    ring advances once every second luma row, between the two output rows of an even one.  Six luma rows per loop trip = three chroma
 *    rows: every ring index is a compile-time constant, the rings never move.
 *  - The writer: U and V are clipped and packed by ONE v_ashr_pk_u8_i32, the two bytes address the workgroup's LDS tables of the
 *    chroma terms (r(V), gv(V) / b(U), gu(U): 8 bytes each, see k_sws_colwalk_rgb) through SDWA byte selects; a component is
 *    v_mad_i32_i24(Y >> 19, cy, term) and two components leave as one v_ashr_pk_u8_i32 (>> 16, clipped), the second of a dword in
 *    the high-half form: no merging v_perm.
 *
 * Per 8 pixels of an output row: 16 + 16 vertical dots, 16 for the chroma tables, 8 + 24 + 12 for the writer = 92, plus
 * 39 / 2 (luma) + 42 / 4 (chroma) for the horizontal passes: ~15 VALU instructions per pixel.
 *
 * Geometry: a wave owns 64 lanes x 8 pixels (24 bytes) of a strip of rows.  A lane reads 12 luma bytes at 4g - 4 per luma row and
 * 8 + 8 chroma bytes at (2g - 2) & ~3 per chroma row (g = its group in the row), and writes 24 bytes of two output rows per luma row
 * — through the wave's 1.5 KiB of LDS, so that a store instruction covers 512 contiguous bytes of the row.
 */
#include <stdlib.h>
#include <vector>

#include "common.h"
#include "sws_kernels.h"

typedef uint32_t ur_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t ur_u3 __attribute__((ext_vector_type(3)));
typedef uint32_t ur_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t ur_u8 __attribute__((ext_vector_type(8)));
typedef ur_u3 __attribute__((aligned(4))) ur_u3a;
typedef ur_u2 __attribute__((aligned(4))) ur_u2a;
typedef const uint8_t __attribute__((address_space(1))) *ur_gcp;
typedef uint8_t __attribute__((address_space(1))) *ur_gp;
typedef const ur_u3a __attribute__((address_space(1))) *ur_gc3;
typedef const ur_u2a __attribute__((address_space(1))) *ur_gc2;
typedef ur_u2 __attribute__((address_space(1))) *ur_g2;
typedef ur_u4 __attribute__((address_space(1))) *ur_g4;
typedef const ur_u8 __attribute__((address_space(4))) *ur_cc8; /* constant address space: scalar loads */
typedef const uint32_t __attribute__((address_space(4))) *ur_cc1;

/* four horizontal samples of adjacent columns (even, odd, even, odd): d[i] = (pa[i] . c01 + pb[i] . c23) >> 7 with the coefficient pairs
 * of the column's parity in SGPRs — away from the row's ends an exact-2x bank has two rows of coefficients, not one per column.  (The
 * block of sws_up2.hip: every DOT result is consumed >= 3 instructions after it was written.) */
__device__ __forceinline__ void ur_h4(int (&d)[4], uint32_t pa0, uint32_t pa1, uint32_t pa2, uint32_t pa3, uint32_t pb0,
                                      uint32_t pb1, uint32_t pb2, uint32_t pb3, uint32_t e01, uint32_t e23, uint32_t o01, uint32_t o23)
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
        : "v"(pa0), "v"(pa1), "v"(pa2), "v"(pa3), "v"(pb0), "v"(pb1), "v"(pb2), "v"(pb3), "s"(e01), "s"(e23), "s"(o01), "s"(o23));
}
/* the three columns next to a row's end, whose coefficients the reference renormalised after folding the taps onto the edge sample
 * (initFilter(), libswscale/utils.c:519-561): c = their six coefficient dwords (column i: c[2i], c[2i + 1]) */
__device__ __forceinline__ void ur_h3(int (&d)[3], uint32_t pa0, uint32_t pa1, uint32_t pa2, uint32_t pb0, uint32_t pb1, uint32_t pb2,
                                      const uint32_t (&c)[6])
{
    asm("v_dot2_i32_i16 %0, %3, %9, 0\n\t"
        "v_dot2_i32_i16 %1, %4, %11, 0\n\t"
        "v_dot2_i32_i16 %2, %5, %13, 0\n\t"
        "s_nop 0\n\t"
        "v_dot2_i32_i16 %0, %6, %10, %0\n\t"
        "v_dot2_i32_i16 %1, %7, %12, %1\n\t"
        "v_dot2_i32_i16 %2, %8, %14, %2\n\t"
        "s_nop 1\n\t"
        "v_ashrrev_i32 %0, 7, %0\n\t"
        "v_ashrrev_i32 %1, 7, %1\n\t"
        "v_ashrrev_i32 %2, 7, %2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2])
        : "v"(pa0), "v"(pa1), "v"(pa2), "v"(pb0), "v"(pb1), "v"(pb2), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]));
}

/* the sixteen vertical dots of 8 sums t[i] = seed + pa[i] . f01 + pb[i] . f23 (the pairs are (row, row + 1) int16 halves; f in SGPRs).
 * The consumers of the sums sit in the same block, >= 3 instructions behind the DOT that wrote their operand (the gfx950 DOT hazard:
 * nothing the compiler schedules may read a DOT result first). */
#define UR_DOTS16                                   \
        "v_dot2_i32_i16 %8, %16, %32, %34\n\t"      \
        "v_dot2_i32_i16 %9, %17, %32, %34\n\t"      \
        "v_dot2_i32_i16 %10, %18, %32, %34\n\t"     \
        "v_dot2_i32_i16 %11, %19, %32, %34\n\t"     \
        "v_dot2_i32_i16 %12, %20, %32, %34\n\t"     \
        "v_dot2_i32_i16 %13, %21, %32, %34\n\t"     \
        "v_dot2_i32_i16 %14, %22, %32, %34\n\t"     \
        "v_dot2_i32_i16 %15, %23, %32, %34\n\t"     \
        "v_dot2_i32_i16 %8, %24, %33, %8\n\t"       \
        "v_dot2_i32_i16 %9, %25, %33, %9\n\t"       \
        "v_dot2_i32_i16 %10, %26, %33, %10\n\t"     \
        "v_dot2_i32_i16 %11, %27, %33, %11\n\t"     \
        "v_dot2_i32_i16 %12, %28, %33, %12\n\t"     \
        "v_dot2_i32_i16 %13, %29, %33, %13\n\t"     \
        "v_dot2_i32_i16 %14, %30, %33, %14\n\t"     \
        "v_dot2_i32_i16 %15, %31, %33, %15\n\t"
#define UR_DOTS16_OPERANDS                                                                                              \
        "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)                           \
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pa[6]), "v"(pa[7]),                \
          "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pb[4]), "v"(pb[5]), "v"(pb[6]), "v"(pb[7]),                \
          "s"(f01), "s"(f23), "v"(seed)

/* luma: y[i] = t[i] >> 19 (yuv2rgb_X does not clip Y: the tables have head room, libswscale/output.c:1814-1835) */
__device__ __forceinline__ void ur_vy8(int (&y)[8], const uint32_t (&pa)[8], const uint32_t (&pb)[8], uint32_t f01, uint32_t f23, int seed)
{
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm(UR_DOTS16
        "v_ashrrev_i32 %0, 19, %8\n\t"
        "v_ashrrev_i32 %1, 19, %9\n\t"
        "v_ashrrev_i32 %2, 19, %10\n\t"
        "v_ashrrev_i32 %3, 19, %11\n\t"
        "v_ashrrev_i32 %4, 19, %12\n\t"
        "v_ashrrev_i32 %5, 19, %13\n\t"
        "v_ashrrev_i32 %6, 19, %14\n\t"
        "v_ashrrev_i32 %7, 19, %15"
        : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4]), "=&v"(y[5]), "=&v"(y[6]), "=&v"(y[7]),
          UR_DOTS16_OPERANDS);
}
/* chroma: pa / pb [0..3] U, [4..7] V; uv[m] = clip_u8(U[m] >> 19) | clip_u8(V[m] >> 19) << 8 in the low half (the table index is the
 * clipped value: fill_table(), libswscale/yuv2rgb.c:700-712) */
__device__ __forceinline__ void ur_vc4(uint32_t (&uv)[4], const uint32_t (&pa)[8], const uint32_t (&pb)[8], uint32_t f01, uint32_t f23, int seed)
{
    int t0, t1, t2, t3, t4, t5, t6, t7;
    uint32_t d4, d5, d6, d7;
    asm(UR_DOTS16
        "v_ashr_pk_u8_i32 %0, %8, %12, 19\n\t"
        "v_ashr_pk_u8_i32 %1, %9, %13, 19\n\t"
        "v_ashr_pk_u8_i32 %2, %10, %14, 19\n\t"
        "v_ashr_pk_u8_i32 %3, %11, %15, 19"
        : "=&v"(uv[0]), "=&v"(uv[1]), "=&v"(uv[2]), "=&v"(uv[3]), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7),
          UR_DOTS16_OPERANDS);
}

/* one dword of four clipped bytes (a, b, c, d) >> 16: the second instruction writes the high half and keeps the low one */
__device__ __forceinline__ uint32_t ur_pk4(int a, int b, int c, int d)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 16 op_sel:[0,0,0,1]"
        : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}
/* v_mad_i32_i24 with the multiplier in an SGPR: one instruction per colour component */
__device__ __forceinline__ int ur_mad24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

__device__ __forceinline__ void ur_wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct UrRawL { uint32_t q[3]; };
struct UrRawC { uint32_t u[2], v[2]; };

/* LAY: 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra (alpha = 255).  ST: 0 direct stores, 1 / 2 through the LDS transposer in 8- /
 * 16-byte pieces.  NTS: non-temporal stores of the picture.  SIL: the chroma plane is byte-interleaved (NV12; A.swap: NV21). */
template <int LAY, int ST, bool NTS, bool SIL = false>
__global__ __launch_bounds__(256, 4) void k_sws_up2_rgb(FFHipUp2RgbArgs A)
{
    __shared__ __attribute__((aligned(16))) uint32_t tiles[4][LAY < 2 ? 384 : 512];
    __shared__ uint2 lut[512]; /* [U] = { b(U), gu(U) }, [256 + V] = { r(V), gv(V) }: the chroma terms of the closed form, cy-scaled, rounding in */
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
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave;
    /* units: (pack of A.fpp frames, strip, lane block).  The lanes of a pack's blocks run through the groups of its frames one after
     * the other — lane index L = 64 * block + lane is group L % G of the pack's frame L / G — so a row width that is not a multiple of
     * 64 groups leaves no lane idle (3840 pixels = 480 groups: two frames fill 15 waves); a wave may hold the end of one frame's row
     * and the start of the next one's */
    const uint32_t upp = (uint32_t)A.wpp * (uint32_t)A.nstrips;
    if (gw >= upp * (uint32_t)A.npacks)
        return;
    const int pack = (int)(gw / upp);
    const int u = (int)(gw - (uint32_t)pack * upp);
    const int strip = u / A.wpp, cb = u - strip * A.wpp;

    const int G = A.ngroups;
    const int f0 = pack * A.fpp;                          /* the pack's first frame */
    const int nf = min(A.fpp, A.nframes - f0);            /* its frames that exist */
    const int Lraw = cb * 64 + lane;
    const bool act = Lraw < nf * G;
    const int L = min(Lraw, nf * G - 1);                  /* idle lanes shadow the last one */
    const int fs = (L >= G) + (L >= 2 * G) + (L >= 3 * G); /* A.fpp <= 4 */
    const int g = L - fs * G;
    const bool lb = g == 0, rb = g == G - 1;
    const bool first = __builtin_amdgcn_ballot_w64(lb) != 0, last = __builtin_amdgcn_ballot_w64(rb) != 0; /* wave-uniform */
    const bool border = first || last;
    const uint32_t yoff = (uint32_t)(lb ? 0 : rb ? 4 * g - 8 : 4 * g - 4);
    const uint32_t coff = SIL ? yoff /* (u, v) pairs 2g - 2 .. 2g + 3: 12 bytes at 4g - 4, as the luma span */
                              : (uint32_t)(lb ? 0 : rb ? 2 * G - 8 : (2 * g - 2) & ~3);
    const uint32_t soffY = (uint32_t)fs * (uint32_t)A.sfp[0] + yoff;
    const uint32_t soffU = (uint32_t)fs * (uint32_t)A.sfp[1] + coff;
    const uint32_t soffV = SIL ? soffU : (uint32_t)fs * (uint32_t)A.sfp[2] + coff;

    /* horizontal coefficients (virtual banks: regular windows of the replicated rows), all wave-uniform: A.hco = luma (even c01, c23,
     * odd c01, c23), chroma the same, then the six dwords of the three columns at the left / right end of a luma row and of a chroma
     * row — read by scalar loads; the end sets only by the waves that hold a row's first / last lane */
    const ur_cc1 hco = (ur_cc1)A.hco;
    const uint32_t LE01 = hco[0], LE23 = hco[1], LO01 = hco[2], LO23 = hco[3];
    const uint32_t CE01 = hco[4], CE23 = hco[5], CO01 = hco[6], CO23 = hco[7];
    /* chroma byte selectors: sample j of the lane's six (2g - 2 + j, replicated at the row's ends) is byte o + j of its 8 bytes,
     * pair j = (sample j, sample j + 1) as int16s */
    uint32_t csel[5];
    {
        const uint32_t o = (uint32_t)(2 * g - 2) & 3u;
#pragma unroll
        for (int j = 0; j < 5; j++)
            csel[j] = 0x0c000c00u | (o + j) | ((o + j + 1) << 16);
    }

    const int S = A.steps_per_strip;
    const int a = 1 + strip * S, b = min(a + S, A.srcH + 2); /* luma step r emits rows 2r-3 and 2r-2 */
    const int srcH = A.srcH, chrH = A.srcH >> 1, dstH = 2 * A.srcH;
    const uint8_t *sy = A.src[0] + (size_t)f0 * A.sfp[0];
    const uint8_t *su = A.src[1] + (size_t)f0 * A.sfp[1];
    const uint8_t *sv = SIL ? su : A.src[2] + (size_t)f0 * A.sfp[2];
    const ptrdiff_t ystride = A.sstride[0], ustride = A.sstride[1], vstride = SIL ? A.sstride[1] : A.sstride[2], dstride = A.dstride;

    int pr = a - 3; /* next luma row to fetch (unclamped) */
    const uint8_t *pfy = sy + (ptrdiff_t)min(max(pr, 0), srcH - 1) * ystride;
    int cr = (a + 1) / 2 - 3; /* next chroma row to fetch (unclamped): the first step's group is c = (a + 1) / 2, its window c-3 .. c */
    const uint8_t *pfu = su + (ptrdiff_t)min(max(cr, 0), chrH - 1) * ustride;
    const uint8_t *pfv = sv + (ptrdiff_t)min(max(cr, 0), chrH - 1) * vstride;
    uint8_t *dr = A.dst + (size_t)f0 * A.dfp + (ptrdiff_t)(2 * a - 3) * dstride; /* row 2a-3 (row -1 of the first strip is never stored) */
    asm("" : "+s"(pfy), "+s"(pfu), "+s"(pfv), "+s"(dr));

    auto load_luma = [&](UrRawL &o) {
        uint32_t off = soffY;
        asm volatile("" : "+v"(off)); /* keeps `uniform base + zext(lane offset)` next to the access: saddr addressing */
        const ur_u3 w = *(ur_gc3)((ur_gcp)pfy + off);
        o.q[0] = w.x; o.q[1] = w.y; o.q[2] = w.z;
        pr++;
        pfy += (pr >= 1 && pr <= srcH - 1) ? ystride : 0; /* rows above / below the plane replicate the edge row */
        asm("" : "+s"(pfy));
    };
    auto load_chroma = [&](UrRawC &o) {
        uint32_t off = soffU, offv = soffV;
        asm volatile("" : "+v"(off), "+v"(offv));
        if (SIL) {
            const ur_u3 w = *(ur_gc3)((ur_gcp)pfu + off);
            o.u[0] = w.x; o.u[1] = w.y; o.v[0] = w.z; o.v[1] = 0;
        } else {
            const ur_u2 wu = *(ur_gc2)((ur_gcp)pfu + off);
            const ur_u2 wv = *(ur_gc2)((ur_gcp)pfv + offv);
            o.u[0] = wu.x; o.u[1] = wu.y; o.v[0] = wv.x; o.v[1] = wv.y;
        }
        cr++;
        const bool adv = cr >= 1 && cr <= chrH - 1;
        pfu += adv ? ustride : 0;
        if (!SIL)
            pfv += adv ? vstride : 0;
        asm("" : "+s"(pfu), "+s"(pfv));
    };

    uint32_t ring[3][8];   /* luma: (h[r-1], h[r]) pairs of the last three rows */
    uint32_t cring[3][8];  /* chroma: the same, [0..3] U, [4..7] V */
    int hprev[8], cprev[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        hprev[i] = cprev[i] = 0;
    int kround = A.vround;
    asm volatile("" : "+v"(kround));

    /* horizontal pass of one luma row: 8 samples appended to the ring (int16-saturated pack == min(., 32767): no sum of the bank falls
     * below -32768, host-checked) */
    auto hpassL = [&](const UrRawL &w, uint32_t (&Pnew)[8]) {
        uint32_t v0 = w.q[0], v1 = w.q[1], v2 = w.q[2];
        if (border) { /* the first / last lane of a row loaded its span one dword further inside */
            const uint32_t f0 = __builtin_amdgcn_perm(w.q[0], w.q[0], 0x00000000u), f2 = __builtin_amdgcn_perm(w.q[2], w.q[2], 0x03030303u);
            v0 = lb ? f0 : rb ? w.q[1] : w.q[0];
            v1 = lb ? w.q[0] : rb ? w.q[2] : w.q[1];
            v2 = lb ? w.q[1] : rb ? f2 : w.q[2];
        }
        /* samples b0..b7 = bytes 2..9 of the span */
        const uint32_t p0 = __builtin_amdgcn_perm(v1, v0, 0x0c030c02u), p1 = __builtin_amdgcn_perm(v1, v0, 0x0c040c03u);
        const uint32_t p2 = __builtin_amdgcn_perm(v1, v0, 0x0c050c04u), p3 = __builtin_amdgcn_perm(v1, v0, 0x0c060c05u);
        const uint32_t p4 = __builtin_amdgcn_perm(v1, v0, 0x0c070c06u), p5 = __builtin_amdgcn_perm(v2, v1, 0x0c040c03u);
        const uint32_t p6 = __builtin_amdgcn_perm(v2, v1, 0x0c050c04u);
        int hl[4], hh[4];
        ur_h4(hl, p0, p1, p1, p2, p2, p3, p3, p4, LE01, LE23, LO01, LO23);
        ur_h4(hh, p2, p3, p3, p4, p4, p5, p5, p6, LE01, LE23, LO01, LO23);
        if (first) { /* columns 0..2 of the row: lane 0's */
            const uint32_t c[6] = { hco[8], hco[9], hco[10], hco[11], hco[12], hco[13] };
            int d[3];
            ur_h3(d, p0, p1, p1, p2, p3, p3, c);
#pragma unroll
            for (int i = 0; i < 3; i++)
                hl[i] = lb ? d[i] : hl[i];
        }
        if (last) { /* the last three columns: columns 5..7 of the last lane */
            const uint32_t c[6] = { hco[14], hco[15], hco[16], hco[17], hco[18], hco[19] };
            int d[3];
            ur_h3(d, p3, p3, p4, p5, p5, p6, c);
#pragma unroll
            for (int i = 0; i < 3; i++)
                hh[1 + i] = rb ? d[i] : hh[1 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            Pnew[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[i], hl[i]));
            hprev[i] = hl[i];
            Pnew[4 + i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[4 + i], hh[i]));
            hprev[4 + i] = hh[i];
        }
    };
    /* the same for one chroma row: 4 U + 4 V samples */
    auto hpassC = [&](const UrRawC &w, uint32_t (&Pnew)[8]) {
        uint32_t i0 = w.u[0], i1 = w.u[1], i2 = w.v[0]; /* SIL: the 12 bytes of six (u, v) pairs */
        if (SIL && border) { /* the first / last lane of a row loaded its span one dword further inside: replicate the end pair */
            const uint32_t f0 = __builtin_amdgcn_perm(i0, i0, 0x01000100u), f2 = __builtin_amdgcn_perm(i2, i2, 0x03020302u);
            const uint32_t q0 = i0, q1 = i1, q2 = i2;
            i0 = lb ? f0 : rb ? q1 : q0;
            i1 = lb ? q0 : rb ? q2 : q1;
            i2 = lb ? q1 : rb ? f2 : q2;
        }
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            uint32_t p0, p1, p2, p3, p4;
            if (SIL) {
                /* a channel's samples sit at every second byte: pair j = bytes (2j, 2j + 2), + 1 for the channel at the odd bytes */
                const uint32_t odd = (ch != 0) != (A.swap != 0) ? 0x00010001u : 0u;
                const uint32_t s0 = 0x0c020c00u + odd, s1 = 0x0c040c02u + odd, s2 = 0x0c060c04u + odd;
                p0 = __builtin_amdgcn_perm(i1, i0, s0); p1 = __builtin_amdgcn_perm(i1, i0, s1); p2 = __builtin_amdgcn_perm(i1, i0, s2);
                p3 = __builtin_amdgcn_perm(i2, i1, s1); p4 = __builtin_amdgcn_perm(i2, i1, s2);
            } else {
                const uint32_t q0 = ch ? w.v[0] : w.u[0], q1 = ch ? w.v[1] : w.u[1];
                uint32_t v0 = q0, v1 = q1;
                if (border) { /* left: samples -2, -1 are sample 0; right: the lane loaded the row's last 8 bytes, samples 4, 5 are the last one */
                    v0 = lb ? __builtin_amdgcn_perm(q0, q0, 0x00000000u) : rb ? q1 : q0;
                    v1 = lb ? q0 : rb ? __builtin_amdgcn_perm(q1, q1, 0x03030303u) : q1;
                }
                p0 = __builtin_amdgcn_perm(v1, v0, csel[0]); p1 = __builtin_amdgcn_perm(v1, v0, csel[1]);
                p2 = __builtin_amdgcn_perm(v1, v0, csel[2]); p3 = __builtin_amdgcn_perm(v1, v0, csel[3]);
                p4 = __builtin_amdgcn_perm(v1, v0, csel[4]);
            }
            int h[4];
            ur_h4(h, p0, p1, p1, p2, p2, p3, p3, p4, CE01, CE23, CO01, CO23);
            if (first) {
                const uint32_t c[6] = { hco[20], hco[21], hco[22], hco[23], hco[24], hco[25] };
                int d[3];
                ur_h3(d, p0, p1, p1, p2, p3, p3, c);
#pragma unroll
                for (int i = 0; i < 3; i++)
                    h[i] = lb ? d[i] : h[i];
            }
            if (last) { /* columns 1..3 of the last lane */
                const uint32_t c[6] = { hco[26], hco[27], hco[28], hco[29], hco[30], hco[31] };
                int d[3];
                ur_h3(d, p1, p1, p2, p3, p3, p4, c);
#pragma unroll
                for (int i = 0; i < 3; i++)
                    h[1 + i] = rb ? d[i] : h[1 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Pnew[4 * ch + i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(cprev[4 * ch + i], h[i]));
                cprev[4 * ch + i] = h[i];
            }
        }
    };

    const int cy = __builtin_amdgcn_readfirstlane(A.k.cy);
    uint32_t *tile = tiles[wave];
    constexpr int NW = LAY < 2 ? 6 : 8; /* dwords of a lane's 8 pixels */
    const bool fullw = cb * 64 + 64 <= nf * G;                         /* wave-uniform: every lane has a group */
    const uint32_t dcol = (uint32_t)fs * (uint32_t)A.dfp + 4u * NW * (uint32_t)g; /* this lane's pixels from the pack's row pointer */
    const char *lutb = reinterpret_cast<const char *>(lut);
    /* the transposer's pieces: the bytes at offset o of the wave's tile are lane o / (4 NW)'s — where they go from the pack's row
     * pointer (that lane's frame and group; frames meet at multiples of 48 bytes: G is even), and whether that lane has a group */
    constexpr int NP = ST == 2 ? 2 : NW / 2;
    uint32_t poff[NP];
    bool pok[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const uint32_t o = ST == 2 ? (i == 0 ? 16u * (uint32_t)lane : 1024u + (LAY >= 2 ? 16u : 8u) * (uint32_t)lane)
                                   : 512u * (uint32_t)i + 8u * (uint32_t)lane;
        const uint32_t ls = o / (4u * NW);
        const int Ls = cb * 64 + (int)ls;
        const int fl = (Ls >= G) + (Ls >= 2 * G) + (Ls >= 3 * G);
        pok[i] = Ls < nf * G;
        poff[i] = (uint32_t)fl * (uint32_t)A.dfp + 4u * NW * (uint32_t)(Ls - fl * G) + (o - 4u * NW * ls);
    }

    auto st16 = [&](ur_gp d, const ur_u4 &v) {
        if (NTS) __builtin_nontemporal_store(v, (ur_g4)d);
        else *(ur_g4)d = v;
    };
    auto st8 = [&](ur_gp d, const ur_u2 &v) {
        if (NTS) __builtin_nontemporal_store(v, (ur_g2)d);
        else *(ur_g2)d = v;
    };
    /* the wave's row segment through the tile: a lane writes its 24 / 32 bytes, reads the 16 at 16 * lane (and 1024 + 16 * lane; rgb24:
     * the 8 at 1024 + 8 * lane) */
    auto tile_write = [&](const uint32_t (&w)[NW]) {
        uint32_t *t = tile + lane * NW;
        if (LAY >= 2) {
            *reinterpret_cast<uint4 *>(t) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4 *>(t + 4) = make_uint4(w[4 % NW], w[5 % NW], w[6 % NW], w[7 % NW]);
        } else {
            *reinterpret_cast<uint2 *>(t) = make_uint2(w[0], w[1]);
            *reinterpret_cast<uint2 *>(t + 2) = make_uint2(w[2], w[3]);
            *reinterpret_cast<uint2 *>(t + 4) = make_uint2(w[4], w[5]);
        }
    };
    auto tile_read = [&](uint4 &q0, uint4 &q1, uint2 &q2) {
        q0 = *reinterpret_cast<const uint4 *>(tile + lane * 4);
        q1 = make_uint4(0, 0, 0, 0);
        q2 = make_uint2(0, 0);
        if (LAY >= 2)
            q1 = *reinterpret_cast<const uint4 *>(tile + 256 + lane * 4);
        else
            q2 = *reinterpret_cast<const uint2 *>(tile + 256 + lane * 2);
    };
    auto put_pieces = [&](uint8_t *drow, const uint4 &q0, const uint4 &q1, const uint2 &q2) {
        ur_gp d = (ur_gp)drow;
        ur_u4 v0, v1;
        v0.x = q0.x; v0.y = q0.y; v0.z = q0.z; v0.w = q0.w;
        v1.x = q1.x; v1.y = q1.y; v1.z = q1.z; v1.w = q1.w;
        ur_u2 v2;
        v2.x = q2.x; v2.y = q2.y;
        if (fullw || pok[0])
            st16(d + poff[0], v0);
        if (LAY >= 2) {
            if (fullw || pok[1 % NP])
                st16(d + poff[1 % NP], v1);
        } else if (fullw || pok[1 % NP]) {
            st8(d + poff[1 % NP], v2);
        }
    };

    /* one output row: La / Lb the luma pairs (rows s0, s0+1) (s0+2, s0+3), Ca / Cb the chroma ones; coefficient dwords in SGPRs */
    auto emit = [&](const uint32_t (&La)[8], const uint32_t (&Lb)[8], const uint32_t (&Ca)[8], const uint32_t (&Cb)[8], uint32_t lf01,
                    uint32_t lf23, uint32_t cf01, uint32_t cf23, bool store) {
        uint32_t uv[4];
        int ysh[8];
        ur_vc4(uv, Ca, Cb, cf01, cf23, kround);
        ur_vy8(ysh, La, Lb, lf01, lf23, kround);
        int c0[4], c1[4], c2[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint2 tu = *reinterpret_cast<const uint2 *>(lutb + ((uv[m] & 0xffu) << 3));
            const uint2 tv = *reinterpret_cast<const uint2 *>(lutb + 2048 + (((uv[m] >> 8) & 0xffu) << 3));
            constexpr bool BGR = LAY == 1 || LAY == 4 || LAY == 5;
            c0[m] = (int)(BGR ? tu.x : tv.x);
            c1[m] = (int)(tu.y + tv.y);
            c2[m] = (int)(BGR ? tv.x : tu.x);
        }
        int val[24];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            val[3 * p] = ur_mad24(ysh[p], cy, c0[p >> 1]);
            val[3 * p + 1] = ur_mad24(ysh[p], cy, c1[p >> 1]);
            val[3 * p + 2] = ur_mad24(ysh[p], cy, c2[p >> 1]);
        }
        uint32_t w[NW];
        if (LAY >= 2) {
            int alpha = 255 << 16;
            asm("" : "+v"(alpha));
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int x = val[3 * p], y = val[3 * p + 1], z = val[3 * p + 2]; /* (R, G, B) or, BGR layouts, (B, G, R) */
                w[p] = (LAY == 2 || LAY == 4) ? ur_pk4(alpha, x, y, z) : ur_pk4(x, y, z, alpha);
            }
        } else {
#pragma unroll
            for (int d = 0; d < 6; d++)
                w[d] = ur_pk4(val[4 * d], val[4 * d + 1], val[4 * d + 2], val[4 * d + 3]);
        }
        if (ST == 0) {
            /* direct: a lane's 24 / 32 contiguous bytes, 8 / 16 per store instruction */
            if (act && store) {
                ur_gp d = (ur_gp)dr + dcol;
                if (LAY >= 2) {
                    ur_u4 s0, s1;
                    s0.x = w[0]; s0.y = w[1]; s0.z = w[2]; s0.w = w[3];
                    s1.x = w[4 % NW]; s1.y = w[5 % NW]; s1.z = w[6 % NW]; s1.w = w[7 % NW];
                    st16(d, s0);
                    st16(d + 16, s1);
                } else {
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        ur_u2 v;
                        v.x = w[2 * i]; v.y = w[2 * i + 1];
                        st8(d + 8 * i, v);
                    }
                }
            }
            return;
        }
        /* transpose through the wave's 1.5 / 2 KiB of LDS: a store instruction covers 512 (ST 1) or 1024 (ST 2) contiguous bytes of
         * the row instead of 8 / 16 bytes in every 24 / 32 */
        tile_write(w);
        ur_wave_sync_lds();
        if (ST == 2) {
            uint4 q0, q1;
            uint2 q2;
            tile_read(q0, q1, q2);
            ur_wave_sync_lds();
            if (store) /* uniform */
                put_pieces(dr, q0, q1, q2);
        } else {
            uint2 q[NW / 2];
#pragma unroll
            for (int i = 0; i < NW / 2; i++)
                q[i] = *reinterpret_cast<const uint2 *>(tile + i * 128 + lane * 2);
            ur_wave_sync_lds();
            if (store) {
                ur_gp d = (ur_gp)dr;
#pragma unroll
                for (int i = 0; i < NW / 2; i++) {
                    ur_u2 v;
                    v.x = q[i].x; v.y = q[i].y;
                    if (fullw || pok[i % NP])
                        st8(d + poff[i % NP], v);
                }
            }
        }
    };

    /* ---- prologue: luma rows a-3 .. a-1 and chroma rows c-3 .. c into the rings, the next ones in flight ---- */
    UrRawL buf[3];
    UrRawC cnext;
#pragma unroll
    for (int k = 0; k < 3; k++)
        load_luma(buf[k]);
    load_chroma(cnext);
    {
        uint32_t seed[8];
        UrRawC cur = cnext;
        load_chroma(cnext);
        hpassC(cur, seed); /* row c-3: only its samples matter (the low halves of the next pairs) */
#pragma unroll
        for (int k = 0; k < 3; k++) {
            cur = cnext;
            load_chroma(cnext);
            hpassC(cur, cring[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        hpassL(buf[k], ring[k]);
        load_luma(buf[k]);
    }

    /* vertical coefficients: row y at dwords 4 (y + 1) .. + 3 = (luma 01, luma 23, chroma 01, chroma 23); a step reads rows
     * 2r-3, 2r-2 = 8 consecutive dwords from 8r - 8 */
    const uint32_t *vt = A.vt;
    for (int r = a; r < b; r += 6) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            if (r + k < b) { /* uniform */
                const ur_u8 cc = *(ur_cc8)(vt + 8 * (r + k) - 8);
                const int y = 2 * (r + k) - 3;
                /* chroma ring: n advances so far in this trip (one in the middle of every odd k); newest pair in slot (n + 2) % 3, the
                 * pair two rows older in slot n % 3 */
                hpassL(buf[k % 3], ring[k % 3]);
                {
                    const int n = k >> 1; /* advances done before this step's first row */
                    emit(ring[(k + 1) % 3], ring[k % 3], cring[n % 3], cring[(n + 2) % 3], cc.s0, cc.s1, cc.s2, cc.s3, y >= 0);
                }
                dr += dstride;
                asm("" : "+s"(dr));
                if (k & 1) { /* the chroma window moves down one row between the two output rows of an even luma row */
                    const UrRawC cur = cnext;
                    load_chroma(cnext);
                    hpassC(cur, cring[(k >> 1) % 3]);
                }
                {
                    const int n = (k + 1) >> 1;
                    emit(ring[(k + 1) % 3], ring[k % 3], cring[n % 3], cring[(n + 2) % 3], cc.s4, cc.s5, cc.s6, cc.s7, y + 1 < dstH);
                }
                dr += dstride;
                asm("" : "+s"(dr));
            }
            load_luma(buf[k % 3]);
        }
    }
}

/* ================================================================================================== */
/* host side */

/*
 * Re-express a 4-tap bank of an exact `ratio`x up-scale (2 or 4) as coefficients on the REGULAR windows of the edge-replicated row:
 * output x reads samples clamp(s0 + k), k = 0..3, s0 = (x >> 1) - 2 + (x & 1) at 2x (ffhip_up2_virtual_bank), ((x + 2) >> 2) - 2 at 4x.
 * Every non-zero tap of the bank row must sit on one of those samples; taps the reference folded onto the edge sample land on one of
 * the replicas.  Output: n_dst x 2 dwords, (c0, c1) (c2, c3) as int16 pairs.  Returns 0 when the bank is not of this shape.
 */
int ffhip_upn_virtual_bank(const int16_t *filter, const int32_t *pos, int n_dst, int n_src, int ratio, std::vector<uint32_t> *out)
{
    if (ratio == 2)
        return ffhip_up2_virtual_bank(filter, pos, n_dst, n_src, out);
    if (ratio != 4 || n_dst != 4 * n_src || n_src < 4)
        return 0;
    out->assign((size_t)n_dst * 2, 0);
    for (int x = 0; x < n_dst; x++) {
        const int s0 = ((x + 2) >> 2) - 2;
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

/*
 * The two horizontal virtual banks (luma 2 srcW columns, chroma srcW columns; 2 dwords per column) in the 32 dwords the kernel reads
 * with scalar loads: away from the ends of a row an exact-2x bank repeats with period 2, so the even and the odd column's coefficients
 * stand for all of them; the three columns at either end (the only ones whose windows reach a replicated sample) keep their own.
 *   [0..3] luma even c01, c23, odd c01, c23   [4..7] chroma   [8..13] luma columns 0..2   [14..19] luma columns n-3..n-1
 *   [20..25] chroma columns 0..2   [26..31] chroma columns n-3..n-1
 * Returns 0 when a bank does not repeat (then the kernel is not used).
 */
int ffhip_up2rgb_hco(const std::vector<uint32_t> &hl, const std::vector<uint32_t> &hc, uint32_t out[32])
{
    const std::vector<uint32_t> *bank[2] = { &hl, &hc };
    for (int b = 0; b < 2; b++) {
        const std::vector<uint32_t> &v = *bank[b];
        const int n = (int)(v.size() / 2);
        if (n < 16 || (n & 1))
            return 0;
        for (int x = 3; x < n - 3; x++)
            if (v[2 * (size_t)x] != v[2 * (size_t)(4 + (x & 1))] || v[2 * (size_t)x + 1] != v[2 * (size_t)(4 + (x & 1)) + 1])
                return 0;
        for (int i = 0; i < 4; i++)
            out[4 * b + i] = v[8 + i]; /* columns 4 (even) and 5 (odd) */
        for (int i = 0; i < 6; i++) {
            out[8 + 12 * b + i] = v[i];
            out[14 + 12 * b + i] = v[2 * (size_t)(n - 3) + i];
        }
    }
    return 1;
}

/* strips of about `want` luma steps (a multiple of 6: the row loop is unrolled six times), evened out over the plane; frames per pack:
 * the 1, 2 or 4 (never more than the batch has) whose groups leave the fewest lanes of the pack's last wave idle */
void ffhip_up2rgb_plan(FFHipUp2RgbArgs *a, int want, int fpp)
{
    const int steps = a->srcH + 1;
    const int n = cdiv(steps, want);
    const int s = cdiv(cdiv(steps, n), 6) * 6;
    a->steps_per_strip = s;
    a->nstrips = cdiv(steps, s);
    int best = 1;
    long long best_idle = -1;
    for (int p = 1; p <= 4 && p <= (a->nframes > 0 ? a->nframes : 1); p *= 2) {
        const long long lanes = (long long)cdiv(p * a->ngroups, 64) * 64, idle = (lanes - (long long)p * a->ngroups) * 4 / p; /* per 4 frames */
        if (best_idle < 0 || idle < best_idle) {
            best_idle = idle;
            best = p;
        }
    }
    if (fpp == 1 || fpp == 2 || fpp == 4) /* measure build: forced */
        best = fpp;
    a->fpp = best;
    a->wpp = cdiv(best * a->ngroups, 64);
    a->npacks = cdiv(a->nframes > 0 ? a->nframes : 1, best);
}

int ffhip_launch_up2rgb(FFHipUp2RgbArgs &A, int var, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    const long long waves = (long long)A.wpp * A.nstrips * A.npacks;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    /* var (measure build): 0 the product (LDS transposer in 16-byte pieces, non-temporal stores); 1 plain (temporal) stores; 2 direct
     * stores (no transposer); 3 transposer in 8-byte pieces.  (Reading the tile back one row later, so that neither LDS side waits, was
     * measured too: no faster — profiles/r05_up2rgb_c.txt, _d.txt — and is not kept.) */
#define UR_LAUNCH(L) do { if (A.sil) hipLaunchKernelGGL((k_sws_up2_rgb<L, 2, true, true>), grid, block, 0, stream, A); \
                          else if (var == 2) hipLaunchKernelGGL((k_sws_up2_rgb<L, 0, true>), grid, block, 0, stream, A); \
                          else if (var == 1) hipLaunchKernelGGL((k_sws_up2_rgb<L, 2, false>), grid, block, 0, stream, A); \
                          else if (var == 3) hipLaunchKernelGGL((k_sws_up2_rgb<L, 1, true>), grid, block, 0, stream, A); \
                          else hipLaunchKernelGGL((k_sws_up2_rgb<L, 2, true>), grid, block, 0, stream, A); } while (0)
    switch (A.lay) {
    case 0: UR_LAUNCH(0); break;
    case 1: UR_LAUNCH(1); break;
    case 2: UR_LAUNCH(2); break;
    case 3: UR_LAUNCH(3); break;
    case 4: UR_LAUNCH(4); break;
    case 5: UR_LAUNCH(5); break;
    default:
        ffhip_set_error("ffhip_sws: packed layout %d is not one of the exact-2x RGB writer's", A.lay);
        return FFHIP_EINVAL;
    }
#undef UR_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
