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

template <int PAIR>
__device__ __forceinline__ void u32_unit(const FFHipU32Args &A, const FFHipU32Job &J, int frame, int gbase, int strip, int lane)
{
    constexpr int NW = PAIR ? 6 : 4; /* dwords of a source row under this lane's windows */
    const int graw = gbase + lane;
    const bool act = graw < J.ngroups;
    const int g = min(graw, J.ngroups - 1);
    const bool lb = g == 0, rb = g == J.ngroups - 1;
    const bool border = gbase == 0 || gbase + 64 >= J.ngroups; /* wave-uniform */
    /* plane: samples 4g - 2 .. 4g + 5; pair: columns 2g - 2 .. 2g + 3.  The first / last lane of a row loads one dword (pair: two)
     * further inside and rebuilds the replicated ones */
    const uint32_t soff = PAIR ? (uint32_t)(lb ? 0 : 8 * g - 8 - (rb ? 8 : 0)) : (uint32_t)(lb ? 0 : 8 * g - 4 - (rb ? 4 : 0));
    const int hsh = A.sdepth - 1, smsb = A.smsb ? 16 - A.sdepth : 0;
    const uint32_t dmsb = (uint32_t)(A.dmsb ? 16 - A.ddepth : 0) * 0x00010001u;
    const int vsh = 27 - A.ddepth, vseed = 1 << (26 - A.ddepth);
    const uint32_t maxpk = (uint32_t)((1 << A.ddepth) - 1) * 0x00010001u;
    uint32_t cf[6][2];
    {
        /* plane: outputs 6g .. 6g + 5; pair: columns 3g .. 3g + 2, both channels of a column share its coefficients */
        const uint32_t *p = J.hfv + (size_t)g * (PAIR ? 6 : 12);
#pragma unroll
        for (int j = 0; j < 6; j++)
#pragma unroll
            for (int k = 0; k < 2; k++)
                cf[j][k] = p[2 * (PAIR ? j >> 1 : j) + k];
    }
    const int a = strip * J.strip_rows, b = min(a + J.strip_rows, J.dstH); /* this strip's output rows; a is a multiple of 9 */
    const uint8_t *sbase = J.src + (size_t)frame * J.sfp;
    uint8_t *dbase = J.dst + (size_t)frame * J.dfp;
    const ptrdiff_t sstride = J.sstride, dstride = J.dstride;
    const int srcH = J.srcH;

    auto load_row = [&](int r, uint32_t (&w)[NW]) {
        const uint8_t *p = sbase + (ptrdiff_t)min(max(r, 0), srcH - 1) * sstride; /* rows above / below the plane replicate the edge row */
        const u3_u4 v = *(u3_gc4)((u3_gcp)p + soff);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        if (PAIR) {
            const u3_u2 e = *(u3_gc2)((u3_gcp)p + soff + 16);
            w[NW - 2] = e.x; w[NW - 1] = e.y;
        }
    };
    /* the horizontal pass of one source row: this lane's 6 samples (plane: x0 .. x5; pair: u0 v0 u1 v1 u2 v2) */
    auto hpass = [&](const uint32_t (&raw)[NW], int (&h)[6]) {
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
                const uint32_t f0 = __builtin_amdgcn_perm(raw[0], raw[0], 0x01000100u), fl = __builtin_amdgcn_perm(raw[3], raw[3], 0x03020302u);
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
        uint32_t pa[6], pb[6];
        if (PAIR) {
            /* column j of the lane's period reads columns j .. j + 3 from the lane's base: the channels' pairs (c, c + 1), (c + 2, c + 3) */
#pragma unroll
            for (int j = 0; j < 3; j++) {
                pa[2 * j] = __builtin_amdgcn_perm(w[j + 1], w[j], 0x05040100u);
                pa[2 * j + 1] = __builtin_amdgcn_perm(w[j + 1], w[j], 0x07060302u);
                pb[2 * j] = __builtin_amdgcn_perm(w[j + 3], w[j + 2], 0x05040100u);
                pb[2 * j + 1] = __builtin_amdgcn_perm(w[j + 3], w[j + 2], 0x07060302u);
            }
        } else {
            /* sample x = 3 k + j of the lane: j = 0: w[k], w[k + 1]; j = 1: o[k], o[k + 1] (the pairs that start on an odd sample);
             * j = 2: w[k + 1], w[k + 2] */
            uint32_t o[3];
#pragma unroll
            for (int k = 0; k < 3; k++)
                o[k] = __builtin_amdgcn_alignbit(w[k + 1], w[k], 16);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                pa[3 * k] = w[k];         pb[3 * k] = w[k + 1];
                pa[3 * k + 1] = o[k];     pb[3 * k + 1] = o[k + 1];
                pa[3 * k + 2] = w[k + 1]; pb[3 * k + 2] = w[k + 2];
            }
        }
        u3_h6(h, pa, pb, cf, hsh);
    };

    /* source rows in trips of six from rbase = 2 (a / 3) - 6 (a multiple of 6); rows rbase + 4, + 5 only fill the ring.  Straight-line code:
     * every due row is computed, the store alone looks at the strip's bounds (branches around the arithmetic made the compiler copy the
     * rows in flight at every join — and wait for them); the walk ends at uniform exits after the row the strip's last output ends on */
    const int rbase0 = 2 * (a / 3) - 6, r_last = 2 * (b / 3) + 1, dstH = J.dstH;
    uint32_t ring[3][6];
    int hprev[6];
#pragma unroll
    for (int c = 0; c < 6; c++)
        hprev[c] = 0;
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int c = 0; c < 6; c++)
            ring[s][c] = 0;
    constexpr int D = 3; /* source rows in flight */
    uint32_t nxt[D][NW];
#pragma unroll
    for (int i = 0; i < D; i++)
        load_row(rbase0 + 4 + i, nxt[(4 + i) % D]);
    const u3_cc vt = (u3_cc)J.vfv;
    const uint32_t doff = 12u * (uint32_t)g;

    auto emit = [&](int y, const uint32_t (&p0)[6], const uint32_t (&p1)[6]) {
        const int yc = min(max(y, 0), dstH - 1);
        const uint32_t c0 = vt[2 * yc], c1 = vt[2 * yc + 1];
        uint32_t o[3];
        u3_v6(o, p0, p1, c0, c1, vseed, vsh, maxpk, dmsb);
        if (act && y >= a && y < b)
            *(u3_g3)((u3_gp)(dbase + (ptrdiff_t)y * dstride) + doff) = (u3_u3){ o[0], o[1], o[2] };
    };
    /* row r = rbase + u: P(r - 1) = (row r - 1, row r) — v_cvt_pk_i16_i32 saturates = min(., 32767) (no sum of an admitted bank falls below
     * -32768) — then r = 2m + 1: rows 3m - 1 and 3m;  r = 2m + 2: row 3m + 1 — all on P(r - 3), P(r - 1) */
#define U3_STEP(u, EMIT)                                                                                                                  \
    {                                                                                                                                     \
        const int r = rbase + (u);                                                                                                        \
        uint32_t cur[NW];                                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NW; i++) cur[i] = nxt[(u) % D][i];                                                         \
        load_row(r + D, nxt[(u) % D]);                                                                                                    \
        int h[6];                                                                                                                         \
        hpass(cur, h);                                                                                                                    \
        _Pragma("unroll") for (int c = 0; c < 6; c++) {                                                                                  \
            ring[((u) + 2) % 3][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[c], h[c]));                          \
            hprev[c] = h[c];                                                                                                              \
        }                                                                                                                                 \
        if (EMIT) {                                                                                                                       \
            if ((u) & 1) {                                                                                                                \
                const int y1 = 3 * ((r - 1) >> 1) - 1;                                                                                    \
                emit(y1, ring[(u) % 3], ring[((u) + 2) % 3]);                                                                             \
                emit(y1 + 1, ring[(u) % 3], ring[((u) + 2) % 3]);                                                                         \
            } else {                                                                                                                      \
                emit(3 * ((r - 2) >> 1) + 1, ring[(u) % 3], ring[((u) + 2) % 3]);                                                         \
            }                                                                                                                             \
        }                                                                                                                                 \
    }
    {
        const int rbase = rbase0;
        U3_STEP(4, false)
        U3_STEP(5, false)
    }
    for (int rbase = rbase0 + 6; ; rbase += 6) {
        U3_STEP(0, true)
        U3_STEP(1, true)
        if (rbase + 2 > r_last)
            return;
        U3_STEP(2, true)
        U3_STEP(3, true)
        if (rbase + 4 > r_last)
            return;
        U3_STEP(4, true)
        U3_STEP(5, true)
        if (rbase + 6 > r_last)
            return;
    }
#undef U3_STEP
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
    if (J.pair)
        u32_unit<1>(A, J, frame, cb * 64, strip, lane);
    else
        u32_unit<0>(A, J, frame, cb * 64, strip, lane);
}

/* ================================================================================================== */
/* host side */

/*
 * Re-express a bank of an exact 3:2 up-scale (at most 4 taps) as coefficients on the REGULAR windows of the edge-replicated row:
 * output x reads samples clamp(2 (x / 3) - 2 + x % 3 + k), k = 0..3.  Every non-zero tap of the bank row must sit on one of those
 * samples; taps the reference folded onto the edge sample land on one of the replicas.  Output: n_dst x 2 dwords, (c0, c1) (c2, c3) as
 * int16 pairs.  Returns 0 when the bank is not of this shape.
 */
int ffhip_u32_virtual_bank(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, std::vector<uint32_t> *out)
{
    if (2 * n_dst != 3 * n_src || fsize < 1 || fsize > 16 || (n_dst % 3))
        return 0;
    out->assign((size_t)n_dst * 2, 0);
    for (int x = 0; x < n_dst; x++) {
        const int s0 = 2 * (x / 3) - 2 + x % 3;
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
            if (j.ngroups < 3 || j.dstH <= 0 || (j.dstH % 3)) {
                ffhip_set_error("ffhip_sws: the exact-3:2 up-scaler takes rows of three groups or more and a multiple of three output rows");
                return FFHIP_EINVAL;
            }
            const int n = cdiv(j.dstH, want);
            j.strip_rows = cdiv(cdiv(j.dstH, n), 9) * 9;
            j.nstrips = cdiv(j.dstH, j.strip_rows);
            j.ncb = cdiv(j.ngroups, 64);
            u += (long long)j.ncb * j.nstrips;
        }
        if (u * A.nframes >= 8192 || want <= 9)
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
    hipLaunchKernelGGL(k_sws_up32, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, A);
    LAUNCH_CHECK();
    return 0;
}
