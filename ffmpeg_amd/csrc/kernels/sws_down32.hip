/*
 * sws_down32.hip — the fused H+V scaler for EXACT 3:2 down-scaling (1080p -> 720p, 4K -> 1440p, 1440p -> 960p ...) with banks of up to
 * 6 taps in both directions (bicubic, bilinear, point), planes and byte-interleaved U/V pairs (NV12 / NV21) in and out (round 5).
 *
 * Arithmetic: hScale8To15_c (libswscale/swscale.c:128-142), nv12ToUV_c (input.c:936), yuv2planeX_8_c / yuv2nv12cX_c (output.c:468-529):
 * int32 sums, >> 7 and min(., 32767) for the horizontal pass, the 64 << 12 seed, >> 19 and the clip to 8 bits for the vertical one.
 * Same results as sws_lwalk.hip / sws_scale.hip, bit for bit (tests/test_gpu_sws_fast.py).
 *
 * The commonest down-scale that is not 2:1 ran on the wide walker (LDS row buffers and a vertical ring in LDS, dynamically indexed: 0.17 –
 * 0.24 of HBM).  At exactly 3:2 everything is regular with period (3 in, 2 out):
 *  - output x = 2k + j reads source samples 3k - 2 + j .. 3k + 3 + j (initFilter, libswscale/utils.c:519-561, folds the taps that fall
 *    outside the row onto the edge sample: the regular bank over an edge-REPLICATED row with its own coefficients next to either edge —
 *    ffhip_d32_virtual_bank re-expresses every bank row that way, tap by tap, else this kernel is not used).  A lane owns 8 outputs = 4
 *    periods = 12 source samples: 20 source bytes at the dword-aligned offset 12g - 4, every window a FIXED byte position in them (static
 *    v_perm_b32 selectors), three v_dot2_i32_i16 per output.  A U/V pair is the same 20 bytes: 4 columns x 2 channels per lane.
 *  - the vertical schedule is static with period (3 source rows, 2 output rows): the pairs P(q) = (row q, row q + 1) of horizontally
 *    filtered samples for EVERY q (an even output row starts on 3m - 2, an odd one on 3m - 1) sit in a ring of six in registers; after
 *    source row r = 0 or 1 (mod 3) an output row is due and reads P(r - 5), P(r - 3), P(r - 1).  Six source rows per loop trip make
 *    every ring index a constant.  The rows' coefficient pairs are wave-uniform (scalar loads).
 * Round 6: the dots in hand-scheduled blocks (VOP3P form, no v_mov per chain) and every row straight-line, the strip's bounds on the store
 * alone (see the loop); 8 bytes out per lane and row.
 */
#include <type_traits>

#include "common.h"
#include "sws_kernels.h"

typedef short d3_s2 __attribute__((ext_vector_type(2)));
typedef uint32_t d3_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t d3_u4 __attribute__((ext_vector_type(4)));
typedef d3_u4 __attribute__((aligned(4))) d3_u4a;
typedef d3_u2 __attribute__((aligned(4))) d3_u2a;
typedef const uint8_t __attribute__((address_space(1))) *d3_gcp;
typedef uint8_t __attribute__((address_space(1))) *d3_gp;
typedef const d3_u4a __attribute__((address_space(1))) *d3_gc4;
typedef const uint32_t __attribute__((address_space(1))) *d3_gc1;
typedef d3_u2a __attribute__((address_space(1))) *d3_g2;
typedef uint32_t __attribute__((address_space(1))) *d3_g1;

__device__ __forceinline__ int d3_dot(uint32_t p, uint32_t c, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(d3_s2, p), __builtin_bit_cast(d3_s2, c), acc, false);
}
/* clip_u8(a) | clip_u8(b) << 8 | clip_u8(c) << 16 | clip_u8(d) << 24 of values already shifted */
__device__ __forceinline__ uint32_t d3_pk4(int a, int b, int c, int d)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 0\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 0 op_sel:[0,0,0,1]"
        : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}

/* the int16 pair (byte q, byte q + STEP) of the 20 bytes in w[0..4]: q and STEP are compile-time constants */
template <int Q, int STEP>
__device__ __forceinline__ uint32_t d3_pair(const uint32_t (&w)[5])
{
    constexpr int i = Q >> 2, r = Q & 3;
    constexpr uint32_t sel = 0x0c000c00u | (uint32_t)(r + STEP) << 16 | (uint32_t)r;
    return __builtin_amdgcn_perm(w[i + 1 > 4 ? 4 : i + 1], w[i], sel);
}


/*
 * Hand-scheduled dot products (round 6, as in sws_up32.hip): the builtin becomes the accumulate-in-place v_dot2c_i32_i16 plus a v_mov per chain;
 * the VOP3P form takes the seed as a third source.  gfx950: a DOT result may feed the same opcode as src2 at once, any other VALU only
 * after 3 wait states — four (eight) chains are interleaved and every result is first read three instructions after its last DOT.
 */
/* four horizontal samples: d[i] = (p[i][0] . c[i][0] + p[i][1] . c[i][1] + p[i][2] . c[i][2]) >> 7 */
__device__ __forceinline__ void d3_h4(int &d0, int &d1, int &d2, int &d3, const uint32_t (&p)[4][3], const uint32_t (&c)[4][3])
{
    asm("v_dot2_i32_i16 %0, %4, %16, 0\n\t"
        "v_dot2_i32_i16 %1, %5, %17, 0\n\t"
        "v_dot2_i32_i16 %2, %6, %18, 0\n\t"
        "v_dot2_i32_i16 %3, %7, %19, 0\n\t"
        "v_dot2_i32_i16 %0, %8, %20, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %21, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %22, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %23, %3\n\t"
        "v_dot2_i32_i16 %0, %12, %24, %0\n\t"
        "v_dot2_i32_i16 %1, %13, %25, %1\n\t"
        "v_dot2_i32_i16 %2, %14, %26, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %27, %3\n\t"
        "v_ashrrev_i32 %0, 7, %0\n\t"
        "v_ashrrev_i32 %1, 7, %1\n\t"
        "v_ashrrev_i32 %2, 7, %2\n\t"
        "v_ashrrev_i32 %3, 7, %3"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
        : "v"(p[0][0]), "v"(p[1][0]), "v"(p[2][0]), "v"(p[3][0]), "v"(p[0][1]), "v"(p[1][1]), "v"(p[2][1]), "v"(p[3][1]), "v"(p[0][2]), "v"(p[1][2]),
          "v"(p[2][2]), "v"(p[3][2]), "v"(c[0][0]), "v"(c[1][0]), "v"(c[2][0]), "v"(c[3][0]), "v"(c[0][1]), "v"(c[1][1]), "v"(c[2][1]), "v"(c[3][1]),
          "v"(c[0][2]), "v"(c[1][2]), "v"(c[2][2]), "v"(c[3][2]));
}
/* one output row of 8 samples: t[i] = seed + pa[i] . c0 + pb[i] . c1 + pc[i] . c2, bytes clip_u8(t[i] >> 19) packed in sample order */
__device__ __forceinline__ void d3_v8(uint32_t &w0, uint32_t &w1, const uint32_t (&pa)[8], const uint32_t (&pb)[8], const uint32_t (&pc)[8], uint32_t c0,
                                      uint32_t c1, uint32_t c2, int seed)
{
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_dot2_i32_i16 %2, %10, %34, %37\n\t"
        "v_dot2_i32_i16 %3, %11, %34, %37\n\t"
        "v_dot2_i32_i16 %4, %12, %34, %37\n\t"
        "v_dot2_i32_i16 %5, %13, %34, %37\n\t"
        "v_dot2_i32_i16 %6, %14, %34, %37\n\t"
        "v_dot2_i32_i16 %7, %15, %34, %37\n\t"
        "v_dot2_i32_i16 %8, %16, %34, %37\n\t"
        "v_dot2_i32_i16 %9, %17, %34, %37\n\t"
        "v_dot2_i32_i16 %2, %18, %35, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %35, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %35, %4\n\t"
        "v_dot2_i32_i16 %5, %21, %35, %5\n\t"
        "v_dot2_i32_i16 %6, %22, %35, %6\n\t"
        "v_dot2_i32_i16 %7, %23, %35, %7\n\t"
        "v_dot2_i32_i16 %8, %24, %35, %8\n\t"
        "v_dot2_i32_i16 %9, %25, %35, %9\n\t"
        "v_dot2_i32_i16 %2, %26, %36, %2\n\t"
        "v_dot2_i32_i16 %3, %27, %36, %3\n\t"
        "v_dot2_i32_i16 %4, %28, %36, %4\n\t"
        "v_dot2_i32_i16 %5, %29, %36, %5\n\t"
        "v_dot2_i32_i16 %6, %30, %36, %6\n\t"
        "v_dot2_i32_i16 %7, %31, %36, %7\n\t"
        "v_dot2_i32_i16 %8, %32, %36, %8\n\t"
        "v_dot2_i32_i16 %9, %33, %36, %9\n\t"
        "v_ashr_pk_u8_i32 %0, %2, %3, 19\n\t"
        "v_ashr_pk_u8_i32 %1, %6, %7, 19\n\t"
        "v_ashr_pk_u8_i32 %0, %4, %5, 19 op_sel:[0,0,0,1]\n\t"
        "v_ashr_pk_u8_i32 %1, %8, %9, 19 op_sel:[0,0,0,1]"
        : "=&v"(w0), "=&v"(w1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pa[6]), "v"(pa[7]),
          "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pb[4]), "v"(pb[5]), "v"(pb[6]), "v"(pb[7]),
          "v"(pc[0]), "v"(pc[1]), "v"(pc[2]), "v"(pc[3]), "v"(pc[4]), "v"(pc[5]), "v"(pc[6]), "v"(pc[7]),
          "s"(c0), "s"(c1), "s"(c2), "v"(seed));
}

template <int PAIR>
__device__ __forceinline__ void d32_unit(const FFHipD32Job &J, int frame, int gbase, int strip, int lane)
{
    const int graw = gbase + lane;
    const bool act = graw < J.ngroups;
    const int g = min(graw, J.ngroups - 1);
    const bool lb = g == 0, rb = g == J.ngroups - 1;
    const bool border = gbase == 0 || gbase + 64 >= J.ngroups; /* wave-uniform */
    /* the first / last lane of a row loads its 20 bytes one dword further inside and rebuilds the replicated ones */
    const uint32_t soff = (uint32_t)(lb ? 0 : 12 * g - 4 - (rb ? 4 : 0));
    uint32_t cf[8][3];
    {
        /* plane: outputs 8g .. 8g + 7; pair: columns 4g .. 4g + 3, both channels of a column share its coefficients */
        const uint32_t *p = J.hfv + (size_t)g * (PAIR ? 12 : 24);
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                cf[j][k] = p[3 * (PAIR ? j >> 1 : j) + k];
    }
    const int a = strip * J.strip_rows, b = min(a + J.strip_rows, J.dstH); /* this strip's output rows; a is a multiple of 4 */
    const uint8_t *sbase = J.src + (size_t)frame * J.sfp;
    uint8_t *dbase = J.dst + (size_t)frame * J.dfp;
    const ptrdiff_t sstride = J.sstride, dstride = J.dstride;
    const int srcH = J.srcH;
    const bool swap = PAIR && J.swap;

    auto load_row = [&](int r, uint32_t (&w)[5]) {
        const uint8_t *p = sbase + (ptrdiff_t)min(max(r, 0), srcH - 1) * sstride; /* rows above / below the plane replicate the edge row */
        const d3_u4 v = *(d3_gc4)((d3_gcp)p + soff);
        const uint32_t e = *(d3_gc1)((d3_gcp)p + soff + 16);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; w[4] = e;
    };
    /* the horizontal pass of one source row: this lane's 8 samples, >> 7 (sample order: plane x0..x7; pair u0 v0 u1 v1 u2 v2 u3 v3) */
    auto hpass = [&](const uint32_t (&raw)[5], int (&h)[8]) {
        uint32_t w[5] = { raw[0], raw[1], raw[2], raw[3], raw[4] };
        if (border) {
            const uint32_t f0 = PAIR ? __builtin_amdgcn_perm(raw[0], raw[0], 0x01000100u) : __builtin_amdgcn_perm(raw[0], raw[0], 0x00000000u);
            const uint32_t f4 = PAIR ? __builtin_amdgcn_perm(raw[4], raw[4], 0x03020302u) : __builtin_amdgcn_perm(raw[4], raw[4], 0x03030303u);
            w[0] = lb ? f0 : rb ? raw[1] : raw[0];
            w[1] = lb ? raw[0] : rb ? raw[2] : raw[1];
            w[2] = lb ? raw[1] : rb ? raw[3] : raw[2];
            w[3] = lb ? raw[2] : rb ? raw[4] : raw[3];
            w[4] = lb ? raw[3] : rb ? f4 : raw[4];
        }
        if (swap) { /* uniform: the channel wanted at the even destination bytes sits at the odd source bytes */
#pragma unroll
            for (int i = 0; i < 5; i++)
                w[i] = __builtin_amdgcn_perm(w[i], w[i], 0x02030001u);
        }
        /* window start of output j, in bytes from the loaded base (= sample 12g - 4, column 6g - 2):
         *   plane: j -> 2 + 3 (j >> 1) + (j & 1);   pair: sample e = 2 col + ch -> 2 (3 (col >> 1) + (col & 1)) + ch
         * (the windows overlap: the compiler folds the 24 pairs into the 15 / 18 distinct ones) */
#define D3_S(j) (PAIR ? 2 * (3 * (((j) >> 1) >> 1) + (((j) >> 1) & 1)) + ((j) & 1) : 2 + 3 * ((j) >> 1) + ((j) & 1))
#define D3_P(j, k) d3_pair<D3_S(j) + 2 * (k) * (PAIR ? 2 : 1), (PAIR ? 2 : 1)>(w)
#define D3_Q(j0)                                                                                                                         \
        {                                                                                                                                \
            const uint32_t p[4][3] = { { D3_P(j0, 0), D3_P(j0, 1), D3_P(j0, 2) }, { D3_P(j0 + 1, 0), D3_P(j0 + 1, 1), D3_P(j0 + 1, 2) },  \
                                       { D3_P(j0 + 2, 0), D3_P(j0 + 2, 1), D3_P(j0 + 2, 2) }, { D3_P(j0 + 3, 0), D3_P(j0 + 3, 1), D3_P(j0 + 3, 2) } }; \
            const uint32_t c[4][3] = { { cf[j0][0], cf[j0][1], cf[j0][2] }, { cf[j0 + 1][0], cf[j0 + 1][1], cf[j0 + 1][2] },             \
                                       { cf[j0 + 2][0], cf[j0 + 2][1], cf[j0 + 2][2] }, { cf[j0 + 3][0], cf[j0 + 3][1], cf[j0 + 3][2] } }; \
            d3_h4(h[j0], h[j0 + 1], h[j0 + 2], h[j0 + 3], p, c);                                                                         \
        }
        D3_Q(0) D3_Q(4)
#undef D3_Q
#undef D3_P
#undef D3_S
    };

    /* source rows in trips of six from rbase = 3 (a / 2) - 6 (a multiple of 6); rows rbase + 4, + 5 only fill the ring.  Straight-line code
     * (round 6): every due row is computed, the store alone looks at the strip's bounds — branches around the arithmetic made the compiler
     * copy the rows in flight at every join, and wait for them — and the walk ends at uniform exits after the row the strip's last output
     * ends on */
    const int rbase0 = 3 * (a >> 1) - 6, r_last = 3 * (b >> 1) + 1, dstH = J.dstH;
    uint32_t ring[6][8];
    int hprev[8];
#pragma unroll
    for (int c = 0; c < 8; c++)
        hprev[c] = 0;
#pragma unroll
    for (int s = 0; s < 6; s++)
#pragma unroll
        for (int c = 0; c < 8; c++)
            ring[s][c] = 0;
    uint32_t nxt[3][5]; /* three source rows in flight */
    load_row(rbase0 + 4, nxt[1]);
    load_row(rbase0 + 5, nxt[2]);
    load_row(rbase0 + 6, nxt[0]);
    typedef const uint32_t __attribute__((address_space(4))) *d3_cc; /* constant address space: scalar loads */
    const d3_cc vt = (d3_cc)J.vfv;
    const uint32_t doff = 8u * (uint32_t)g;

    /* row r = rbase + u: P(r - 1) = (row r - 1, row r), int16-saturated = min(., 32767) + truncation (no sum of an admitted bank falls below
     * -32768); then r = 3 (m + 1): output row 2m, r = 3m + 4: output row 2m + 1, on P(r - 5), P(r - 3), P(r - 1) */
#define D3_STEP(u, EMIT)                                                                                                                 \
    {                                                                                                                                    \
        const int r = rbase + (u);                                                                                                       \
        uint32_t cur[5];                                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 5; i++) cur[i] = nxt[(u) % 3][i];                                                         \
        load_row(r + 3, nxt[(u) % 3]);                                                                                                   \
        int h[8];                                                                                                                        \
        hpass(cur, h);                                                                                                                   \
        _Pragma("unroll") for (int c = 0; c < 8; c++) {                                                                                 \
            ring[((u) + 5) % 6][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[c], h[c]));                         \
            hprev[c] = h[c];                                                                                                             \
        }                                                                                                                                \
        if (EMIT && (u) % 3 != 2) {                                                                                                      \
            const int y = (u) % 3 == 0 ? 2 * (r / 3) - 2 : 2 * ((r - 1) / 3) - 1;                                                        \
            const int yc = min(max(y, 0), dstH - 1);                                                                                     \
            uint32_t o0, o1;                                                                                                             \
            d3_v8(o0, o1, ring[((u) + 1) % 6], ring[((u) + 3) % 6], ring[((u) + 5) % 6], vt[4 * yc], vt[4 * yc + 1], vt[4 * yc + 2], 64 << 12); \
            if (act && y >= a && y < b)                                                                                                  \
                *(d3_g2)((d3_gp)(dbase + (ptrdiff_t)y * dstride) + doff) = (d3_u2){ o0, o1 };                                            \
        }                                                                                                                                \
    }
    {
        const int rbase = rbase0;
        D3_STEP(4, false)
        D3_STEP(5, false)
    }
    for (int rbase = rbase0 + 6; ; rbase += 6) {
        D3_STEP(0, true)
        D3_STEP(1, true)
        if (rbase + 2 > r_last)
            return;
        D3_STEP(2, true)
        D3_STEP(3, true)
        D3_STEP(4, true)
        if (rbase + 5 > r_last)
            return;
        D3_STEP(5, true)
    }
#undef D3_STEP
}

__global__ __launch_bounds__(256) void k_sws_down32(FFHipD32Args A)
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
    const FFHipD32Job &J = A.job[j];
    const int local = u - J.unit_begin;
    const int strip = local / J.ncb, cb = local - strip * J.ncb;
    if (J.pair)
        d32_unit<1>(J, frame, cb * 64, strip, lane);
    else
        d32_unit<0>(J, frame, cb * 64, strip, lane);
}


/* ================================================================================================== */
/*
 * The 16-bit twin (round 6): exact 3:2 down-scaling of 9..14-bit samples — planes of words and interleaved (u, v) planes of words (P01x) in
 * and out, hScale16To15_c / yuv2planeX_10_c / yuv2p01xlX as in sws_walk16.hip / sws_up32.hip.  A lane owns 4 outputs = 2 periods: a plane's
 * 10 source samples are the 5 dwords at 12g - 4 (the pairs of the windows that start on an even sample are its dwords, the others one
 * v_alignbit away), a pair's 7 (u, v) columns the 7 dwords at 12g - 8 (the channels' pairs split with v_perm_b32); 8 bytes out per lane
 * and row.  Same schedule as above: P(q) for every q in a ring of six, outputs after rows 0 and 1 (mod 3).
 */
typedef unsigned short d3_h2 __attribute__((ext_vector_type(2)));
typedef const d3_u2a __attribute__((address_space(1))) *d3_gc2;
/* four horizontal samples, 16-bit sources: d[i] = (p[i][0] . c[i][0] + p[i][1] . c[i][1] + p[i][2] . c[i][2]) >> sh */
__device__ __forceinline__ void d3_h4s(int &d0, int &d1, int &d2, int &d3, const uint32_t (&p)[4][3], const uint32_t (&c)[4][3], int sh)
{
    asm("v_dot2_i32_i16 %0, %4, %16, 0\n\t"
        "v_dot2_i32_i16 %1, %5, %17, 0\n\t"
        "v_dot2_i32_i16 %2, %6, %18, 0\n\t"
        "v_dot2_i32_i16 %3, %7, %19, 0\n\t"
        "v_dot2_i32_i16 %0, %8, %20, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %21, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %22, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %23, %3\n\t"
        "v_dot2_i32_i16 %0, %12, %24, %0\n\t"
        "v_dot2_i32_i16 %1, %13, %25, %1\n\t"
        "v_dot2_i32_i16 %2, %14, %26, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %27, %3\n\t"
        "v_ashrrev_i32 %0, %28, %0\n\t"
        "v_ashrrev_i32 %1, %28, %1\n\t"
        "v_ashrrev_i32 %2, %28, %2\n\t"
        "v_ashrrev_i32 %3, %28, %3"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
        : "v"(p[0][0]), "v"(p[1][0]), "v"(p[2][0]), "v"(p[3][0]), "v"(p[0][1]), "v"(p[1][1]), "v"(p[2][1]), "v"(p[3][1]), "v"(p[0][2]), "v"(p[1][2]),
          "v"(p[2][2]), "v"(p[3][2]), "v"(c[0][0]), "v"(c[1][0]), "v"(c[2][0]), "v"(c[3][0]), "v"(c[0][1]), "v"(c[1][1]), "v"(c[2][1]), "v"(c[3][1]),
          "v"(c[0][2]), "v"(c[1][2]), "v"(c[2][2]), "v"(c[3][2]), "s"(sh));
}
/* four output samples as two dwords: t[i] = seed + pa[i] . c0 + pb[i] . c1 + pc[i] . c2, >> sh, clipped to 0 .. 2^depth - 1, << msb */
__device__ __forceinline__ void d3_v4h(uint32_t &w0, uint32_t &w1, const uint32_t (&pa)[4], const uint32_t (&pb)[4], const uint32_t (&pc)[4], uint32_t c0,
                                       uint32_t c1, uint32_t c2, int seed, int sh, uint32_t maxpk, uint32_t msb)
{
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %2, %6, %18, %21\n\t"
        "v_dot2_i32_i16 %3, %7, %18, %21\n\t"
        "v_dot2_i32_i16 %4, %8, %18, %21\n\t"
        "v_dot2_i32_i16 %5, %9, %18, %21\n\t"
        "v_dot2_i32_i16 %2, %10, %19, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %19, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %19, %4\n\t"
        "v_dot2_i32_i16 %5, %13, %19, %5\n\t"
        "v_dot2_i32_i16 %2, %14, %20, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %20, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %20, %4\n\t"
        "v_dot2_i32_i16 %5, %17, %20, %5\n\t"
        "v_ashrrev_i32 %2, %22, %2\n\t"
        "v_ashrrev_i32 %3, %22, %3\n\t"
        "v_ashrrev_i32 %4, %22, %4\n\t"
        "v_ashrrev_i32 %5, %22, %5\n\t"
        "v_cvt_pk_i16_i32 %0, %2, %3\n\t"
        "v_cvt_pk_i16_i32 %1, %4, %5\n\t"
        "v_pk_max_i16 %0, %0, 0\n\t"
        "v_pk_max_i16 %1, %1, 0\n\t"
        "v_pk_min_i16 %0, %0, %23\n\t"
        "v_pk_min_i16 %1, %1, %23\n\t"
        "v_pk_lshlrev_b16 %0, %24, %0\n\t"
        "v_pk_lshlrev_b16 %1, %24, %1"
        : "=&v"(w0), "=&v"(w1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pc[0]), "v"(pc[1]), "v"(pc[2]), "v"(pc[3]),
          "s"(c0), "s"(c1), "s"(c2), "v"(seed), "s"(sh), "s"(maxpk), "s"(msb));
}

/* ff_dither_8x8_128 (libswscale/swscale.c:42-52): the ordered dither of an 8-bit target fed from a deeper source (swscale.c:291,519-522) */
__constant__ __attribute__((aligned(8))) uint8_t d3_dither[8][8] = {
    {  36, 68,  60, 92,  34, 66,  58, 90, }, { 100,  4, 124, 28,  98,  2, 122, 26, }, {  52, 84,  44, 76,  50, 82,  42, 74, },
    { 116, 20, 108, 12, 114, 18, 106, 10, }, {  32, 64,  56, 88,  38, 70,  62, 94, }, {  96,  0, 120, 24, 102,  6, 126, 30, },
    {  48, 80,  40, 72,  54, 86,  46, 78, }, { 112, 16, 104,  8, 118, 22, 110, 14, },
};
/* four output BYTES from 15-bit lines (a 10-bit decoder's frames for an 8-bit consumer): t[i] = (dither[i] << 12) + pa[i] . c0 + pb[i] . c1 +
 * pc[i] . c2, clip_u8(t[i] >> 19) (yuv2planeX_8_c / yuv2nv12cX_c, libswscale/output.c:468-529) */
__device__ __forceinline__ uint32_t d3_v4d(const uint32_t (&pa)[4], const uint32_t (&pb)[4], const uint32_t (&pc)[4], uint32_t c0, uint32_t c1, uint32_t c2,
                                           const int (&sd)[4])
{
    uint32_t out;
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %1, %5, %17, %20\n\t"
        "v_dot2_i32_i16 %2, %6, %17, %21\n\t"
        "v_dot2_i32_i16 %3, %7, %17, %22\n\t"
        "v_dot2_i32_i16 %4, %8, %17, %23\n\t"
        "v_dot2_i32_i16 %1, %9, %18, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %18, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %18, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %18, %4\n\t"
        "v_dot2_i32_i16 %1, %13, %19, %1\n\t"
        "v_dot2_i32_i16 %2, %14, %19, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %19, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %19, %4\n\t"
        "s_nop 0\n\t"
        "v_ashr_pk_u8_i32 %0, %1, %2, 19\n\t"
        "s_nop 1\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 19 op_sel:[0,0,0,1]"
        : "=&v"(out), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pc[0]), "v"(pc[1]), "v"(pc[2]), "v"(pc[3]),
          "s"(c0), "s"(c1), "s"(c2), "v"(sd[0]), "v"(sd[1]), "v"(sd[2]), "v"(sd[3]));
    return out;
}

/* six-chain forms of the two blocks (the 4:3 period: six outputs per lane) */
__device__ __forceinline__ void d3_h6s(int *d, const uint32_t (*p)[3], const uint32_t (*c)[3], int sh)
{
    asm("v_dot2_i32_i16 %0, %6, %24, 0\n\t"
        "v_dot2_i32_i16 %1, %7, %25, 0\n\t"
        "v_dot2_i32_i16 %2, %8, %26, 0\n\t"
        "v_dot2_i32_i16 %3, %9, %27, 0\n\t"
        "v_dot2_i32_i16 %4, %10, %28, 0\n\t"
        "v_dot2_i32_i16 %5, %11, %29, 0\n\t"
        "v_dot2_i32_i16 %0, %12, %30, %0\n\t"
        "v_dot2_i32_i16 %1, %13, %31, %1\n\t"
        "v_dot2_i32_i16 %2, %14, %32, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %33, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %34, %4\n\t"
        "v_dot2_i32_i16 %5, %17, %35, %5\n\t"
        "v_dot2_i32_i16 %0, %18, %36, %0\n\t"
        "v_dot2_i32_i16 %1, %19, %37, %1\n\t"
        "v_dot2_i32_i16 %2, %20, %38, %2\n\t"
        "v_dot2_i32_i16 %3, %21, %39, %3\n\t"
        "v_dot2_i32_i16 %4, %22, %40, %4\n\t"
        "v_dot2_i32_i16 %5, %23, %41, %5\n\t"
        "v_ashrrev_i32 %0, %42, %0\n\t"
        "v_ashrrev_i32 %1, %42, %1\n\t"
        "v_ashrrev_i32 %2, %42, %2\n\t"
        "v_ashrrev_i32 %3, %42, %3\n\t"
        "v_ashrrev_i32 %4, %42, %4\n\t"
        "v_ashrrev_i32 %5, %42, %5"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
        : "v"(p[0][0]), "v"(p[1][0]), "v"(p[2][0]), "v"(p[3][0]), "v"(p[4][0]), "v"(p[5][0]), "v"(p[0][1]), "v"(p[1][1]), "v"(p[2][1]), "v"(p[3][1]),
          "v"(p[4][1]), "v"(p[5][1]), "v"(p[0][2]), "v"(p[1][2]), "v"(p[2][2]), "v"(p[3][2]), "v"(p[4][2]), "v"(p[5][2]),
          "v"(c[0][0]), "v"(c[1][0]), "v"(c[2][0]), "v"(c[3][0]), "v"(c[4][0]), "v"(c[5][0]), "v"(c[0][1]), "v"(c[1][1]), "v"(c[2][1]), "v"(c[3][1]),
          "v"(c[4][1]), "v"(c[5][1]), "v"(c[0][2]), "v"(c[1][2]), "v"(c[2][2]), "v"(c[3][2]), "v"(c[4][2]), "v"(c[5][2]), "s"(sh));
}
__device__ __forceinline__ void d3_v6h(uint32_t *w, const uint32_t *pa, const uint32_t *pb, const uint32_t *pc, uint32_t c0, uint32_t c1, uint32_t c2, int seed,
                                       int sh, uint32_t maxpk, uint32_t msb)
{
    int t0, t1, t2, t3, t4, t5;
    asm("v_dot2_i32_i16 %3, %9, %27, %30\n\t"
        "v_dot2_i32_i16 %4, %10, %27, %30\n\t"
        "v_dot2_i32_i16 %5, %11, %27, %30\n\t"
        "v_dot2_i32_i16 %6, %12, %27, %30\n\t"
        "v_dot2_i32_i16 %7, %13, %27, %30\n\t"
        "v_dot2_i32_i16 %8, %14, %27, %30\n\t"
        "v_dot2_i32_i16 %3, %15, %28, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %28, %4\n\t"
        "v_dot2_i32_i16 %5, %17, %28, %5\n\t"
        "v_dot2_i32_i16 %6, %18, %28, %6\n\t"
        "v_dot2_i32_i16 %7, %19, %28, %7\n\t"
        "v_dot2_i32_i16 %8, %20, %28, %8\n\t"
        "v_dot2_i32_i16 %3, %21, %29, %3\n\t"
        "v_dot2_i32_i16 %4, %22, %29, %4\n\t"
        "v_dot2_i32_i16 %5, %23, %29, %5\n\t"
        "v_dot2_i32_i16 %6, %24, %29, %6\n\t"
        "v_dot2_i32_i16 %7, %25, %29, %7\n\t"
        "v_dot2_i32_i16 %8, %26, %29, %8\n\t"
        "v_ashrrev_i32 %3, %31, %3\n\t"
        "v_ashrrev_i32 %4, %31, %4\n\t"
        "v_ashrrev_i32 %5, %31, %5\n\t"
        "v_ashrrev_i32 %6, %31, %6\n\t"
        "v_ashrrev_i32 %7, %31, %7\n\t"
        "v_ashrrev_i32 %8, %31, %8\n\t"
        "v_cvt_pk_i16_i32 %0, %3, %4\n\t"
        "v_cvt_pk_i16_i32 %1, %5, %6\n\t"
        "v_cvt_pk_i16_i32 %2, %7, %8\n\t"
        "v_pk_max_i16 %0, %0, 0\n\t"
        "v_pk_max_i16 %1, %1, 0\n\t"
        "v_pk_max_i16 %2, %2, 0\n\t"
        "v_pk_min_i16 %0, %0, %32\n\t"
        "v_pk_min_i16 %1, %1, %32\n\t"
        "v_pk_min_i16 %2, %2, %32\n\t"
        "v_pk_lshlrev_b16 %0, %33, %0\n\t"
        "v_pk_lshlrev_b16 %1, %33, %1\n\t"
        "v_pk_lshlrev_b16 %2, %33, %2"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pb[4]), "v"(pb[5]),
          "v"(pc[0]), "v"(pc[1]), "v"(pc[2]), "v"(pc[3]), "v"(pc[4]), "v"(pc[5]), "s"(c0), "s"(c1), "s"(c2), "v"(seed), "s"(sh), "s"(maxpk), "s"(msb));
}

/* where the 6-tap window of output j of a period starts, from the period's first source sample: floor(((2j + 1) PIN - POUT) / (2 POUT)) - 2 */
template <int PIN, int POUT>
__host__ __device__ constexpr int d3_off(int j)
{
    const int n = (2 * j + 1) * PIN - POUT;
    return (n >= 0 ? n / (2 * POUT) : -((-n + 2 * POUT - 1) / (2 * POUT))) - 2;
}

/* PIN source samples become POUT outputs: (3, 2) — 1080p -> 720p, 4K -> 1440p — and (4, 3) — 1440p -> 1080p.  A lane owns NO = 2 POUT
 * outputs: two periods of a plane (2 PIN source samples + 4 of halo = PIN + 2 dwords at byte 4 PIN g - 4), one period x two channels of a
 * pair (PIN columns + 4 of halo = PIN + 4 dwords at byte 4 PIN g - 8).  Output y = POUT m + j ends on source row PIN m + off(j) + 5 and reads
 * P(that - 5), P(that - 3), P(that - 1); which outputs end on which row of a trip of T = lcm(6, PIN) rows is compile-time arithmetic. */
template <int PIN, int POUT, int PAIR>
__device__ __forceinline__ void d32h_unit(const FFHipD32Args &A, const FFHipD32Job &J, int frame, int gbase, int strip, int lane)
{
    constexpr int NO = 2 * POUT, NW = PAIR ? PIN + 4 : PIN + 2;
    const int graw = gbase + lane;
    const bool act = graw < J.ngroups;
    const int g = min(graw, J.ngroups - 1);
    const bool lb = g == 0, rb = g == J.ngroups - 1;
    const bool border = gbase == 0 || gbase + 64 >= J.ngroups; /* wave-uniform */
    /* the first / last lane of a row loads one dword (pair: two) further inside and rebuilds the replicated ones */
    const uint32_t soff = PAIR ? (uint32_t)(lb ? 0 : 4 * PIN * g - 8 - (rb ? 8 : 0)) : (uint32_t)(lb ? 0 : 4 * PIN * g - 4 - (rb ? 4 : 0));
    const int hsh = A.sdepth - 1, smsb = A.smsb ? 16 - A.sdepth : 0;
    const bool to8 = A.ddepth == 8; /* (3, 2) only: the host does not ask for it at (4, 3) */
    const uint32_t dmsb = (uint32_t)(A.dmsb ? 16 - A.ddepth : 0) * 0x00010001u;
    const int vsh = 27 - A.ddepth, vseed = 1 << (26 - A.ddepth);
    const uint32_t maxpk = (uint32_t)((1 << A.ddepth) - 1) * 0x00010001u;
    uint32_t cf[NO][3];
    {
        const uint32_t *p = J.hfv + (size_t)g * (PAIR ? 3 * POUT : 3 * NO);
#pragma unroll
        for (int j = 0; j < NO; j++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                cf[j][k] = p[3 * (PAIR ? j >> 1 : j) + k];
    }
    const int a = strip * J.strip_rows, b = min(a + J.strip_rows, J.dstH);
    const uint8_t *sbase = J.src + (size_t)frame * J.sfp;
    uint8_t *dbase = J.dst + (size_t)frame * J.dfp;
    const ptrdiff_t sstride = J.sstride, dstride = J.dstride;
    const int srcH = J.srcH;

    auto load_row = [&](int r, uint32_t (&w)[NW]) {
        const uint8_t *p = sbase + (ptrdiff_t)min(max(r, 0), srcH - 1) * sstride;
        const d3_u4 v = *(d3_gc4)((d3_gcp)p + soff);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        if (NW == 5) {
            w[4] = *(d3_gc1)((d3_gcp)p + soff + 16);
        } else if (NW == 6) {
            const d3_u2 e = *(d3_gc2)((d3_gcp)p + soff + 16);
            w[4] = e.x; w[5 % NW] = e.y;
        } else if (NW == 7) {
            const d3_u2 e = *(d3_gc2)((d3_gcp)p + soff + 16);
            w[4] = e.x; w[5 % NW] = e.y;
            w[NW - 1] = *(d3_gc1)((d3_gcp)p + soff + 24);
        } else {
            const d3_u4 e = *(d3_gc4)((d3_gcp)p + soff + 16);
            w[4] = e.x; w[5 % NW] = e.y; w[6 % NW] = e.z; w[7 % NW] = e.w;
        }
    };
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
                w[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(d3_h2, w[i]) >> (unsigned short)smsb);
        }
        uint32_t p[NO][3];
        if (PAIR) {
            /* column j of the lane's period reads columns off(j) + 2 .. + 5 more from the lane's base */
#pragma unroll
            for (int j = 0; j < POUT; j++)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int c0 = d3_off<PIN, POUT>(j) + 2 + 2 * k;
                    p[2 * j][k] = __builtin_amdgcn_perm(w[c0 + 1], w[c0], 0x05040100u);
                    p[2 * j + 1][k] = __builtin_amdgcn_perm(w[c0 + 1], w[c0], 0x07060302u);
                }
        } else {
            /* output i = POUT k + j of the lane reads samples s .. s + 5, s = PIN k + off(j) + 2 from the lane's base */
            uint32_t o[NW - 1];
#pragma unroll
            for (int i = 0; i < NW - 1; i++)
                o[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], 16);
#pragma unroll
            for (int i = 0; i < NO; i++)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int s0 = PIN * (i / POUT) + d3_off<PIN, POUT>(i % POUT) + 2 + 2 * k;
                    p[i][k] = (s0 & 1) ? o[s0 >> 1] : w[s0 >> 1];
                }
        }
        if (NO == 4)
            d3_h4s(h[0], h[1], h[2], h[3 % NO], reinterpret_cast<const uint32_t (&)[4][3]>(p), reinterpret_cast<const uint32_t (&)[4][3]>(cf), hsh);
        else
            d3_h6s(h, p, cf, hsh);
    };

    constexpr int T = PIN % 2 ? 2 * PIN : (PIN % 3 ? 3 * PIN : PIN); /* lcm(6, PIN) for PIN = 3, 4 */
    const int rbase0 = PIN * (a / POUT) - T, r_last = PIN * (b / POUT - 1) + d3_off<PIN, POUT>(POUT - 1) + 5, dstH = J.dstH;
    uint32_t ring[6][NO];
    int hprev[NO];
#pragma unroll
    for (int c = 0; c < NO; c++)
        hprev[c] = 0;
#pragma unroll
    for (int s = 0; s < 6; s++)
#pragma unroll
        for (int c = 0; c < NO; c++)
            ring[s][c] = 0;
    constexpr int D = 3; /* source rows in flight */
    uint32_t nxt[D][NW];
#pragma unroll
    for (int i = 0; i < D; i++)
        load_row(rbase0 + T - 2 + i, nxt[(T - 2 + i) % D]);
    typedef const uint32_t __attribute__((address_space(4))) *d3_cc;
    const d3_cc vt = (d3_cc)J.vfv;
    const uint32_t doff = (uint32_t)(2 * NO) * (uint32_t)g;

    auto emit = [&](int y, const uint32_t (&p0)[NO], const uint32_t (&p1)[NO], const uint32_t (&p2)[NO]) {
        const int yc = min(max(y, 0), dstH - 1);
        if (NO == 4 && to8) { /* uniform: an 8-bit target, four bytes per lane with the ordered dither's entry of every sample — (x + offset) & 7,
                               * offset 3 for the V channel / plane (yuv2nv12cX_c, vscale.c's chroma call) */
            const uint2 drow = *reinterpret_cast<const uint2 *>(d3_dither[yc & 7]);
            int sdv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int idx = (PAIR ? 2 * g + (i >> 1) + ((i & 1) ? 3 : 0) : 4 * g + i + J.dither_off) & 7;
                sdv[i] = (int)((((idx >= 4 ? drow.y : drow.x) >> (8 * (idx & 3))) & 255u) << 12);
            }
            const uint32_t o8 = d3_v4d(reinterpret_cast<const uint32_t (&)[4]>(p0), reinterpret_cast<const uint32_t (&)[4]>(p1),
                                       reinterpret_cast<const uint32_t (&)[4]>(p2), vt[4 * yc], vt[4 * yc + 1], vt[4 * yc + 2], sdv);
            if (act && y >= a && y < b)
                *(d3_g1)((d3_gp)(dbase + (ptrdiff_t)y * dstride) + 4u * (uint32_t)g) = o8;
            return;
        }
        uint32_t o[NO / 2];
        if (NO == 4)
            d3_v4h(o[0], o[1], reinterpret_cast<const uint32_t (&)[4]>(p0), reinterpret_cast<const uint32_t (&)[4]>(p1), reinterpret_cast<const uint32_t (&)[4]>(p2),
                   vt[4 * yc], vt[4 * yc + 1], vt[4 * yc + 2], vseed, vsh, maxpk, dmsb);
        else
            d3_v6h(o, p0, p1, p2, vt[4 * yc], vt[4 * yc + 1], vt[4 * yc + 2], vseed, vsh, maxpk, dmsb);
        if (act && y >= a && y < b) {
            d3_gp d = (d3_gp)(dbase + (ptrdiff_t)y * dstride) + doff;
            if (NO == 4) {
                *(d3_g2)d = (d3_u2){ o[0], o[1] };
            } else {
                *(d3_g2)d = (d3_u2){ o[0], o[1] };
                *(d3_g1)(d + 8) = o[2 % (NO / 2)];
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
        /* P(r - 1) = (row r - 1, row r), int16-saturated = min(., 32767) + truncation (no sum of an admitted bank falls below -32768) */
#pragma unroll
        for (int c = 0; c < NO; c++) {
            ring[(u + 5) % 6][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[c], h[c]));
            hprev[c] = h[c];
        }
        if (EMIT) {
#pragma unroll
            for (int j = 0; j < POUT; j++) {
                const int e = u - d3_off<PIN, POUT>(j) - 5; /* = PIN m - rbase */
                if (((e % PIN) + PIN) % PIN == 0)
                    emit(POUT * ((rbase + e) / PIN) + j, ring[(u + 1) % 6], ring[(u + 3) % 6], ring[(u + 5) % 6]);
            }
        }
    };
    step(rbase0, std::integral_constant<int, T - 2>(), std::false_type());
    step(rbase0, std::integral_constant<int, T - 1>(), std::false_type());
    for (int rbase = rbase0 + T; ; rbase += T) {
#define D3H_GO(u)                                                       \
        step(rbase, std::integral_constant<int, u>(), std::true_type()); \
        if (rbase + (u) + 1 > r_last) return;
        D3H_GO(0) D3H_GO(1) D3H_GO(2) D3H_GO(3) D3H_GO(4) D3H_GO(5)
        if constexpr (T == 12) {
            D3H_GO(6) D3H_GO(7) D3H_GO(8) D3H_GO(9) D3H_GO(10) D3H_GO(11)
        }
#undef D3H_GO
    }
}

__global__ __launch_bounds__(256) void k_sws_down32h(FFHipD32Args A)
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
    const FFHipD32Job &J = A.job[j];
    const int local = u - J.unit_begin;
    const int strip = local / J.ncb, cb = local - strip * J.ncb;
    if (A.ratio43) {
        if (J.pair)
            d32h_unit<4, 3, 1>(A, J, frame, cb * 64, strip, lane);
        else
            d32h_unit<4, 3, 0>(A, J, frame, cb * 64, strip, lane);
    } else if (J.pair) {
        d32h_unit<3, 2, 1>(A, J, frame, cb * 64, strip, lane);
    } else {
        d32h_unit<3, 2, 0>(A, J, frame, cb * 64, strip, lane);
    }
}

/* ================================================================================================== */
/* host side */

/*
 * Re-express a bank of an exact 3:2 down-scale (at most 6 taps) as coefficients on the REGULAR windows of the edge-replicated row:
 * output x reads samples clamp(3 (x >> 1) - 2 + (x & 1) + k), k = 0..5.  Every non-zero tap of the bank row must sit on one of those
 * samples; taps the reference folded onto the edge sample land on one of the replicas.  Output: n_dst x `pitch` dwords (pitch 3 or 4),
 * (c0, c1) (c2, c3) (c4, c5) as int16 pairs.  Returns 0 when the bank is not of this shape.
 */
int ffhip_d32_virtual_bank(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, int pitch, std::vector<uint32_t> *out, int pin, int pout)
{
    if ((long)pout * n_src != (long)pin * n_dst || fsize < 1 || fsize > 12 || (n_dst % pout) || pitch < 3 || !((pin == 3 && pout == 2) || (pin == 4 && pout == 3)))
        return 0;
    out->assign((size_t)n_dst * pitch, 0);
    for (int x = 0; x < n_dst; x++) {
        const int s0 = pin == 3 ? 3 * (x >> 1) - 2 + (x & 1) : 4 * (x / 3) + d3_off<4, 3>(x % 3);
        int16_t v[6] = { 0 };
        bool used[6] = { false };
        for (int i = 0; i < fsize; i++) {
            const int16_t c = filter[(size_t)x * fsize + i];
            if (!c)
                continue;
            const int p = pos[x] + i;
            if (p < 0 || p >= n_src)
                return 0;
            int k = 0;
            for (; k < 6; k++) {
                int q = s0 + k;
                q = q < 0 ? 0 : q >= n_src ? n_src - 1 : q;
                if (q == p && !used[k])
                    break;
            }
            if (k == 6)
                return 0;
            used[k] = true;
            v[k] = c;
        }
        for (int k = 0; k < 3; k++)
            (*out)[(size_t)pitch * x + k] = (uint16_t)v[2 * k] | ((uint32_t)(uint16_t)v[2 * k + 1] << 16);
    }
    return 1;
}

int ffhip_launch_down32(FFHipD32Args &A, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    /* strips of 32 output rows, shorter until the launch has the waves the chip keeps resident (a strip re-filters five source rows) */
    for (int want = 32; ; want >>= 1) {
        long long u = 0;
        for (int i = 0; i < A.njobs; i++) {
            FFHipD32Job &j = A.job[i];
            const int mult = A.hb && A.ratio43 ? 9 : 4; /* a strip starts on a whole trip of source rows */
            if (j.ngroups < 3 || j.dstH <= 0 || (j.dstH % (A.hb && A.ratio43 ? 3 : 2))) {
                ffhip_set_error("ffhip_sws: the exact-3:2 kernel takes rows of three groups or more and an even number of output rows");
                return FFHIP_EINVAL;
            }
            const int n = cdiv(j.dstH, want);
            j.strip_rows = cdiv(cdiv(j.dstH, n), mult) * mult;
            j.nstrips = cdiv(j.dstH, j.strip_rows);
            j.ncb = cdiv(j.ngroups, 64);
            u += (long long)j.ncb * j.nstrips;
        }
        if (u * A.nframes >= (A.hb ? 8192 : 4096) || want <= 8)
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
    if (A.hb)
        hipLaunchKernelGGL(k_sws_down32h, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, A);
    else
        hipLaunchKernelGGL(k_sws_down32, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, A);
    LAUNCH_CHECK();
    return 0;
}
