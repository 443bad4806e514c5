/*
 * sws_walk16.hip — the column walker for samples above 8 bits (round 4): the fused H+V scaler for 9..14-bit YUV formats (planar
 * yuv4xxpNN, P010 / P012 with their samples in the high bits) at ANY ratio whose banks have at most 8 taps in either direction —
 * 720p -> 1080p, 1080p -> 1440p, 4K -> 1440p, ...  Exact 2x / 2:1 keep their static-schedule kernels (sws_up2.hip / sws_down2.hip);
 * before this kernel everything else above 8 bits ran on the LDS-tiled k_sws_scale16 (0.05 - 0.08 of HBM).
 *
 * Arithmetic (the reference's, bit for bit):
 *   hScale16To15_c        libswscale/swscale.c:99-126    val = sum src[pos + j] * filter[j];  dst = FFMIN(val >> (depth - 1), 32767)
 *   yuv2planeX_10_c       libswscale/output.c:341-360    val = (1 << (26 - bits)) + sum line[j][i] * filter[j];  av_clip_uintp2(val >> (27 - bits), bits)
 *   yuv2p01xlX / cX       libswscale/output.c:478-529    the same, stored << (16 - bits)
 * int32 sums mod 2^32 (v_dot2_i32_i16 without clamp), samples read as int16 (<= 14 bits: positive).
 *
 * Design — the 8-bit walker's (sws_colwalk.hip), re-cut for two-byte samples:
 *   - one WAVE owns 64 lanes x 4 output columns of a strip of output rows and walks down the source rows; the 15-bit intermediate
 *     never leaves registers: per column a ring of VT - 1 vertically adjacent int16 PAIRS (h[r-1], h[r]) — an output row whose
 *     window ends at r is VT / 2 v_dot2 per sample on the pairs that end at r, r-2, ...; the ring index is static (the row loop is
 *     unrolled VT - 1 times);
 *   - the horizontal windows need NO unpacking: a lane loads the HT samples of each of its columns with one (2-byte aligned) global
 *     load of 2 HT bytes, and consecutive samples in a dword ARE the (s[k], s[k+1]) operand of v_dot2_i32_i16.  An interleaved
 *     (u, v) plane (P010) loads HT dwords per column and splits the channels with v_perm_b32.  Neighbouring columns' windows
 *     overlap: the loads are L1 / L2 hits, HBM sees every source row once per strip (+ VT - 1 halo rows);
 *   - banks are padded on the host to HT in {4, 8} and VT in {4, 8} taps (zero taps, positions pulled inside the plane at the far
 *     edge); a strip has at most 64 output rows, lane l keeps row l's vertical position and coefficient pairs and a row's
 *     descriptors are v_readlane away;
 *   - the next source row's samples are in flight while this one is filtered.
 * Algorithmic bytes: source in + destination out, 2 bytes per sample.
 */
#include <vector>

#include "common.h"
#include "sws_kernels.h"

typedef short w16_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short w16_h2 __attribute__((ext_vector_type(2)));
typedef uint32_t w16_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t w16_u4 __attribute__((ext_vector_type(4)));
typedef w16_u2 __attribute__((aligned(2))) w16_u2a;
typedef w16_u4 __attribute__((aligned(2))) w16_u4a;
typedef w16_u2 __attribute__((aligned(4))) w16_u2d;
typedef w16_u4 __attribute__((aligned(4))) w16_u4d;

/* ff_dither_8x8_128 (libswscale/swscale.c:42-52): the ordered dither of an 8-bit target fed from a deeper source (swscale.c:291,519-522) */
__constant__ __attribute__((aligned(8))) uint8_t w16_dither[8][8] = {
    {  36, 68,  60, 92,  34, 66,  58, 90, }, { 100,  4, 124, 28,  98,  2, 122, 26, }, {  52, 84,  44, 76,  50, 82,  42, 74, },
    { 116, 20, 108, 12, 114, 18, 106, 10, }, {  32, 64,  56, 88,  38, 70,  62, 94, }, {  96,  0, 120, 24, 102,  6, 126, 30, },
    {  48, 80,  40, 72,  54, 86,  46, 78, }, { 112, 16, 104,  8, 118, 22, 110, 14, },
};
/* clip_u8(a >> 19) | clip_u8(b >> 19) << 8 in the low half (the high half is not defined: common.h) */
__device__ __forceinline__ uint32_t w16_pk_u8(int a, int b)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 19" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ int w16_dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(w16_s2, a), __builtin_bit_cast(w16_s2, b), c, false);
}

/*
 * Hand-scheduled dot products, as in the 8-bit walkers: hipcc selects the accumulate-in-place VOP2 form v_dot2c_i32_i16 for the builtin,
 * which costs a v_mov per chain to seed the accumulator; the VOP3P form takes the seed as a third source.  Hazards inside an asm
 * block are ours (gfx950: a DOT result may feed the same opcode as src2 at once, any other VALU only after 3 wait states): four
 * chains are interleaved and a block ends in s_nop 2.  Operands are int16 pairs; a*[c] / b*[c] belong to chain c.
 */
/* d[c] = a0[c] . b0[c] + a1[c] . b1[c] */
__device__ __forceinline__ void w16_dots_first(int (&d)[4], const uint32_t (&a0)[4], const uint32_t (&a1)[4], const uint32_t (&b0)[4], const uint32_t (&b1)[4])
{
    asm("v_dot2_i32_i16 %0, %4, %12, 0\n\t"
        "v_dot2_i32_i16 %1, %5, %13, 0\n\t"
        "v_dot2_i32_i16 %2, %6, %14, 0\n\t"
        "v_dot2_i32_i16 %3, %7, %15, 0\n\t"
        "v_dot2_i32_i16 %0, %8, %16, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %17, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %18, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %19, %3\n\t"
        "s_nop 2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]),
          "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]));
}
/* d[c] += a0[c] . b0[c] + a1[c] . b1[c] */
__device__ __forceinline__ void w16_dots_more(int (&d)[4], const uint32_t (&a0)[4], const uint32_t (&a1)[4], const uint32_t (&b0)[4], const uint32_t (&b1)[4])
{
    asm("v_dot2_i32_i16 %0, %4, %12, %0\n\t"
        "v_dot2_i32_i16 %1, %5, %13, %1\n\t"
        "v_dot2_i32_i16 %2, %6, %14, %2\n\t"
        "v_dot2_i32_i16 %3, %7, %15, %3\n\t"
        "v_dot2_i32_i16 %0, %8, %16, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %17, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %18, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %19, %3\n\t"
        "s_nop 2"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
        : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]),
          "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]));
}
/* the vertical form: wave-uniform coefficient pairs (one SGPR operand per instruction); FIRST: d[c] = seed + ..., else d[c] += ... */
template <bool FIRST>
__device__ __forceinline__ void w16_vdots(int (&d)[4], const uint32_t (&a0)[4], const uint32_t (&a1)[4], uint32_t f0, uint32_t f1, int seed)
{
    if (FIRST)
        asm("v_dot2_i32_i16 %0, %4, %12, %14\n\t"
            "v_dot2_i32_i16 %1, %5, %12, %14\n\t"
            "v_dot2_i32_i16 %2, %6, %12, %14\n\t"
            "v_dot2_i32_i16 %3, %7, %12, %14\n\t"
            "v_dot2_i32_i16 %0, %8, %13, %0\n\t"
            "v_dot2_i32_i16 %1, %9, %13, %1\n\t"
            "v_dot2_i32_i16 %2, %10, %13, %2\n\t"
            "v_dot2_i32_i16 %3, %11, %13, %3\n\t"
            "s_nop 2"
            : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
            : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "s"(f0), "s"(f1), "v"(seed));
    else
        asm("v_dot2_i32_i16 %0, %4, %12, %0\n\t"
            "v_dot2_i32_i16 %1, %5, %12, %1\n\t"
            "v_dot2_i32_i16 %2, %6, %12, %2\n\t"
            "v_dot2_i32_i16 %3, %7, %12, %3\n\t"
            "v_dot2_i32_i16 %0, %8, %13, %0\n\t"
            "v_dot2_i32_i16 %1, %9, %13, %1\n\t"
            "v_dot2_i32_i16 %2, %10, %13, %2\n\t"
            "v_dot2_i32_i16 %3, %11, %13, %3\n\t"
            "s_nop 2"
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
            : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "s"(f0), "s"(f1));
}

/* NCH channels (1: a plane, 4 output columns per lane; 2: the two chroma channels of an interleaved source and / or target, 2 output
 * columns of each per lane — the same four samples and the same register budget either way) */
#define W16_SEG 1056 /* bytes of one staged row segment: 64 lanes x 16 bytes + the dword a window may read past its last sample */

/* STG (round 6): the source row segment under a wave's windows goes through LDS — one 16-byte global load per lane and row instead of
 * four to eight overlapping 8..16-byte loads (r04_walk16_pmc.txt: 43 % of the wave cycles were issue stalls behind them); wl = this wave's
 * 2 x W16_SEG bytes */
template <int HT, int VT, int NCH, bool STG>
__device__ __forceinline__ void w16_unit(const FFHipW16Job &J, const FFHipW16Args &A, int f, int strip, int cb, int lane, uint8_t *wl)
{
    constexpr int R = VT - 1;          /* ring slots: pairs that end at rows r, r-1, ..., r-(VT-2) */
    constexpr int HP = HT / 2, VP = VT / 2;
    constexpr int NC = NCH == 2 ? 2 : 4;
    const int X0 = (cb * 64 + lane) * NC;
    /* the job's fields the loops use, read ONCE: the job is picked by a run-time index out of the kernel arguments, and the compiler
     * re-read J.dstW / J.dstride[] from memory (s_load + s_waitcnt) for every output row */
    const int dstW = J.dstW, dstH = J.dstH, srcH = J.srcH;
    const ptrdiff_t ss0 = J.sstride[0], ss1 = J.sstride[NCH - 1], ds0 = J.dstride[0], ds1 = J.dstride[NCH - 1];
    const int y0 = strip * J.strip_rows, y1 = min(y0 + J.strip_rows, dstH);
    const bool d8 = A.ddepth == 8; /* an 8-bit target fed from the deeper source: yuv2planeX_8_c / yuv2nv12cX_c with the ordered dither */
    const int hsh = A.sdepth - 1, vsh = d8 ? 19 : 27 - A.ddepth;
    const int smsb = A.smsb ? 16 - A.sdepth : 0, dmsb = A.dmsb ? 16 - A.ddepth : 0;
    const int maxv = (1 << A.ddepth) - 1;
    const bool sil = J.sstep == 2, dil = J.dstep == 2;

    /* ---- horizontal descriptors of this lane's columns ---- */
    uint32_t soff[NC];         /* byte offset of the window's first sample in a source row (planar: of the even sample at or below it) */
    uint32_t sodd[NC];         /* planar: 2 when the window starts at an odd sample (the funnel shift), else 0 */
    uint32_t cf[NC][HP];
#pragma unroll
    for (int i = 0; i < NC; i++) {
        const int xi = min(X0 + i, dstW - 1);
        const uint32_t hp = (uint32_t)J.hp[xi];
        soff[i] = sil ? hp * 4u : (hp & ~1u) * 2u;
        sodd[i] = sil ? 0u : (hp & 1u) * 2u;
#pragma unroll
        for (int m = 0; m < HP; m++)
            cf[i][m] = reinterpret_cast<const uint32_t *>(J.hf)[(size_t)xi * HP + m];
    }
    const uint8_t *const sb0 = J.src[0] + (size_t)f * J.sfp[0], *const sb1 = NCH == 2 ? J.src[1] + (size_t)f * J.sfp[1] : sb0;
    uint8_t *const db0 = J.dst[0] + (size_t)f * J.dfp[0], *const db1 = NCH == 2 ? J.dst[1] + (size_t)f * J.dfp[1] : db0;

    /* the sample PAIRS of one source row, per channel and column: pr[ch][i][m] = (s[2m], s[2m + 1]) of the window */
    struct Row { uint32_t pr[NCH][NC][HP]; };
    auto load_row = [&](Row &o, int row) {
        const int rr = min(row, srcH - 1);
        if (NCH == 2 && sil) {
            /* HT (u, v) dwords per column, split into the two channels' pairs */
            const uint8_t *p = sb0 + (ptrdiff_t)rr * ss0;
#pragma unroll
            for (int i = 0; i < NC; i++) {
                uint32_t q[HT];
                const uint32_t so = soff[i];
#pragma unroll
                for (int k = 0; k < HT / 4; k++) { /* HT (u, v) dwords: 16 bytes at a time */
                    const w16_u4 v = *reinterpret_cast<const w16_u4d *>(p + so + 16 * k);
                    q[4 * k] = v.x; q[4 * k + 1] = v.y; q[4 * k + 2] = v.z; q[4 * k + 3] = v.w;
                }
#pragma unroll
                for (int m = 0; m < HP; m++) {
                    o.pr[0][i][m] = __builtin_amdgcn_perm(q[2 * m + 1], q[2 * m], 0x05040100u);
                    o.pr[NCH - 1][i][m] = __builtin_amdgcn_perm(q[2 * m + 1], q[2 * m], 0x07060302u);
                }
            }
            return;
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            /* a planar row: the window starts at an even or an odd sample.  ALIGNED dwords from the even sample below it (a
             * 2-byte-aligned 8-byte load is split by the texture addresser: PMC 46 % issue stalls), the last one only when the
             * window's last sample lies in it, and a funnel shift by the lane's 0 or 2 bytes */
            const uint8_t *p = (ch ? sb1 : sb0) + (ptrdiff_t)rr * (ch ? ss1 : ss0);
#pragma unroll
            for (int i = 0; i < NC; i++) {
                uint32_t q[HP + 1];
                const uint32_t so = soff[i]; /* (forcing the scalar-base form of global_load here — an opaque copy of the offset per
                                              * load — measured 4-8 % slower than the hoisted 64-bit lane addresses) */
                if (HT == 4) {
                    const w16_u2 v = *reinterpret_cast<const w16_u2d *>(p + so);
                    q[0] = v.x; q[1] = v.y;
                } else {
#pragma unroll
                    for (int k = 0; k < HP / 4; k++) { /* HP dwords: 16 bytes at a time (8 taps: one load, 16 taps: two) */
                        const w16_u4 v = *reinterpret_cast<const w16_u4d *>(p + so + 16 * k);
                        q[(4 * k) % (HP + 1)] = v.x; q[(4 * k + 1) % (HP + 1)] = v.y; q[(4 * k + 2) % (HP + 1)] = v.z; q[(4 * k + 3) % (HP + 1)] = v.w;
                    }
                }
                q[HP] = 0;
                if (sodd[i])
                    q[HP] = *reinterpret_cast<const uint32_t *>(p + so + 4 * HP);
#pragma unroll
                for (int m = 0; m < HP; m++)
                    o.pr[ch][i][m] = __builtin_amdgcn_alignbyte(q[m + 1], q[m], sodd[i]);
            }
        }
    };

    /* ---- STG: the wave's segment of a source row: global -> registers (one row ahead) -> LDS -> the lanes' windows ---- */
    constexpr int NPL = NCH == 2 ? 2 : 1;      /* source planes a row comes from (an interleaved source: one) */
    const int npl = NCH == 2 && !sil ? 2 : 1;
    const uint32_t seg0 = STG ? (uint32_t)__builtin_amdgcn_readfirstlane((int)soff[0]) & ~15u : 0u; /* positions ascend: the first lane's first window starts the segment */
    const uint32_t rowbytes = (uint32_t)J.srcW * (sil ? 4u : 2u);
    w16_u4 stg[NPL];
    auto stage_load = [&](int row) {
        const int rr = min(row, srcH - 1);
        const uint32_t o = seg0 + 16u * (uint32_t)lane;
#pragma unroll
        for (int p = 0; p < NPL; p++) {
            if (p >= npl)
                break;
            const uint8_t *rp = (p ? sb1 : sb0) + (ptrdiff_t)rr * (p ? ss1 : ss0);
            w16_u4 v = { 0, 0, 0, 0 };
            if (o + 16u <= rowbytes) {
                v = *reinterpret_cast<const w16_u4d *>(rp + o);
            } else if (o < rowbytes) { /* the row's ragged end: nothing is read past its last sample */
                uint32_t q[4] = { 0, 0, 0, 0 };
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (o + 4u * k + 4u <= rowbytes)
                        q[k] = *reinterpret_cast<const uint32_t *>(rp + o + 4 * k);
                    else if (o + 4u * k + 2u <= rowbytes)
                        q[k] = *reinterpret_cast<const uint16_t *>(rp + o + 4 * k);
                }
                v.x = q[0]; v.y = q[1]; v.z = q[2]; v.w = q[3];
            }
            stg[p] = v;
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int p = 0; p < NPL; p++) {
            if (p >= npl)
                break;
            *reinterpret_cast<w16_u4 *>(wl + p * W16_SEG + 16 * lane) = stg[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    /* the windows of the staged row, as load_row() leaves them */
    auto lds_row = [&](Row &o) {
        if (NCH == 2 && sil) {
#pragma unroll
            for (int i = 0; i < NC; i++) {
                uint32_t q[HT];
                const uint8_t *p = wl + (soff[i] - seg0);
#pragma unroll
                for (int k = 0; k < HT; k++)
                    q[k] = *reinterpret_cast<const uint32_t *>(p + 4 * k);
#pragma unroll
                for (int m = 0; m < HP; m++) {
                    o.pr[0][i][m] = __builtin_amdgcn_perm(q[2 * m + 1], q[2 * m], 0x05040100u);
                    o.pr[NCH - 1][i][m] = __builtin_amdgcn_perm(q[2 * m + 1], q[2 * m], 0x07060302u);
                }
            }
            return;
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
#pragma unroll
            for (int i = 0; i < NC; i++) {
                uint32_t q[HP + 1];
                const uint8_t *p = wl + ch * W16_SEG + (soff[i] - seg0);
#pragma unroll
                for (int k = 0; k <= HP; k++)
                    q[k] = *reinterpret_cast<const uint32_t *>(p + 4 * k);
#pragma unroll
                for (int m = 0; m < HP; m++)
                    o.pr[ch][i][m] = __builtin_amdgcn_alignbyte(q[m + 1], q[m], sodd[i]);
            }
        }
    };

    uint32_t ring[R][NCH][NC];
    int hprev[NCH][NC];
#pragma unroll
    for (int s = 0; s < R; s++)
#pragma unroll
        for (int ch = 0; ch < NCH; ch++)
#pragma unroll
            for (int i = 0; i < NC; i++)
                ring[s][ch][i] = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++)
#pragma unroll
        for (int i = 0; i < NC; i++)
            hprev[ch][i] = 0;

    /* the four chains of a lane, c = ch * NC + i, operands gathered pair by pair */
    uint32_t cfc[HP][4];
#pragma unroll
    for (int m = 0; m < HP; m++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            cfc[m][c] = cf[c % NC][m];
    auto hpass = [&](const Row &w, uint32_t (&Pnew)[NCH][NC]) {
        uint32_t sp[HP][4];
#pragma unroll
        for (int m = 0; m < HP; m++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                /* P01x keeps its samples in the high bits: one packed shift (by 0 for the other layouts — a select on top of it cost
                 * as much again) */
                sp[m][c] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(w16_h2, w.pr[c / NC][c % NC][m]) >> (unsigned short)smsb);
            }
        int acc[4];
        w16_dots_first(acc, sp[0], sp[1], cfc[0], cfc[1]);
#pragma unroll
        for (int m = 2; m < HP; m += 2)
            w16_dots_more(acc, sp[m % HP], sp[(m + 1) % HP], cfc[m % HP], cfc[(m + 1) % HP]);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int h = acc[c] >> hsh;
            /* saturating pack == FFMIN(., 32767) + truncation: no bank row can produce a sum below -32768 (host-checked) */
            Pnew[c / NC][c % NC] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[c / NC][c % NC], h));
            hprev[c / NC][c % NC] = h;
        }
    };

    /* ---- the walk ---- */
    /* the strip's vertical descriptors (<= 64 output rows): lane l holds row y0 + l's window position and coefficient pairs, a row's
     * are then one v_readlane each instead of a chain of dependent scalar loads per output row */
    int vpl;
    uint32_t vcl[VP];
    {
        const int y = min(y0 + lane, dstH - 1);
        vpl = J.vp[y];
#pragma unroll
        for (int m = 0; m < VP; m++)
            vcl[m] = reinterpret_cast<const uint32_t *>(J.vf)[(size_t)y * VP + m];
    }
    int yy = y0;
    /* where this lane's samples of output row yy go: advanced by the stride per row (a 64-bit multiply-add per row before) */
    const bool y16 = NCH == 1 && d8 && J.y16; /* the luma of a packed-RGB target's first stage: the sums >> 19 unclipped, as int16 (sws_y16rgb.hip) */
    uint8_t *dq0 = db0 + (ptrdiff_t)y0 * ds0 + (size_t)X0 * (y16 ? 2 : (NCH == 2 && dil ? 4 : 2) >> (d8 ? 1 : 0));
    uint8_t *dq1 = db1 + (ptrdiff_t)y0 * ds1 + (size_t)X0 * (d8 ? 1 : 2);
    /* d8: column c's entry of the dither row — (x + offset) & 7 with offset 3 for the V channel / plane (yuv2nv12cX_c, output.c:503-529;
     * vscale.c's chroma call): which dword of the row's eight bytes and how far in */
    bool dhi[4];
    int dsh[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int idx = (X0 + c % NC + (NCH == 2 ? (c / NC ? 3 : 0) : J.dither_off)) & 7;
        dhi[c] = idx >= 4;
        dsh[c] = 8 * (idx & 3);
    }
    const bool in_w = X0 < dstW, whole = X0 + (NCH == 2 ? 2 : 4) <= dstW;
    int need = __builtin_amdgcn_readlane(vpl, 0) + VT - 1;
    const int rlast = __builtin_amdgcn_readlane(vpl, y1 - 1 - y0) + VT - 1;
    int r = need - (VT - 1);
    Row cur, nxt;
    if (STG) {
        stage_load(r);
        stage_store();
        lds_row(cur);
        stage_load(r + 1);
    } else {
        load_row(cur, r);
    }
    const int kround = d8 ? 0 : 1 << (vsh - 1);
    while (r <= rlast) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int rr = r + k;
            if (rr <= rlast) { /* uniform */
                if (STG) {
                    stage_store();      /* row rr + 1, in flight since the previous step (the LDS reads of row rr were issued before: in order) */
                    stage_load(rr + 2);
                    lds_row(nxt);
                } else {
                    load_row(nxt, rr + 1);
                }
                hpass(cur, ring[k]);
                while (yy < y1 && need <= rr) {
                    /* the row's VT coefficients as VT / 2 pairs: wave-uniform, scalar loads */
                    uint32_t vc[VP];
#pragma unroll
                    for (int m = 0; m < VP; m++)
                        vc[m] = (uint32_t)__builtin_amdgcn_readlane((int)vcl[m], yy - y0);
                    uint32_t o[NCH][NC / 2];
                    {
                        /* pair m covers rows need - VT + 1 + 2m, + 2m + 1: it ends at rr - (VT - 2 - 2m) -> slot (k - (VT - 2 - 2m)) mod R */
                        uint32_t pp[VP][4];
#pragma unroll
                        for (int m = 0; m < VP; m++)
#pragma unroll
                            for (int c = 0; c < 4; c++)
                                pp[m][c] = ring[((k - (VT - 2 - 2 * m)) % R + R) % R][c / NC][c % NC];
                        int t[4];
                        w16_vdots<true>(t, pp[0], pp[1], vc[0], vc[1], kround);
#pragma unroll
                        for (int m = 2; m < VP; m += 2)
                            w16_vdots<false>(t, pp[m % VP], pp[(m + 1) % VP], vc[m % VP], vc[(m + 1) % VP], 0);
                        if (d8) { /* uniform */
                            /* seed dither << 12, >> 19, clip to 8 bits (yuv2planeX_8_c, output.c:468-486): four bytes per lane */
                            const uint2 drow = A.flat_dither == 2 ? make_uint2(0u, 0u) : A.flat_dither ? make_uint2(0x40404040u, 0x40404040u) : *reinterpret_cast<const uint2 *>(w16_dither[yy & 7]);
                            uint32_t b01, b23;
                            {
                                int z[4];
#pragma unroll
                                for (int c = 0; c < 4; c++)
                                    z[c] = t[c] + (int)((((dhi[c] ? drow.y : drow.x) >> dsh[c]) & 255u) << 12);
                                if (y16) { /* uniform */
                                    if (in_w) {
                                        const uint32_t w01 = ((uint32_t)(z[0] >> 19) & 0xffffu) | ((uint32_t)(z[1] >> 19) << 16);
                                        const uint32_t w23 = ((uint32_t)(z[2] >> 19) & 0xffffu) | ((uint32_t)(z[3] >> 19) << 16);
                                        if (whole) {
                                            *reinterpret_cast<uint2 *>(dq0) = make_uint2(w01, w23);
                                        } else {
                                            uint16_t *d16 = reinterpret_cast<uint16_t *>(dq0);
                                            d16[0] = (uint16_t)w01;
                                            if (X0 + 1 < dstW) d16[1] = (uint16_t)(w01 >> 16);
                                            if (X0 + 2 < dstW) d16[2] = (uint16_t)w23;
                                        }
                                    }
                                    yy++;
                                    dq0 += ds0;
                                    dq1 += ds1;
                                    if (yy < y1)
                                        need = __builtin_amdgcn_readlane(vpl, yy - y0) + VT - 1;
                                    continue;
                                }
                                b01 = w16_pk_u8(z[0], z[1]);
                                b23 = w16_pk_u8(z[2], z[3]);
                            }
                            if (in_w) {
                                if (NCH == 2 && dil) {           /* (u0, u1) (v0, v1) -> bytes u0 v0 u1 v1 */
                                    const uint32_t w = __builtin_amdgcn_perm(b23, b01, 0x05010400u);
                                    if (whole) *reinterpret_cast<uint32_t *>(dq0) = w;
                                    else *reinterpret_cast<uint16_t *>(dq0) = (uint16_t)w;
                                } else if (NCH == 2) {           /* two samples of each channel into its own plane */
                                    if (whole) {
                                        *reinterpret_cast<uint16_t *>(dq0) = (uint16_t)b01;
                                        *reinterpret_cast<uint16_t *>(dq1) = (uint16_t)b23;
                                    } else {
                                        dq0[0] = (uint8_t)b01;
                                        dq1[0] = (uint8_t)b23;
                                    }
                                } else {
                                    const uint32_t w = __builtin_amdgcn_perm(b23, b01, 0x05040100u);
                                    if (whole) {
                                        *reinterpret_cast<uint32_t *>(dq0) = w;
                                    } else {
                                        dq0[0] = (uint8_t)w;
                                        if (X0 + 1 < dstW) dq0[1] = (uint8_t)(w >> 8);
                                        if (X0 + 2 < dstW) dq0[2] = (uint8_t)(w >> 16);
                                    }
                                }
                            }
                            yy++;
                            dq0 += ds0;
                            dq1 += ds1;
                            if (yy < y1)
                                need = __builtin_amdgcn_readlane(vpl, yy - y0) + VT - 1;
                            continue;
                        }
                        /* >> (27 - bits), then the clip to 0 .. 2^bits - 1 on int16 pairs (v_cvt_pk_i16_i32 saturates to int16: the
                         * clip range lies inside) and P01x's alignment */
#pragma unroll
                        for (int c = 0; c < 4; c += 2) {
                            w16_s2 pk = __builtin_amdgcn_cvt_pk_i16(t[c] >> vsh, t[c + 1] >> vsh);
                            const w16_s2 zero = { 0, 0 }, top = { (short)maxv, (short)maxv };
                            pk = __builtin_elementwise_min(__builtin_elementwise_max(pk, zero), top);
                            uint32_t v = __builtin_bit_cast(uint32_t, pk);
                            if (dmsb)
                                v = __builtin_bit_cast(uint32_t, __builtin_bit_cast(w16_h2, v) << (unsigned short)dmsb);
                            o[c / NC][(c % NC) / 2] = v;
                        }
                    }
                    if (in_w) {
                        if (NCH == 2 && dil) {
                            /* (u0, v0) (u1, v1): 8 bytes */
                            uint8_t *d = dq0;
                            const uint32_t uv0 = __builtin_amdgcn_perm(o[NCH - 1][0], o[0][0], 0x05040100u);
                            const uint32_t uv1 = __builtin_amdgcn_perm(o[NCH - 1][0], o[0][0], 0x07060302u);
                            if (whole) {
                                w16_u2 s;
                                s.x = uv0; s.y = uv1;
                                *reinterpret_cast<w16_u2d *>(d) = s;
                            } else {
                                *reinterpret_cast<uint32_t *>(d) = uv0;
                            }
                        } else if (NCH == 2) {
                            /* two samples of each channel into its own plane */
                            uint8_t *du = dq0, *dv = dq1;
                            if (whole) {
                                *reinterpret_cast<uint32_t *>(du) = o[0][0];
                                *reinterpret_cast<uint32_t *>(dv) = o[NCH - 1][0];
                            } else {
                                *reinterpret_cast<uint16_t *>(du) = (uint16_t)o[0][0];
                                *reinterpret_cast<uint16_t *>(dv) = (uint16_t)o[NCH - 1][0];
                            }
                        } else {
                            uint8_t *d = dq0;
                            if (whole) {
                                w16_u2 s;
                                s.x = o[0][0]; s.y = o[0][NC / 2 - 1];
                                *reinterpret_cast<w16_u2a *>(d) = s;
                            } else { /* the ragged last group of a row (no array indexed by a loop counter: that would live in scratch memory) */
                                reinterpret_cast<uint16_t *>(d)[0] = (uint16_t)o[0][0];
                                if (X0 + 1 < dstW) reinterpret_cast<uint16_t *>(d)[1] = (uint16_t)(o[0][0] >> 16);
                                if (X0 + 2 < dstW) reinterpret_cast<uint16_t *>(d)[2] = (uint16_t)o[0][NC / 2 - 1];
                            }
                        }
                    }
                    yy++;
                    dq0 += ds0;
                    dq1 += ds1;
                    if (yy < y1)
                        need = __builtin_amdgcn_readlane(vpl, yy - y0) + VT - 1;
                }
                cur = nxt;
            }
        }
        r += R;
    }
}

template <int HT, int VT, bool STG = false>
__global__ __launch_bounds__(256) void k_sws_walk16(FFHipW16Args A)
{
    __shared__ __align__(16) uint8_t seg[STG ? 4 : 1][STG ? 2 * W16_SEG : 16];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave;
    if (gw >= (uint32_t)A.units_per_frame * (uint32_t)A.nframes)
        return;
    const int f = (int)(gw / (uint32_t)A.units_per_frame);
    const int u = (int)(gw - (uint32_t)f * (uint32_t)A.units_per_frame);
    int j = 0;
    if (A.njobs > 1 && u >= A.job[1].unit_begin) j = 1;
    if (A.njobs > 2 && u >= A.job[2].unit_begin) j = 2;
    const FFHipW16Job &J = A.job[j];
    const int local = u - J.unit_begin;
    const int strip = local / J.ncb, cb = local - strip * J.ncb;
    if (J.nch == 2) /* (both forms keep four samples per lane: the same register budget) */
        w16_unit<HT, VT, 2, STG>(J, A, f, strip, cb, lane, seg[STG ? wave : 0]);
    else
        w16_unit<HT, VT, 1, STG>(J, A, f, strip, cb, lane, seg[STG ? wave : 0]);
}

/* ================================================================================================== */
/* host side */

/* A bank of `size` taps as one of `T` taps (size <= T): zero taps appended, the window pulled back inside the plane at the far
 * edge (pos + T <= nsrc).  Returns false when the plane is narrower than T samples. */
bool ffhip_w16_pad_bank(const int16_t *filter, const int32_t *pos, int size, int n, int nsrc, int T, std::vector<int16_t> *of, std::vector<int32_t> *op)
{
    if (size > T || nsrc < T)
        return false;
    of->assign((size_t)n * T, 0);
    op->assign((size_t)n, 0);
    for (int x = 0; x < n; x++) {
        int p = pos[x];
        if (p < 0)
            return false;
        int np = p + T <= nsrc ? p : nsrc - T;
        const int sh = p - np;
        for (int j = 0; j < size; j++) {
            const int16_t c = filter[(size_t)x * size + j];
            if (!c)
                continue;
            if (sh + j >= T || p + j >= nsrc)
                return false;
            (*of)[(size_t)x * T + sh + j] = c;
        }
        (*op)[x] = np;
    }
    return true;
}

/* the widest span, in bytes from a 16-byte boundary, that the windows of `cols` adjacent columns cover in a source row (`bytes_per_column`:
 * 2 planar, 4 an interleaved pair; planar windows start at the even sample at or below theirs and may read the dword after their last
 * sample); 0 when the positions do not ascend */
int ffhip_w16_span(const int32_t *pos, int n, int cols, int bytes_per_column, int taps)
{
    int mx = 0;
    for (int x = 1; x < n; x++)
        if (pos[x] < pos[x - 1])
            return 0;
    for (int b = 0; b < n; b += cols) {
        const int first = pos[b], last = pos[(b + cols < n ? b + cols : n) - 1];
        const int lo = ((bytes_per_column == 2 ? first & ~1 : first) * bytes_per_column) & ~15;
        const int hi = bytes_per_column == 2 ? (last & ~1) * 2 + 2 * taps + 4 : (last + taps) * 4;
        if (hi - lo > mx)
            mx = hi - lo;
    }
    return mx;
}

void ffhip_w16_plan_job(FFHipW16Job *j, int strip_target)
{
    j->ncb = cdiv(j->dstW, j->nch == 2 ? 128 : 256); /* 64 lanes x 4 columns of a plane / 2 columns of both channels */
    const int n = cdiv(j->dstH, strip_target > 0 ? strip_target : 1);
    j->strip_rows = cdiv(j->dstH, n);
    j->nstrips = cdiv(j->dstH, j->strip_rows);
}

int ffhip_launch_walk16(FFHipW16Args &A, hipStream_t stream)
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
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    bool stg = A.ht <= 8 && A.vt <= 8;
    for (int i = 0; i < A.njobs; i++)
        stg = stg && A.job[i].stage;
    if (stg) { /* round 6: source row segments through LDS */
        if (A.ht == 4 && A.vt == 4)      hipLaunchKernelGGL((k_sws_walk16<4, 4, true>), grid, block, 0, stream, A);
        else if (A.ht == 8 && A.vt == 4) hipLaunchKernelGGL((k_sws_walk16<8, 4, true>), grid, block, 0, stream, A);
        else if (A.ht == 4 && A.vt == 8) hipLaunchKernelGGL((k_sws_walk16<4, 8, true>), grid, block, 0, stream, A);
        else                             hipLaunchKernelGGL((k_sws_walk16<8, 8, true>), grid, block, 0, stream, A);
        LAUNCH_CHECK();
        return 0;
    }
    if (A.ht == 4 && A.vt == 4)
        hipLaunchKernelGGL((k_sws_walk16<4, 4>), grid, block, 0, stream, A);
    else if (A.ht == 8 && A.vt == 4)
        hipLaunchKernelGGL((k_sws_walk16<8, 4>), grid, block, 0, stream, A);
    else if (A.ht == 4 && A.vt == 8)
        hipLaunchKernelGGL((k_sws_walk16<4, 8>), grid, block, 0, stream, A);
    else if (A.ht == 8 && A.vt == 8)
        hipLaunchKernelGGL((k_sws_walk16<8, 8>), grid, block, 0, stream, A);
    /* round 5: banks of 9..16 taps (ratios down to 1/4: a 4K HDR frame into 720p) */
    else if (A.ht == 16 && A.vt == 16)
        hipLaunchKernelGGL((k_sws_walk16<16, 16>), grid, block, 0, stream, A);
    else if (A.ht == 16 && A.vt == 8)
        hipLaunchKernelGGL((k_sws_walk16<16, 8>), grid, block, 0, stream, A);
    else if (A.ht == 8 && A.vt == 16)
        hipLaunchKernelGGL((k_sws_walk16<8, 16>), grid, block, 0, stream, A);
    else {
        ffhip_set_error("ffhip_sws: no 16-bit column walker for %d x %d taps", A.ht, A.vt);
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}
