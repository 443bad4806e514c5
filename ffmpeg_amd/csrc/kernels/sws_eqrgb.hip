/*
 * sws_eqrgb.hip — 4:2:0 (yuv420p, NV12, NV21) into packed RGB AT THE SOURCE'S SIZE through the scaler (round 5): what
 * sws_scale() runs for NV12 -> rgb24 — the reference has no table converter for semi-planar sources, so the frame takes the generic
 * path (ff_swscale(), libswscale/swscale.c:259-560) with one-tap luma and horizontal banks and the 4-tap VERTICAL chroma bank that
 * brings the chroma lines up to a line per output line (chrDstH == dstH for packed targets, utils.c:1560-1575) — and for yuv420p
 * whenever the context is not given the table converter (SWS_ACCURATE_RND, swscale_unscaled.c:2425-2431).  The most common conversion
 * there is (a decoder's frame for a display or a network), and it ran on the general column walker at 0.355 of HBM.
 *
 * Arithmetic (bit for bit): hScale8To15_c with its one coefficient of 1 << 14 is s << 7 (swscale.c:128-142); yuv2rgb_X_c_template
 * (output.c:1789-1840) then has Y = (Y15 * 4096 + (1 << 18)) >> 19 = the source byte, and U, V = the vertical 4-tap sums of the chroma
 * lines >> 19; yuv2rgb_write() (output.c:1663-1787) as in sws_up2rgb.hip.
 *
 * Schedule: the vertical chroma bank is an exact 2x bank — row y reads chroma rows (y >> 1) - 2 + (y & 1) .. + 3 of the edge-replicated
 * plane (host: ffhip_up2_virtual_bank) — so chroma row c completes output rows 2c-3 and 2c-2, both on rows c-3 .. c: a ring of three
 * (row, row + 1) int16 pairs per chroma sample, three chroma rows per loop trip, every index a constant.  Per chroma row and lane
 * (4 U + 4 V samples, 8 pixels): 16 instructions make the pairs ((prev | cur << 16) << 7: one v_perm, one shift); per output row 16
 * dots, the chroma tables (LDS), 8 byte extracts + 24 v_mad_i32_i24 + 12 v_ashr_pk_u8_i32 for the pixels: ~11 VALU per pixel against
 * the walker's 25.  Rows leave through the wave's LDS tile in 16-byte pieces, non-temporal; a pack's frames share its waves lane by lane
 * (1920 pixels = 240 groups: four frames fill 15 waves).
 */
#include <stdlib.h>
#include <vector>

#include "common.h"
#include "sws_kernels.h"

typedef uint32_t er_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t er_u4 __attribute__((ext_vector_type(4)));
typedef const uint8_t __attribute__((address_space(1))) *er_gcp;
typedef uint8_t __attribute__((address_space(1))) *er_gp;
typedef const er_u2 __attribute__((address_space(1))) *er_gc2;
typedef const uint32_t __attribute__((address_space(1))) *er_gc1;
typedef er_u2 __attribute__((address_space(1))) *er_g2;
typedef er_u4 __attribute__((address_space(1))) *er_g4;
typedef const er_u4 __attribute__((address_space(4))) *er_cc4; /* constant address space: scalar loads */

/* the vertical chroma sums of one output row: pa / pb [0..3] U, [4..7] V row pairs; uv[m] = clip_u8(U[m] >> 19) | clip_u8(V[m] >> 19) << 8
 * in the low half.  The consumers of the sums sit in the block, >= 3 instructions behind the DOT that wrote their operand. */
__device__ __forceinline__ void er_vc4(uint32_t (&uv)[4], const uint32_t (&pa)[8], const uint32_t (&pb)[8], uint32_t f01, uint32_t f23, int seed)
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
        "v_ashr_pk_u8_i32 %0, %4, %8, 19\n\t"
        "v_ashr_pk_u8_i32 %1, %5, %9, 19\n\t"
        "v_ashr_pk_u8_i32 %2, %6, %10, 19\n\t"
        "v_ashr_pk_u8_i32 %3, %7, %11, 19"
        : "=&v"(uv[0]), "=&v"(uv[1]), "=&v"(uv[2]), "=&v"(uv[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6),
          "=&v"(t7)
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pa[6]), "v"(pa[7]),
          "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pb[4]), "v"(pb[5]), "v"(pb[6]), "v"(pb[7]),
          "s"(f01), "s"(f23), "v"(seed));
}
/* one dword of four clipped bytes (a, b, c, d) >> 16: the second instruction writes the high half and keeps the low one */
__device__ __forceinline__ uint32_t er_pk4(int a, int b, int c, int d)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 16 op_sel:[0,0,0,1]"
        : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}
__device__ __forceinline__ int er_mad24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void er_wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct ErRawC { uint32_t q[2]; };                 /* SIL: 4 (u, v) pairs; planar: q[0] = 4 U, q[1] = 4 V */
struct ErRawY { uint32_t q[2]; };                 /* 8 luma bytes */

/* LAY: 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra (alpha = 255).  SIL: the chroma plane is byte-interleaved (NV12; A.swap: NV21). */
template <int LAY, bool SIL>
__global__ __launch_bounds__(256, 4) void k_sws_eq_rgb(FFHipEqRgbArgs A)
{
    constexpr int NW = LAY < 2 ? 6 : 8; /* dwords of a lane's 8 pixels */
    __shared__ __attribute__((aligned(16))) uint32_t tiles[4][64 * NW];
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
    /* units: (pack of A.fpp frames, strip, lane block); lane index L = 64 * block + lane is group L % G of the pack's frame L / G */
    const uint32_t upp = (uint32_t)A.wpp * (uint32_t)A.nstrips;
    if (gw >= upp * (uint32_t)A.npacks)
        return;
    const int pack = (int)(gw / upp);
    const int u = (int)(gw - (uint32_t)pack * upp);
    const int strip = u / A.wpp, cb = u - strip * A.wpp;
    const int G = A.ngroups;
    const int f0 = pack * A.fpp, nf = min(A.fpp, A.nframes - f0);
    const int Lraw = cb * 64 + lane;
    const int L = min(Lraw, nf * G - 1);                   /* idle lanes shadow the last one */
    const int fs = (L >= G) + (L >= 2 * G) + (L >= 3 * G); /* A.fpp <= 4 */
    const int g = L - fs * G;
    const bool fullw = cb * 64 + 64 <= nf * G;             /* wave-uniform: every lane has a group */
    const uint32_t soffY = (uint32_t)fs * (uint32_t)A.sfp[0] + 8u * (uint32_t)g;
    const uint32_t soffU = (uint32_t)fs * (uint32_t)A.sfp[1] + (SIL ? 8u : 4u) * (uint32_t)g;
    const uint32_t soffV = SIL ? soffU : (uint32_t)fs * (uint32_t)A.sfp[2] + 4u * (uint32_t)g;

    const int S = A.steps_per_strip;
    const int a = 1 + strip * S, b = min(a + S, A.chrH + 2); /* chroma step c emits rows 2c-3 and 2c-2 */
    const int chrH = A.chrH, dstH = 2 * A.chrH;
    const uint8_t *sy = A.src[0] + (size_t)f0 * A.sfp[0];
    const uint8_t *su = A.src[1] + (size_t)f0 * A.sfp[1];
    const uint8_t *sv = SIL ? su : A.src[2] + (size_t)f0 * A.sfp[2];
    const ptrdiff_t ystride = A.sstride[0], ustride = A.sstride[1], vstride = SIL ? A.sstride[1] : A.sstride[2], dstride = A.dstride;

    int cr = a - 3; /* next chroma row to fetch (unclamped) */
    const uint8_t *pfu = su + (ptrdiff_t)min(max(cr, 0), chrH - 1) * ustride;
    const uint8_t *pfv = sv + (ptrdiff_t)min(max(cr, 0), chrH - 1) * vstride;
    int yr = 2 * a - 3; /* next luma row to fetch = next output row (row -1 of the first strip and row dstH of the last are not stored) */
    const uint8_t *pfy = sy + (ptrdiff_t)min(max(yr, 0), dstH - 1) * ystride;
    uint8_t *dr = A.dst + (size_t)f0 * A.dfp + (ptrdiff_t)(2 * a - 3) * dstride;
    asm("" : "+s"(pfy), "+s"(pfu), "+s"(pfv), "+s"(dr));

    auto load_chroma = [&](ErRawC &o) {
        uint32_t off = soffU, offv = soffV;
        asm volatile("" : "+v"(off), "+v"(offv));
        if (SIL) {
            const er_u2 w = *(er_gc2)((er_gcp)pfu + off);
            o.q[0] = w.x; o.q[1] = w.y;
        } else {
            o.q[0] = *(er_gc1)((er_gcp)pfu + off);
            o.q[1] = *(er_gc1)((er_gcp)pfv + offv);
        }
        cr++;
        const bool adv = cr >= 1 && cr <= chrH - 1; /* rows above / below the plane replicate the edge row */
        pfu += adv ? ustride : 0;
        if (!SIL)
            pfv += adv ? vstride : 0;
        asm("" : "+s"(pfu), "+s"(pfv));
    };
    auto load_luma = [&](ErRawY &o) {
        uint32_t off = soffY;
        asm volatile("" : "+v"(off));
        const er_u2 w = *(er_gc2)((er_gcp)pfy + off);
        o.q[0] = w.x; o.q[1] = w.y;
        yr++;
        pfy += (yr >= 1 && yr <= dstH - 1) ? ystride : 0;
        asm("" : "+s"(pfy));
    };

    /* chroma ring: (row, row + 1) pairs of 15-bit samples (byte << 7: hScale8To15_c with its one coefficient of 1 << 14), [0..3] U, [4..7] V */
    uint32_t cring[3][8];
    uint32_t cprev[2] = { 0, 0 }; /* the previous row's 4 U and 4 V bytes */
    const uint32_t sel_u = A.swap ? 0x07050301u : 0x06040200u, sel_v = A.swap ? 0x06040200u : 0x07050301u;
    auto cpass = [&](const ErRawC &w, uint32_t (&Pnew)[8]) {
        uint32_t cu = w.q[0], cv = w.q[1];
        if (SIL) { /* split the pairs: the channel at the even bytes, then the one at the odd bytes */
            cu = __builtin_amdgcn_perm(w.q[1], w.q[0], sel_u);
            cv = __builtin_amdgcn_perm(w.q[1], w.q[0], sel_v);
        }
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            const uint32_t cur = ch ? cv : cu, prev = cprev[ch];
#pragma unroll
            for (int i = 0; i < 4; i++) /* (prev byte i, cur byte i) as two zero-extended int16s, << 7 */
                Pnew[4 * ch + i] = __builtin_amdgcn_perm(cur, prev, 0x0c000c00u | (uint32_t)i | ((uint32_t)(4 + i) << 16)) << 7;
            cprev[ch] = cur;
        }
    };

    int kround = A.vround;
    asm volatile("" : "+v"(kround));
    const int cy = __builtin_amdgcn_readfirstlane(A.k.cy);
    uint32_t *tile = tiles[wave];
    const char *lutb = reinterpret_cast<const char *>(lut);
    /* the transposer's pieces: the bytes at offset o of the wave's tile are lane o / (4 NW)'s — where they go from the pack's row pointer */
    uint32_t poff[2];
    bool pok[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t o = i == 0 ? 16u * (uint32_t)lane : 1024u + (LAY >= 2 ? 16u : 8u) * (uint32_t)lane;
        const uint32_t ls = o / (4u * NW);
        const int Ls = cb * 64 + (int)ls;
        const int fl = (Ls >= G) + (Ls >= 2 * G) + (Ls >= 3 * G);
        pok[i] = Ls < nf * G;
        poff[i] = (uint32_t)fl * (uint32_t)A.dfp + 4u * NW * (uint32_t)(Ls - fl * G) + (o - 4u * NW * ls);
    }
    /* 24-bit pixels, an ODD number of groups in the wave (a row of 8 (2k + 1) pixels: one frame per pack then, ffhip_eqrgb_plan): the
     * valid bytes end in the middle of the last 16-byte piece — its first half leaves as an 8-byte store */
    const int vbytes = 4 * NW * min(nf * G - cb * 64, 64);
    const bool p0half = LAY < 2 && 16 * lane + 8 <= vbytes && 16 * lane + 16 > vbytes;
    if (LAY < 2 && 16 * lane + 16 > vbytes)
        pok[0] = false;
    auto st16 = [&](er_gp d, const er_u4 &v) { __builtin_nontemporal_store(v, (er_g4)d); };
    auto st8 = [&](er_gp d, const er_u2 &v) { __builtin_nontemporal_store(v, (er_g2)d); };

    /* one output row: its 8 luma bytes, the chroma pairs (rows s0, s0+1) (s0+2, s0+3), the row's coefficient dwords in SGPRs */
    auto emit = [&](const ErRawY &yq, const uint32_t (&Ca)[8], const uint32_t (&Cb)[8], uint32_t cf01, uint32_t cf23, bool store) {
        uint32_t uv[4];
        er_vc4(uv, Ca, Cb, cf01, cf23, kround);
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
            const int ys = (int)((yq.q[p >> 2] >> (8 * (p & 3))) & 0xffu);
            val[3 * p] = er_mad24(ys, cy, c0[p >> 1]);
            val[3 * p + 1] = er_mad24(ys, cy, c1[p >> 1]);
            val[3 * p + 2] = er_mad24(ys, cy, c2[p >> 1]);
        }
        uint32_t w[NW];
        if (LAY >= 2) {
            int alpha = 255 << 16;
            asm("" : "+v"(alpha));
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int x = val[3 * p], y = val[3 * p + 1], z = val[3 * p + 2]; /* (R, G, B) or, BGR layouts, (B, G, R) */
                w[p] = (LAY == 2 || LAY == 4) ? er_pk4(alpha, x, y, z) : er_pk4(x, y, z, alpha);
            }
        } else {
#pragma unroll
            for (int d = 0; d < 6; d++)
                w[d] = er_pk4(val[4 * d], val[4 * d + 1], val[4 * d + 2], val[4 * d + 3]);
        }
        uint32_t *t = tile + lane * NW;
        if (LAY >= 2) {
            *reinterpret_cast<uint4 *>(t) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4 *>(t + 4) = make_uint4(w[4 % NW], w[5 % NW], w[6 % NW], w[7 % NW]);
        } else {
            *reinterpret_cast<uint2 *>(t) = make_uint2(w[0], w[1]);
            *reinterpret_cast<uint2 *>(t + 2) = make_uint2(w[2], w[3]);
            *reinterpret_cast<uint2 *>(t + 4) = make_uint2(w[4], w[5]);
        }
        er_wave_sync_lds();
        const uint4 q0 = *reinterpret_cast<const uint4 *>(tile + lane * 4);
        uint4 q1 = make_uint4(0, 0, 0, 0);
        uint2 q2 = make_uint2(0, 0);
        if (LAY >= 2)
            q1 = *reinterpret_cast<const uint4 *>(tile + 256 + lane * 4);
        else
            q2 = *reinterpret_cast<const uint2 *>(tile + 256 + lane * 2);
        er_wave_sync_lds();
        if (store) { /* uniform */
            er_gp d = (er_gp)dr;
            er_u4 v0, v1;
            v0.x = q0.x; v0.y = q0.y; v0.z = q0.z; v0.w = q0.w;
            v1.x = q1.x; v1.y = q1.y; v1.z = q1.z; v1.w = q1.w;
            er_u2 v2;
            v2.x = q2.x; v2.y = q2.y;
            if (fullw || pok[0])
                st16(d + poff[0], v0);
            if (!fullw && p0half) {
                er_u2 h;
                h.x = q0.x; h.y = q0.y;
                st8(d + poff[0], h);
            }
            if (LAY >= 2) {
                if (fullw || pok[1])
                    st16(d + poff[1], v1);
            } else if (fullw || pok[1]) {
                st8(d + poff[1], v2);
            }
        }
    };

    /* ---- prologue: chroma rows a-3 .. a-1 into the ring, the next chroma row and the first two luma rows in flight ---- */
    ErRawC cnext;
    ErRawY y0, y1;
    load_chroma(cnext);
    {
        uint32_t seed[8];
        ErRawC cur = cnext;
        load_chroma(cnext);
        cpass(cur, seed); /* row a-3: only its samples matter (the low halves of the next pairs) */
        cur = cnext;
        load_chroma(cnext);
        cpass(cur, cring[1]); /* P[a-2] */
        cur = cnext;
        load_chroma(cnext);
        cpass(cur, cring[2]); /* P[a-1] */
    }
    load_luma(y0);
    load_luma(y1);

    /* vertical coefficients: row y at dwords 2 (y + 1), 2 (y + 1) + 1; a step reads rows 2c-3, 2c-2 = 4 consecutive dwords from 4c - 4 */
    const uint32_t *vt = A.vt;
    for (int c = a; c < b; c += 3) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (c + k < b) { /* uniform */
                const er_u4 cc = *(er_cc4)(vt + 4 * (c + k) - 4);
                const int y = 2 * (c + k) - 3;
                const ErRawC cur = cnext;
                load_chroma(cnext);
                cpass(cur, cring[k % 3]); /* P[c]: slot k; P[c-2]: slot (k + 1) % 3 */
                const ErRawY ya = y0, yb = y1;
                load_luma(y0);
                load_luma(y1);
                emit(ya, cring[(k + 1) % 3], cring[k % 3], cc.x, cc.y, y >= 0);
                dr += dstride;
                asm("" : "+s"(dr));
                emit(yb, cring[(k + 1) % 3], cring[k % 3], cc.z, cc.w, y + 1 < dstH);
                dr += dstride;
                asm("" : "+s"(dr));
            }
        }
    }
}

/* ================================================================================================== */
/* host side */

/* strips of about `want` chroma rows (a multiple of 3: the row loop is unrolled three times), evened out over the plane; frames per
 * pack: the 1, 2 or 4 (never more than the batch has) whose groups leave the fewest lanes of the pack's last wave idle */
void ffhip_eqrgb_plan(FFHipEqRgbArgs *a, int want, int fpp)
{
    const int steps = a->chrH + 1;
    const int n = cdiv(steps, want);
    const int s = cdiv(cdiv(steps, n), 3) * 3;
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
    if ((a->ngroups & 1) && a->lay < 2)
        best = 1; /* 24-byte groups: two frames would meet in the middle of a 16-byte piece */
    a->fpp = best;
    a->wpp = cdiv(best * a->ngroups, 64);
    a->npacks = cdiv(a->nframes > 0 ? a->nframes : 1, best);
}

int ffhip_launch_eqrgb(FFHipEqRgbArgs &A, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    const long long waves = (long long)A.wpp * A.nstrips * A.npacks;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define ER_LAUNCH(L) do { if (A.sil) hipLaunchKernelGGL((k_sws_eq_rgb<L, true>), grid, block, 0, stream, A); \
                          else hipLaunchKernelGGL((k_sws_eq_rgb<L, false>), grid, block, 0, stream, A); } while (0)
    switch (A.lay) {
    case 0: ER_LAUNCH(0); break;
    case 1: ER_LAUNCH(1); break;
    case 2: ER_LAUNCH(2); break;
    case 3: ER_LAUNCH(3); break;
    case 4: ER_LAUNCH(4); break;
    case 5: ER_LAUNCH(5); break;
    default:
        ffhip_set_error("ffhip_sws: packed layout %d is not one of the RGB writer's", A.lay);
        return FFHIP_EINVAL;
    }
#undef ER_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
