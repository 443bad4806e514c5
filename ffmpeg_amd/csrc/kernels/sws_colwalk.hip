/*
 * sws_colwalk.hip — the fast path of the fused H+V scaler for 4-tap x 4-tap banks (bicubic / bilinear
 * up-scaling: BASELINE config "nv12 1080p -> 4K bicubic"), planar and NV12/NV21 in and out.
 *
 * Same arithmetic as sws_scale.hip (hScale8To15_c, libswscale/swscale.c:128-142; yuv2planeX_8_c /
 * yuv2nv12cX_c, libswscale/output.c:468-529; nv12ToUV_c, input.c:936) but no LDS and no barrier:
 *
 *   one WAVE owns 64 lanes x 4 output columns (x NG groups) of a strip of output rows and walks DOWN
 *   the source rows.  Per source row a lane loads the 8 source bytes its 4 columns read (its 4-tap
 *   windows start within 4 bytes of each other — any ratio >= 0.75: checked on the host, else the LDS-tiled
 *   kernel runs; the span is dword aligned whenever the windows allow it, e.g. always at 2x), computes the 4 horizontal 15-bit samples with v_perm_b32 (bytes -> int16 pairs) +
 *   v_dot2_i32_i16 against its register-resident coefficients, and keeps for every column the last
 *   three vertically adjacent int16 PAIRS (h[r-3],h[r-2]) (h[r-2],h[r-1]) (h[r-1],h[r]).  An output row
 *   whose 4-tap vertical window ends at r is then two more v_dot2 per sample from those pairs against
 *   the row's (wave-uniform) coefficient pairs, v_ashr_pk_u8_i32 to shift/clamp/pack, one store.
 *   The 15-bit intermediate lives in registers only; source rows are read once per strip (+3 halo
 *   rows), HBM traffic = source in + destination out.
 *
 * Integer semantics are the reference's: int32 accumulation, >>7 and min(.,32767) then truncation to
 * int16 for the horizontal pass; unsigned-wrapping int32 accumulation from 64<<12, >>19 (arithmetic),
 * clip to u8 for the vertical pass.  v_dot2_i32_i16 without clamp is exactly that mod 2^32.
 */
#include <stdlib.h>
#include <vector>

#include "common.h"
#include "sws_kernels.h"

typedef short cw_short2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int cw_dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(cw_short2, a), __builtin_bit_cast(cw_short2, b), c, false);
}

/* {clip_u8(a >> 19), clip_u8(b >> 19)} in bits 0..15; bits 16..31 are unspecified (see common.h) */
template <bool PLAIN>
__device__ __forceinline__ uint32_t cw_pk_u8(int a, int b)
{
    uint32_t r;
    if (PLAIN)
        r = (uint32_t)clip_u8(a >> 19) | ((uint32_t)clip_u8(b >> 19) << 8);
    else
        asm("v_ashr_pk_u8_i32 %0, %1, %2, 19" : "=v"(r) : "v"(a), "v"(b));
    return r;
}


/*
 * Hand-scheduled dot products (OPT variant).  hipcc only selects the accumulate-in-place VOP2 form
 * v_dot2c_i32_i16 for __builtin_amdgcn_sdot2, which costs a v_mov per chain to seed the accumulator;
 * the VOP3P form v_dot2_i32_i16 takes the seed as a third source (inline 0, or a register).  Hazards are
 * ours inside asm (gfx90a+/gfx950: a DOT result may feed the SAME opcode as src2 at once, any other VALU
 * only after 3 wait states — LLVM GCNHazardRecognizer::checkMAIVALUHazards): the 4 chains are interleaved
 * and the block ends in s_nop 2, so whatever follows is safe.
 */
__device__ __forceinline__ void cw_hdots4(int (&d)[4], const uint32_t (&a)[4], const uint32_t (&b)[4], const uint32_t *cf)
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
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
          "v"(cf[0]), "v"(cf[2]), "v"(cf[4]), "v"(cf[6]), "v"(cf[1]), "v"(cf[3]), "v"(cf[5]), "v"(cf[7]));
}

/* d[i] = Pa[i] . f01 + Pb[i] . f23 + kround; f01/f23 wave-uniform (one SGPR operand per instruction) */
__device__ __forceinline__ void cw_vdots4(int (&d)[4], const uint32_t (&pa)[4], const uint32_t (&pb)[4], uint32_t f01,
                                          uint32_t f23, int kround)
{
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
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]),
          "s"(f01), "s"(f23), "v"(kround));
}

typedef short cw_v2i16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cw_opaque(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}

struct CwRaw { uint32_t q[4]; };

/* global (address space 1) views: an opaque SGPR pointer would otherwise decay to flat accesses */
typedef const uint8_t __attribute__((address_space(1))) *cw_gcptr;
typedef uint8_t __attribute__((address_space(1))) *cw_gptr;
typedef uint32_t cw_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t cw_u4 __attribute__((ext_vector_type(4)));
typedef const cw_u2 __attribute__((address_space(1))) *cw_gc2;
typedef const cw_u4 __attribute__((address_space(1))) *cw_gc4;
typedef cw_u2 __attribute__((address_space(1))) *cw_g2;
typedef uint32_t __attribute__((address_space(1))) *cw_g1;

/* raw source bytes of one row for this lane: 8 or 16 bytes at row + byte_base (dword aligned) */
template <int NRAW>
__device__ __forceinline__ void cw_load(CwRaw &o, int slot, const uint8_t *row, uint32_t byte_base)
{
    cw_gcptr g = (cw_gcptr)row;
    /* keep the zero-extension of the lane offset next to the access (an empty volatile asm cannot be hoisted out
     * of the row loop): only then does instruction selection see `uniform base + zext(32-bit lane offset)` and
     * pick the global_load saddr form instead of a 64-bit VALU add per access */
    asm volatile("" : "+v"(byte_base));
    if (NRAW == 16) {
        const cw_u4 w = *(cw_gc4)(g + byte_base);
        o.q[0] = w.x; o.q[1] = w.y; o.q[2] = w.z; o.q[3] = w.w;
    } else {
        const cw_u2 w = *(cw_gc2)(g + byte_base);
        o.q[slot] = w.x; o.q[slot + 1] = w.y;
    }
}

/*
 * First byte of the 8-byte source span a 4-column group reads per row: dword aligned when the four windows fit
 * such a span (2x up-scaling: always), else the lowest window start itself — the host has checked that the
 * starts lie within 4 bytes of each other, and global loads need no alignment.  Spans that would cross the end of
 * the row slide left (the columns then select from the upper bytes).
 */
__device__ __forceinline__ int cw_span_base(const int (&p)[4], int srcW)
{
    const int lo = min(min(p[0], p[1]), min(p[2], p[3])), hi = max(max(p[0], p[1]), max(p[2], p[3]));
    int base = lo & ~3;
    if (hi + 3 - base > 7)
        base = lo;
    return min(base, srcW - 8);
}

/*
 * One unit.  KIND 0: one plane, one 4-column group per lane.  KIND 1: one plane, two adjacent groups
 * per lane.  KIND 2/3/4: a U/V pair, one group of each per lane — 2: interleaved -> interleaved,
 * 3: interleaved -> planar, 4: planar -> interleaved (planar -> planar is two KIND 0/1 jobs).
 * D = source rows in flight (prefetch depth), a multiple of 3 so that the ring of vertical pairs
 * is indexed statically inside the unrolled row loop.
 */
template <int KIND, int D, bool PLAIN, bool OPT, bool DUP = false>
__device__ __forceinline__ void cw_unit(const FFHipCwJob &J, int f, int strip, int cb, int lane)
{
    constexpr int NG = KIND == 0 ? 1 : 2;           /* groups per lane                 */
    constexpr int NH = KIND == 1 ? 2 : 1;           /* distinct horizontal descriptors */
    const int y0 = strip * J.strip_rows;
    const int y1 = min(y0 + J.strip_rows, J.dstH);

    /* ---- per-lane horizontal descriptors ---------------------------------------------------- */
    int X0[NH], base[NH];
    uint32_t sel[NH][8], cf[NH][8];
#pragma unroll
    for (int g = 0; g < NH; g++) {
        X0[g] = ((cb * 64 + lane) * NH + g) * 4;
        int p[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int xi = min(X0[g] + i, J.dstW - 1);
            p[i] = J.hp[xi];
            const uint2 c = *reinterpret_cast<const uint2 *>(J.hf + (size_t)xi * 4);
            cf[g][2 * i] = c.x;
            cf[g][2 * i + 1] = c.y;
        }
        base[g] = cw_span_base(p, J.srcW);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t o = (uint32_t)(p[i] - base[g]); /* 0..4, host-checked */
            sel[g][2 * i]     = 0x0c000c00u | o | ((o + 1) << 16);
            sel[g][2 * i + 1] = 0x0c000c00u | (o + 2) | ((o + 3) << 16);
        }
    }
    const bool act = X0[0] < J.dstW;

    /* ---- source / destination row addressing ----------------------------------------------- */
    constexpr bool PAIR = KIND >= 2, sil = KIND == 2 || KIND == 3, dil = KIND == 2 || KIND == 4;
    const uint8_t *s0 = J.src[0] + (size_t)f * J.sfp[0];
    const uint8_t *s1 = PAIR ? J.src[1] + (size_t)f * J.sfp[1] : s0;
    const ptrdiff_t sstride0 = J.sstride[0], sstride1 = J.sstride[1], dstride0 = J.dstride[0], dstride1 = J.dstride[1];
    const int dstW = J.dstW;
    /* NV21: V is the first byte of a pair — only the byte selectors change */
    const uint32_t sel_u = J.src_swap ? 0x07050301u : 0x06040200u, sel_v = J.src_swap ? 0x06040200u : 0x07050301u;
    const uint32_t sel_uv = J.dst_swap ? 0x04050001u : 0x05040100u;
    /* unsigned lane offsets: uniform row pointer + 32-bit lane offset addressing */
    const uint32_t bb0 = sil ? 2 * base[0] : base[0];
    const uint32_t bb1 = KIND == 1 ? base[1] : bb0;

    auto rowptr = [&](const uint8_t *base, int r, ptrdiff_t stride) { return base + (ptrdiff_t)r * stride; };
    auto load_row = [&](CwRaw &o, int r) {
        if (sil) {
            cw_load<16>(o, 0, rowptr(s0, r, sstride0), bb0);
        } else if (PAIR) {
            cw_load<8>(o, 0, rowptr(s0, r, sstride0), bb0);
            cw_load<8>(o, 2, rowptr(s1, r, sstride1), bb0);
        } else {
            const uint8_t *rp = rowptr(s0, r, sstride0);
            cw_load<8>(o, 0, rp, bb0);
            if (KIND == 1)
                cw_load<8>(o, 2, rp, bb1);
        }
    };

    /* ring of vertical pairs: slot (k % 3) receives (h[r-1], h[r]) of the row handled at unroll step k */
    uint32_t Pw[3][NG][4];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                Pw[t][g][i] = 0;

    int hprev[NG][4]; /* OPT: last row's (sum >> 7), unsaturated */
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            hprev[g][i] = 0;
    int kround = 64 << 12; /* OPT: the rounding seed of the vertical sums, pinned in a VGPR */
    if (OPT)
        asm volatile("" : "+v"(kround));

    auto hpass = [&](const CwRaw &w, uint32_t (&Pnew)[NG][4], const uint32_t (&Pprev)[NG][4]) {
        uint32_t d[NG][2];
        if (sil) {
            /* de-interleave: even bytes -> first channel in memory, odd -> second (nv12ToUV_c) */
            d[0][0] = __builtin_amdgcn_perm(w.q[1], w.q[0], sel_u);
            d[0][1] = __builtin_amdgcn_perm(w.q[3], w.q[2], sel_u);
            d[NG - 1][0] = __builtin_amdgcn_perm(w.q[1], w.q[0], sel_v);
            d[NG - 1][1] = __builtin_amdgcn_perm(w.q[3], w.q[2], sel_v);
        } else {
#pragma unroll
            for (int g = 0; g < NG; g++) {
                d[g][0] = w.q[2 * g];
                d[g][1] = w.q[2 * g + 1];
            }
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int hg = KIND == 1 ? g : 0;
            if (OPT) {
                uint32_t a[4], b[4];
                int acc[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (DUP && i == 2) {
                        /* columns 1 and 2 of every group read the same window (exact 2x up-scaling, host-checked) */
                        a[2] = a[1];
                        b[2] = b[1];
                    } else {
                        a[i] = __builtin_amdgcn_perm(d[g][1], d[g][0], sel[hg][2 * i]);
                        b[i] = __builtin_amdgcn_perm(d[g][1], d[g][0], sel[hg][2 * i + 1]);
                    }
                }
                cw_hdots4(acc, a, b, cf[hg]);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    /* (h[r-1], h[r]) with both halves saturated to int16: equals min(.,32767) + truncation
                     * because the host has checked that no horizontal sum can fall below -32768 */
                    const int h = acc[i] >> 7;
                    Pnew[g][i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[g][i], h));
                    hprev[g][i] = h;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t a = __builtin_amdgcn_perm(d[g][1], d[g][0], sel[hg][2 * i]);
                    const uint32_t b = __builtin_amdgcn_perm(d[g][1], d[g][0], sel[hg][2 * i + 1]);
                    int acc = cw_dot2(a, cf[hg][2 * i], 0);
                    acc = cw_dot2(b, cf[hg][2 * i + 1], acc);
                    const uint32_t h = (uint32_t)min(acc >> 7, 32767);
                    Pnew[g][i] = __builtin_amdgcn_perm(h, Pprev[g][i], 0x05040302); /* (prev.hi16, h.lo16) */
                }
            }
        }
    };

    uint8_t *d0 = J.dst[0] + (size_t)f * J.dfp[0];
    uint8_t *d1 = PAIR ? J.dst[1] + (size_t)f * J.dfp[1] : d0;
    const uint32_t dc0 = (dil ? 2 : 1) * X0[0], dc1 = X0[0]; /* this lane's first byte of a row */

    /* Pa = pairs (h[p], h[p+1]), Pb = (h[p+2], h[p+3]) of the row's window */
    auto emit = [&](uint8_t *r0p, uint8_t *r1p, uint32_t f01, uint32_t f23, const uint32_t (&Pa)[NG][4],
                    const uint32_t (&Pb)[NG][4]) {
        int v[NG][4];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            if (OPT) {
                cw_vdots4(v[g], Pa[g], Pb[g], f01, f23, kround);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int acc = cw_dot2(Pa[g][i], f01, 64 << 12);
                    v[g][i] = cw_dot2(Pb[g][i], f23, acc);
                }
            }
        }
        if (dil) {
            /* yuv2nv12cX_c: bytes U0 V0 U1 V1 ... (V first for NV21) */
            constexpr int b = NG - 1;
            const uint32_t k0 = cw_pk_u8<PLAIN>(v[0][0], v[b][0]), k1 = cw_pk_u8<PLAIN>(v[0][1], v[b][1]);
            const uint32_t k2 = cw_pk_u8<PLAIN>(v[0][2], v[b][2]), k3 = cw_pk_u8<PLAIN>(v[0][3], v[b][3]);
            cw_u2 w;
            w.x = __builtin_amdgcn_perm(k1, k0, sel_uv);
            w.y = __builtin_amdgcn_perm(k3, k2, sel_uv);
            if (act)
                *(cw_g2)((cw_gptr)r0p + cw_opaque(dc0)) = w;
        } else if (KIND == 1) {
            /* two adjacent groups: one 8-byte store when both exist (dstW % 4 == 0, host-checked) */
            cw_u2 w;
            w.x = __builtin_amdgcn_perm(cw_pk_u8<PLAIN>(v[0][2], v[0][3]), cw_pk_u8<PLAIN>(v[0][0], v[0][1]), 0x05040100);
            w.y = __builtin_amdgcn_perm(cw_pk_u8<PLAIN>(v[NG - 1][2], v[NG - 1][3]),
                                        cw_pk_u8<PLAIN>(v[NG - 1][0], v[NG - 1][1]), 0x05040100);
            cw_gptr d = (cw_gptr)r0p + cw_opaque(dc0);
            if (X0[0] + 8 <= dstW)
                *(cw_g2)d = w;
            else if (act)
                *(cw_g1)d = w.x;
        } else {
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const uint32_t w = __builtin_amdgcn_perm(cw_pk_u8<PLAIN>(v[g][2], v[g][3]),
                                                         cw_pk_u8<PLAIN>(v[g][0], v[g][1]), 0x05040100);
                cw_gptr d = g ? (cw_gptr)r1p + cw_opaque(dc1) : (cw_gptr)r0p + cw_opaque(dc0);
                if (act)
                    *(cw_g1)d = w;
            }
        }
    };

    /* ---- vertical descriptors of the strip's <= 128 output rows: two registers each, read by v_readlane.
     * Loaded before the row loop so that no vector-memory wait inside it has to drain the prefetches. ---- */
    int vpl[2];
    uint32_t vf01[2], vf23[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int y = min(y0 + 64 * k + lane, J.dstH - 1);
        vpl[k] = J.vp[y];
        const uint2 c = *reinterpret_cast<const uint2 *>(J.vf + (size_t)y * 4);
        vf01[k] = c.x;
        vf23[k] = c.y;
    }

    /* ---- walk down the source rows ------------------------------------------------------------ */
    const int ny = y1 - y0;
    int yy = 0; /* next output row of the strip to emit */
    int need = __builtin_amdgcn_readlane(vpl[0], 0) + 3;
    const int rlast = __builtin_amdgcn_readfirstlane(J.vp[y1 - 1]) + 3;
    int r = need - 3;
    /* Scalar running pointers (two SALU adds per step instead of a 64-bit multiply per access): pf* = the next
     * source row to prefetch (stops advancing at rlast, which makes the clamped prefetch unconditional), dr* =
     * the next destination row.  Kept opaque so they stay in SGPRs. */
    int pfrow = r;
    const uint8_t *pf0 = s0 + (ptrdiff_t)r * sstride0, *pf1 = s1 + (ptrdiff_t)r * sstride1;
    uint8_t *dr0 = d0 + (ptrdiff_t)y0 * dstride0, *dr1 = d1 + (ptrdiff_t)y0 * dstride1;
    asm("" : "+s"(pf0), "+s"(pf1), "+s"(dr0), "+s"(dr1));
    auto load_next = [&](CwRaw &o) {
        if (sil) {
            cw_load<16>(o, 0, pf0, bb0);
        } else if (PAIR) {
            cw_load<8>(o, 0, pf0, bb0);
            cw_load<8>(o, 2, pf1, bb0);
        } else {
            cw_load<8>(o, 0, pf0, bb0);
            if (KIND == 1)
                cw_load<8>(o, 2, pf0, bb1);
        }
        const bool adv = pfrow < rlast;
        pfrow = min(pfrow + 1, rlast);
        pf0 += adv ? sstride0 : 0;
        pf1 += adv ? sstride1 : 0;
        asm("" : "+s"(pf0), "+s"(pf1));
    };
    /* current descriptor set (rows 0..63 of the strip, then 64..127) */
    int cvpl = vpl[0];
    uint32_t cvf01 = vf01[0], cvf23 = vf23[0];

    CwRaw buf[D];
#pragma unroll
    for (int k = 0; k < D; k++) {
        if (OPT)
            load_next(buf[k]);
        else
            load_row(buf[k], min(r + k, rlast));
    }
    for (; r <= rlast; r += D) {
#pragma unroll
        for (int k = 0; k < D; k++) {
            const int rr = r + k;
            /* the prefetch is unconditional (its row index is clamped): a load under the branch would be
             * copied into place after it, and that copy would wait for the load just issued */
            const CwRaw cur = buf[k];
            if (OPT)
                load_next(buf[k]);
            else
                load_row(buf[k], min(rr + D, rlast));
            if (rr <= rlast) {
                hpass(cur, Pw[k % 3], Pw[(k + 2) % 3]);
                while (yy < ny && need <= rr) {
                    if (OPT) {
                        const int ll = yy & 63;
                        emit(dr0, dr1, __builtin_amdgcn_readlane(cvf01, ll), __builtin_amdgcn_readlane(cvf23, ll),
                             Pw[(k + 1) % 3], Pw[k % 3]);
                        dr0 += dstride0;
                        dr1 += dstride1;
                        asm("" : "+s"(dr0), "+s"(dr1));
                        yy++;
                        if (yy == 64) { /* a real (uniform) branch, taken once per strip: not three selects per row */
                            asm volatile("; second descriptor set");
                            cvpl = vpl[1];
                            cvf01 = vf01[1];
                            cvf23 = vf23[1];
                        }
                        if (yy < ny)
                            need = __builtin_amdgcn_readlane(cvpl, yy & 63) + 3;
                    } else {
                        const bool hi = yy >= 64;
                        const int ll = yy & 63;
                        const uint32_t f01 = __builtin_amdgcn_readlane(hi ? vf01[1] : vf01[0], ll);
                        const uint32_t f23 = __builtin_amdgcn_readlane(hi ? vf23[1] : vf23[0], ll);
                        emit(d0 + (ptrdiff_t)(y0 + yy) * dstride0, d1 + (ptrdiff_t)(y0 + yy) * dstride1, f01, f23,
                             Pw[(k + 1) % 3], Pw[k % 3]);
                        yy++;
                        if (yy < ny)
                            need = __builtin_amdgcn_readlane(yy >= 64 ? vpl[1] : vpl[0], yy & 63) + 3;
                    }
                }
            }
        }
    }
}

template <int LK, int D, bool PLAIN, bool OPT, bool DUP>
__device__ __forceinline__ void cw_kernel_body(const FFHipCwArgs &A)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave; /* < 2^31, checked by the launcher */
    if (gw >= (uint32_t)A.units_per_frame * (uint32_t)A.nframes)
        return;
    const int f = (int)(gw / (uint32_t)A.units_per_frame);
    const int u = (int)(gw - (uint32_t)f * (uint32_t)A.units_per_frame);
    int j = 0;
    if (A.njobs > 1 && u >= A.job[1].unit_begin) j = 1;
    if (A.njobs > 2 && u >= A.job[2].unit_begin) j = 2;
    const FFHipCwJob &J = A.job[j];
    const int local = u - J.unit_begin;
    const int strip = local / J.ncb, cb = local - strip * J.ncb;
    if (J.kind == 2)
        cw_unit<2, D, PLAIN, OPT, DUP>(J, f, strip, cb, lane);
    else if (J.kind == 3)
        cw_unit<3, D, PLAIN, OPT, DUP>(J, f, strip, cb, lane);
    else if (J.kind == 4)
        cw_unit<4, D, PLAIN, OPT, DUP>(J, f, strip, cb, lane);
    else
        cw_unit<LK, D, PLAIN, OPT, DUP>(J, f, strip, cb, lane);
}

template <int LK, int D, bool PLAIN, bool OPT, bool DUP = false>
__global__ __launch_bounds__(256) void k_sws_colwalk(FFHipCwArgs A)
{
    cw_kernel_body<LK, D, PLAIN, OPT, DUP>(A);
}


/* ================================================================================================== */
/*
 * k_sws_colwalk_rgb — the column walker with packed rgb24/bgr24 output (yuv2rgb_X_c_template,
 * libswscale/output.c:1797-1850 via yuv2packedX, vscale.c:126-170) for 4-tap vertical banks.
 *
 * A lane owns 8 output pixels (two luma column groups) and the 4 chroma columns under them.  The luma walk
 * is the statically indexed, D-deep prefetched one of cw_unit; the chroma planes advance far less often
 * (chrDstH == dstH: 4x vertically for 4:2:0 at 2x), so their ring of vertical pairs simply shifts through
 * register moves and their source rows are fetched one advance ahead.  Per output row: 16 luma + 16 chroma
 * vertical dots, the closed-form yuv2rgb of sws_yuv2rgb.hip, v_ashr_pk_u8_i32 pairs, three 8-byte stores
 * (24 contiguous bytes per lane, 1536 per wave).  Nothing but source rows and RGB rows touches memory.
 */
template <int SH>
__device__ __forceinline__ uint32_t cw_pk_sh(int a, int b)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "n"(SH));
    return r;
}

/* horizontal descriptor of one 4-column group: byte selectors + coefficient pairs, window base (dword aligned) */
__device__ __forceinline__ int cw_hdesc(uint32_t (&sel)[8], uint32_t (&cf)[8], const int16_t *hf, const int32_t *hp, int X0,
                                        int n, int srcW)
{
    int p[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int xi = min(X0 + i, n - 1);
        p[i] = hp[xi];
        const uint2 c = *reinterpret_cast<const uint2 *>(hf + (size_t)xi * 4);
        cf[2 * i] = c.x;
        cf[2 * i + 1] = c.y;
    }
    const int base = cw_span_base(p, srcW);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t o = (uint32_t)(p[i] - base);
        sel[2 * i]     = 0x0c000c00u | o | ((o + 1) << 16);
        sel[2 * i + 1] = 0x0c000c00u | (o + 2) | ((o + 3) << 16);
    }
    return base;
}

/* 4 horizontal samples of one group from its 8 raw bytes -> (h[r-1], h[r]) pairs, int16-saturated (nowrap banks) */
__device__ __forceinline__ void cw_hgroup(uint32_t (&Pnew)[4], int (&hprev)[4], uint32_t d0, uint32_t d1, const uint32_t (&sel)[8],
                                          const uint32_t (&cf)[8])
{
    uint32_t a[4], b[4];
    int acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = __builtin_amdgcn_perm(d1, d0, sel[2 * i]);
        b[i] = __builtin_amdgcn_perm(d1, d0, sel[2 * i + 1]);
    }
    cw_hdots4(acc, a, b, cf);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int h = acc[i] >> 7;
        Pnew[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[i], h));
        hprev[i] = h;
    }
}

__device__ __forceinline__ void cw_wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* LUT (round 4): the chroma part of the closed form — 23 VALU instructions per chroma sample and output row, a third of the kernel — is
 * a function of one byte per term: r(V), b(U), g = gu(U) + gv(V) (the products distribute over the sum exactly in wrapping int32
 * arithmetic).  The workgroup builds the four 256-entry tables in 4 KB of LDS once (one entry per thread) and a sample costs two
 * 8-byte LDS reads and an add. */
template <bool SIL, int LAY, int D, bool TR, bool LUT = true> /* LAY: 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra (alpha = 255) */
__global__ __launch_bounds__(256) void k_sws_colwalk_rgb(FFHipCwRgbArgs A)
{
    __shared__ uint32_t tiles[4][384];
    __shared__ uint2 lutU[LUT ? 256 : 1], lutV[LUT ? 256 : 1]; /* { b(U), gu(U) }, { r(V), gv(V) } */
    if (LUT) {
        const int t = (int)threadIdx.x;
        const FFHipYuv2RgbK Kt = A.k;
        lutU[t] = make_uint2((uint32_t)(__mul24(Kt.off_b + (__mul24(t, Kt.cbu) >> 16), Kt.cy) + Kt.kb),
                             (uint32_t)(__mul24(Kt.off_g + (__mul24(t, Kt.cgu) >> 16), Kt.cy) + Kt.kb));
        lutV[t] = make_uint2((uint32_t)(__mul24(Kt.off_r + (__mul24(t, Kt.crv) >> 16), Kt.cy) + Kt.kb),
                             (uint32_t)__mul24(__mul24(t, Kt.cgv) >> 16, Kt.cy));
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t upf = (uint32_t)A.ncb * (uint32_t)A.nstrips;
    if (gw >= upf * (uint32_t)A.nframes)
        return;
    const int f = (int)(gw / upf);
    const int u = (int)(gw - (uint32_t)f * upf);
    const int strip = u / A.ncb, cb = u - strip * A.ncb;
    const int y0 = strip * A.strip_rows;
    const int y1 = min(y0 + A.strip_rows, A.dstH); /* <= 64 rows: one register per vertical descriptor */

    /* ---- horizontal descriptors: two luma groups, one chroma group ---- */
    const int X0 = (cb * 64 + lane) * 8;
    uint32_t lsel[2][8], lcf[2][8], csel[8], ccf[8];
    const uint32_t lb0 = (uint32_t)cw_hdesc(lsel[0], lcf[0], A.hlf, A.hlp, X0, A.dstW, A.srcW);
    const uint32_t lb1 = (uint32_t)cw_hdesc(lsel[1], lcf[1], A.hlf, A.hlp, X0 + 4, A.dstW, A.srcW);
    const uint32_t cbase = (uint32_t)cw_hdesc(csel, ccf, A.hcf, A.hcp, X0 >> 1, A.dstW >> 1, A.chrSrcW);
    const bool act = X0 < A.dstW; /* dstW % 8 == 0 (host-checked): a lane is all or nothing */
    const uint32_t cbb = SIL ? 2 * cbase : cbase;
    const uint32_t sel_u = A.src_swap ? 0x07050301u : 0x06040200u, sel_v = A.src_swap ? 0x06040200u : 0x07050301u;

    const uint8_t *sy = A.src[0] + (size_t)f * A.sfp[0];
    const uint8_t *su = A.src[1] + (size_t)f * A.sfp[1];
    const uint8_t *sv = SIL ? su : A.src[2] + (size_t)f * A.sfp[2];
    const ptrdiff_t ystride = A.sstride[0], ustride = A.sstride[1], vstride = SIL ? A.sstride[1] : A.sstride[2];
    const ptrdiff_t dstride = A.dstride;

    /* ---- vertical descriptors (wave-uniform per output row, read with v_readlane) ---- */
    int vpl, cpl;
    uint32_t lf01, lf23, cf01, cf23;
    {
        const int y = min(y0 + lane, A.dstH - 1);
        vpl = A.vlp[y];
        cpl = A.vcp[y];
        const uint2 a = *reinterpret_cast<const uint2 *>(A.vlf + (size_t)y * 4);
        const uint2 b = *reinterpret_cast<const uint2 *>(A.vcf + (size_t)y * 4);
        lf01 = a.x; lf23 = a.y; cf01 = b.x; cf23 = b.y;
    }
    const FFHipYuv2RgbK K = A.k;

    uint32_t Pw[3][2][4], Ca[2][4], Cm[2][4], Cb[2][4];
    int hprev[2][4], cprev[2][4];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            Pw[0][g][i] = Pw[1][g][i] = Pw[2][g][i] = 0;
            Ca[g][i] = Cm[g][i] = Cb[g][i] = 0;
            hprev[g][i] = cprev[g][i] = 0;
        }
    int kround = A.vround;
    asm volatile("" : "+v"(kround));

    const int ny = y1 - y0;
    int yy = 0;
    int need = __builtin_amdgcn_readlane(vpl, 0) + 3;
    int cneed = __builtin_amdgcn_readlane(cpl, 0) + 3;
    const int rlast = __builtin_amdgcn_readfirstlane(A.vlp[y1 - 1]) + 3;
    const int crlast = __builtin_amdgcn_readfirstlane(A.vcp[y1 - 1]) + 3;
    int r = need - 3;
    int ccur = cneed - 4; /* newest chroma row in the ring */

    /* running row pointers (SGPRs) */
    int pfrow = r, cpfrow = ccur + 1;
    const uint8_t *pfy = sy + (ptrdiff_t)r * ystride;
    const uint8_t *pfu = su + (ptrdiff_t)cpfrow * ustride, *pfv = sv + (ptrdiff_t)cpfrow * vstride;
    uint8_t *dr = A.dst + (size_t)f * A.dfp + (ptrdiff_t)y0 * dstride;
    asm("" : "+s"(pfy), "+s"(pfu), "+s"(pfv), "+s"(dr));
    const uint32_t dcol = 3u * (uint32_t)X0;
    uint32_t *tile = tiles[wave];
    const uint32_t tcol = 1536u * (uint32_t)cb + 8u * (uint32_t)lane; /* transposed: this lane's bytes of each 512-byte run */
    const int nbytes = 3 * min(A.dstW - cb * 512, 512);                /* valid bytes of this wave's row segment (% 24 == 0) */

    auto load_luma = [&](CwRaw &o) {
        cw_load<8>(o, 0, pfy, lb0);
        cw_load<8>(o, 2, pfy, lb1);
        const bool adv = pfrow < rlast;
        pfrow = min(pfrow + 1, rlast);
        pfy += adv ? ystride : 0;
        asm("" : "+s"(pfy));
    };
    auto load_chroma = [&](CwRaw &o) {
        if (SIL) {
            cw_load<16>(o, 0, pfu, cbb);
        } else {
            cw_load<8>(o, 0, pfu, cbb);
            cw_load<8>(o, 2, pfv, cbb);
        }
        const bool adv = cpfrow < crlast;
        cpfrow = min(cpfrow + 1, crlast);
        pfu += adv ? ustride : 0;
        pfv += adv ? vstride : 0;
        asm("" : "+s"(pfu), "+s"(pfv));
    };

    CwRaw cnext;
    load_chroma(cnext);
    CwRaw buf[D];
#pragma unroll
    for (int k = 0; k < D; k++)
        load_luma(buf[k]);

    auto chroma_advance = [&]() {
        const CwRaw cur = cnext;
        load_chroma(cnext);
        uint32_t u0, u1, v0, v1;
        if (SIL) {
            u0 = __builtin_amdgcn_perm(cur.q[1], cur.q[0], sel_u);
            u1 = __builtin_amdgcn_perm(cur.q[3], cur.q[2], sel_u);
            v0 = __builtin_amdgcn_perm(cur.q[1], cur.q[0], sel_v);
            v1 = __builtin_amdgcn_perm(cur.q[3], cur.q[2], sel_v);
        } else {
            u0 = cur.q[0]; u1 = cur.q[1]; v0 = cur.q[2]; v1 = cur.q[3];
        }
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Ca[g][i] = Cm[g][i];
                Cm[g][i] = Cb[g][i];
            }
        cw_hgroup(Cb[0], cprev[0], u0, u1, csel, ccf);
        cw_hgroup(Cb[1], cprev[1], v0, v1, csel, ccf);
        ccur++;
    };

    auto emit = [&](const uint32_t (&La)[2][4], const uint32_t (&Lb)[2][4], int ll) {
        const uint32_t f01 = __builtin_amdgcn_readlane(lf01, ll), f23 = __builtin_amdgcn_readlane(lf23, ll);
        const uint32_t g01 = __builtin_amdgcn_readlane(cf01, ll), g23 = __builtin_amdgcn_readlane(cf23, ll);
        int Yv[2][4], Uv[4], Vv[4];
        cw_vdots4(Yv[0], La[0], Lb[0], f01, f23, kround);
        cw_vdots4(Yv[1], La[1], Lb[1], f01, f23, kround);
        cw_vdots4(Uv, Ca[0], Cb[0], g01, g23, kround);
        cw_vdots4(Vv, Ca[1], Cb[1], g01, g23, kround);
        constexpr bool BGR = LAY == 1;
        uint32_t w[LAY < 2 ? 6 : 8];
        int alpha = 255 << 16; /* v_ashr_pk_u8_i32 ..., 16 of it is the alpha byte */
        asm("" : "+v"(alpha));
#pragma unroll
        for (int h = 0; h < 2; h++) {
            int val[12];
#pragma unroll
            for (int mm = 0; mm < 2; mm++) {
                const int m = 2 * h + mm;
                const int Uc = min(max(Uv[m] >> 19, 0), 255), Vc = min(max(Vv[m] >> 19, 0), 255);
                int br, bb, bg;
                if (LUT) {
                    const uint2 tu = lutU[Uc], tv = lutV[Vc];
                    br = (int)tv.x;
                    bb = (int)tu.x;
                    bg = (int)(tu.y + tv.y);
                } else {
                    br = __mul24(K.off_r + (__mul24(Vc, K.crv) >> 16), K.cy) + K.kb;
                    bb = __mul24(K.off_b + (__mul24(Uc, K.cbu) >> 16), K.cy) + K.kb;
                    bg = __mul24(K.off_g + (__mul24(Uc, K.cgu) >> 16) + (__mul24(Vc, K.cgv) >> 16), K.cy) + K.kb;
                }
                const int c0 = BGR ? bb : br, c2 = BGR ? br : bb;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int p = 2 * mm + e;
                    const int yc = __mul24(Yv[h][p] >> 19, K.cy);
                    val[3 * p] = yc + c0;
                    val[3 * p + 1] = yc + bg;
                    val[3 * p + 2] = yc + c2;
                }
            }
            if (LAY < 2) {
#pragma unroll
                for (int d = 0; d < 3; d++)
                    w[3 * h + d] = __builtin_amdgcn_perm(cw_pk_sh<16>(val[4 * d + 2], val[4 * d + 3]),
                                                         cw_pk_sh<16>(val[4 * d], val[4 * d + 1]), 0x05040100);
            } else {
                /* a pixel is a dword: (val[3p], val[3p+1], val[3p+2]) = (R, G, B) pre-shift sums */
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int R = val[3 * p], G = val[3 * p + 1], B = val[3 * p + 2];
                    uint32_t lo, hi;
                    if (LAY == 2)      { lo = cw_pk_sh<16>(alpha, R); hi = cw_pk_sh<16>(G, B); }
                    else if (LAY == 3) { lo = cw_pk_sh<16>(R, G); hi = cw_pk_sh<16>(B, alpha); }
                    else if (LAY == 4) { lo = cw_pk_sh<16>(alpha, B); hi = cw_pk_sh<16>(G, R); }
                    else               { lo = cw_pk_sh<16>(B, G); hi = cw_pk_sh<16>(R, alpha); }
                    w[(LAY < 2 ? 0 : 4 * h) + p] = __builtin_amdgcn_perm(hi, lo, 0x05040100);
                }
            }
        }
        if (LAY >= 2) {
            /* 32 contiguous bytes per lane, 2 KiB per wave and row: two aligned 16-byte stores */
            if (act) {
                cw_gptr d = (cw_gptr)dr + cw_opaque(4u * (uint32_t)X0);
                cw_u4 s0, s1;
                s0.x = w[0]; s0.y = w[1]; s0.z = w[2]; s0.w = w[3];
                s1.x = w[4 % (LAY < 2 ? 6 : 8)]; s1.y = w[5 % (LAY < 2 ? 6 : 8)]; s1.z = w[6 % (LAY < 2 ? 6 : 8)]; s1.w = w[7 % (LAY < 2 ? 6 : 8)];
                typedef cw_u4 __attribute__((address_space(1))) *cw_g4;
                if (A.nts) { /* written once, not read back: non-temporal (as sws_yuv2rgb.hip, round 5) */
                    __builtin_nontemporal_store(s0, (cw_g4)d);
                    __builtin_nontemporal_store(s1, (cw_g4)(d + 16));
                } else {
                    *(cw_g4)d = s0;
                    *(cw_g4)(d + 16) = s1;
                }
            }
            dr += dstride;
            asm("" : "+s"(dr));
            return;
        }
        if (TR) {
            /* transpose through the wave's 1.5 KiB of LDS: each store instruction then covers 512 contiguous bytes
             * of the row instead of 8 bytes in every 24 */
            uint32_t *t = tile + lane * 6;
            *reinterpret_cast<uint2 *>(t) = make_uint2(w[0], w[1]);
            *reinterpret_cast<uint2 *>(t + 2) = make_uint2(w[2], w[3]);
            *reinterpret_cast<uint2 *>(t + 4) = make_uint2(w[4], w[5]);
            cw_wave_sync_lds();
            cw_gptr d = (cw_gptr)dr + cw_opaque(tcol);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const uint2 v = *reinterpret_cast<const uint2 *>(tile + i * 128 + lane * 2);
                cw_u2 s;
                s.x = v.x; s.y = v.y;
                if (i * 512 + lane * 8 < nbytes) {
                    if (A.nts) __builtin_nontemporal_store(s, (cw_g2)(d + i * 512));
                    else *(cw_g2)(d + i * 512) = s;
                }
            }
            cw_wave_sync_lds();
        } else if (act) {
            cw_gptr d = (cw_gptr)dr + cw_opaque(dcol);
            cw_u2 s;
            s.x = w[0]; s.y = w[1]; *(cw_g2)d = s;
            s.x = w[2]; s.y = w[3]; *(cw_g2)(d + 8) = s;
            s.x = w[4]; s.y = w[5]; *(cw_g2)(d + 16) = s;
        }
        dr += dstride;
        asm("" : "+s"(dr));
    };

    for (; r <= rlast; r += D) {
#pragma unroll
        for (int k = 0; k < D; k++) {
            const int rr = r + k;
            const CwRaw cur = buf[k];
            load_luma(buf[k]);
            if (rr <= rlast) {
                cw_hgroup(Pw[k % 3][0], hprev[0], cur.q[0], cur.q[1], lsel[0], lcf[0]);
                cw_hgroup(Pw[k % 3][1], hprev[1], cur.q[2], cur.q[3], lsel[1], lcf[1]);
                while (yy < ny && need <= rr) {
                    while (ccur < cneed)
                        chroma_advance();
                    emit(Pw[(k + 1) % 3], Pw[k % 3], yy);
                    yy++;
                    if (yy < ny) {
                        need = __builtin_amdgcn_readlane(vpl, yy) + 3;
                        cneed = __builtin_amdgcn_readlane(cpl, yy) + 3;
                    }
                }
            }
        }
    }
}

/* ================================================================================================== */
/*
 * k_sws_mfma — the same scaler with the HORIZONTAL pass on the matrix cores.
 *
 * Why: the column walker is bound by integer VALU issue (~7 half-rate instructions per output sample, PMC:
 * VALU 75-94 % busy), and more than half of that is the horizontal pass (byte unpack + dots + pair
 * packing).  H[r][x] = sum_k src[r][k] * Bt[k][x] is a banded matrix product; on i8 MFMA the band's zeros
 * cost nothing that matters (v_mfma_i32_32x32x32_i8: 32K MACs in 32 cycles on a pipe of its own), the byte
 * unpack disappears (source bytes ARE the A operand), and NV12 de-interleaving is just which k have
 * non-zero coefficients.  Exactness: int32 accumulation of exact integer products.
 *   - 14-bit coefficients f = 256*fh + fl (fl signed low byte): two MFMAs chained through
 *     C2 = (D_hi << 8) + bias;
 *   - source bytes are unsigned, i8 MFMA is signed: A = src ^ 0x80 (= src - 128), bias = 128 * sum(f)
 *     per output column restores the difference.
 * A workgroup owns 16 tiles of 32 horizontal samples (512 luma columns, or 256 chroma columns x {U,V}) of
 * a strip of output rows and alternates two phases over chunks of 28 source rows:
 *   H  every wave runs its 4 tiles: 32 A rows = source rows c0..c0+15 (lanes 0-31's results) and
 *      c0+15..c0+30 (lanes 32-63's) — the D layout (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) then puts
 *      16 CONSECUTIVE source rows of one column in each lane, so >>7, int16 saturation and the vertical
 *      pairs (h[r-1], h[r]) are made in registers (v_cvt_pk_i16_i32) and written to LDS as pairs[30][512];
 *   V  the column walker's vertical pass unchanged (two v_dot2_i32_i16 per sample on pairs, v_ashr_pk_u8_i32,
 *      8-byte stores), reading its pairs from LDS; the chunk's output rows are split over the 4 waves.
 * The tile records (B_hi, B_lo in MFMA lane order, bias, window base) are built on the host from the
 * caller's filter bank (ffhip_mf_build_tiles) and stay resident in registers for the whole strip.
 */
typedef int mf_i4 __attribute__((ext_vector_type(4)));
typedef int mf_i16 __attribute__((ext_vector_type(16)));
typedef const mf_i4 __attribute__((address_space(1))) *mf_gc4;

#define MF_PR 30      /* pair rows per chunk              */
#define MF_ADV 28     /* window positions (source rows) a chunk completes */
#define MF_W 512      /* dwords per pair row              */
#define MF_REC 2320   /* bytes per tile record            */

template <int PAIR>
__device__ __forceinline__ void mf_block(const FFHipMfJob &J, int f, int strip, int cb, uint32_t *lds)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, g = lane >> 5;

    /* ---- this wave's four tiles ---- */
    mf_i4 Bhi[4], Blo[4];
    int bias[4], kbase[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int t = min(cb * 16 + wave * 4 + q, J.ntiles - 1);
        const uint8_t *rec = J.tiles + (size_t)t * MF_REC;
        Bhi[q] = reinterpret_cast<const mf_i4 *>(rec)[lane];
        Blo[q] = reinterpret_cast<const mf_i4 *>(rec + 1024)[lane];
        bias[q] = reinterpret_cast<const int *>(rec + 2048)[lane];
        kbase[q] = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(rec + 2304));
    }
    /* A row i = lane & 31 carries source row c0 + 15*half(i) + t(i) (see the header comment) */
    const int arow = 15 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);

    /* ---- vertical-pass lane constants ---- */
    const int X0 = PAIR ? cb * 256 + 4 * lane : cb * 512 + 8 * lane; /* first column (per channel) */
    const bool act = X0 < J.dstW;
    const uint32_t lidx = PAIR ? (uint32_t)((lane >> 2) * 32 + 4 * (lane & 3)) : (uint32_t)(8 * lane);
    const uint32_t dcol = PAIR ? 2u * (uint32_t)X0 : (uint32_t)X0;
    const uint32_t sel_uv = J.dst_swap ? 0x04050001u : 0x05040100u;
    int kround = 64 << 12;
    asm volatile("" : "+v"(kround));

    const uint8_t *src = J.src + (size_t)f * J.sfp;
    uint8_t *dst = J.dst + (size_t)f * J.dfp;
    const int y0 = strip * J.strip_rows, y1 = min(y0 + J.strip_rows, J.dstH);
    int c0 = __builtin_amdgcn_readfirstlane(J.vp[y0]);
    /* Descriptors run one chunk ahead of their use so that their L2 round trips hide behind an H phase:
     * ys_* = first/last output row of the chunk, (vpl, cf) = window start and coefficient pairs of this
     * wave's rows (row w0 + lane), read later with v_readlane. */
    int ysb_raw = J.ys[min(c0 + MF_ADV, J.srcH)];
    int ya = max(__builtin_amdgcn_readfirstlane(J.ys[c0]), y0);
    int yb = min(__builtin_amdgcn_readfirstlane(ysb_raw), y1);
    int w0, w1, vpl;
    uint2 cf;
    auto load_desc = [&](int from) {
        const int yl = min(from + lane, J.dstH - 1);
        vpl = J.vp[yl];
        cf = *reinterpret_cast<const uint2 *>(J.vf + (size_t)yl * 4);
    };
    auto split_rows = [&]() {
        const int per = (yb - ya + 3) >> 2;
        w0 = min(ya + wave * per, yb);
        w1 = min(w0 + per, yb);
    };
    split_rows();
    load_desc(w0);

    /* the A operands (source bytes) of a chunk are fetched one chunk ahead, during the previous V phase */
    mf_i4 anext[4];
    auto load_a = [&](int cbase) {
        const int srow = min(cbase + arow, J.srcH - 1);
        const uint8_t *rowp = src + (ptrdiff_t)srow * J.sstride + 16 * g;
#pragma unroll
        for (int q = 0; q < 4; q++)
            anext[q] = *(mf_gc4)((cw_gcptr)rowp + (uint32_t)kbase[q]);
    };
    load_a(c0);

    for (;;) {
        const int ysn_raw = J.ys[min(c0 + 2 * MF_ADV, J.srcH)]; /* next chunk's end, needed after this one */

        /* ---------------- H phase ---------------- */
        {
            mf_i4 acur[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                acur[q] = anext[q] ^ (mf_i4){ (int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080 };
            load_a(c0 + MF_ADV); /* in flight across the MFMAs and the whole V phase */
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const mf_i4 a = acur[q];
                mf_i16 acc = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Bhi[q], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)bias[q]);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Blo[q], acc, 0, 0, 0);
                uint32_t *col = lds + (wave * 4 + q) * 32 + j + (15 * g) * MF_W;
#pragma unroll
                for (int t = 1; t < 16; t++)
                    col[(t - 1) * MF_W] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(acc[t - 1] >> 7, acc[t] >> 7));
            }
        }
        __syncthreads();

        /* ---------------- V phase: this wave's rows [w0, w1) of the chunk ---------------- */
        for (int yb0 = w0; yb0 < w1; yb0 += 64) {
            if (yb0 != w0)
                load_desc(yb0); /* more than 64 rows per wave and chunk: only for > 9x vertical up-scaling */
            const int cnt = min(64, w1 - yb0);
            /* the pair rows of output row yy+1 are read while row yy is computed */
            uint4 na0, na1, nb0, nb1;
            auto read_pairs = [&](int yy) {
                const int p = __builtin_amdgcn_readlane(vpl, yy) - c0; /* pair row holding (h[p], h[p+1]) */
                const uint32_t *pa = lds + p * MF_W + lidx, *pb = pa + 2 * MF_W;
                na0 = *reinterpret_cast<const uint4 *>(pa);
                nb0 = *reinterpret_cast<const uint4 *>(pb);
                na1 = *reinterpret_cast<const uint4 *>(pa + (PAIR ? 16 : 4));
                nb1 = *reinterpret_cast<const uint4 *>(pb + (PAIR ? 16 : 4));
            };
            read_pairs(0);
            for (int yy = 0; yy < cnt; yy++) {
                const uint32_t f01 = __builtin_amdgcn_readlane(cf.x, yy), f23 = __builtin_amdgcn_readlane(cf.y, yy);
                const uint4 a0 = na0, a1 = na1, b0 = nb0, b1 = nb1;
                read_pairs(min(yy + 1, cnt - 1));
                uint32_t Pa[2][4], Pb[2][4];
                Pa[0][0] = a0.x; Pa[0][1] = a0.y; Pa[0][2] = a0.z; Pa[0][3] = a0.w;
                Pa[1][0] = a1.x; Pa[1][1] = a1.y; Pa[1][2] = a1.z; Pa[1][3] = a1.w;
                Pb[0][0] = b0.x; Pb[0][1] = b0.y; Pb[0][2] = b0.z; Pb[0][3] = b0.w;
                Pb[1][0] = b1.x; Pb[1][1] = b1.y; Pb[1][2] = b1.z; Pb[1][3] = b1.w;
                int v[2][4];
                cw_vdots4(v[0], Pa[0], Pb[0], f01, f23, kround);
                cw_vdots4(v[1], Pa[1], Pb[1], f01, f23, kround);
                cw_u2 w;
                if (PAIR) { /* U0 V0 U1 V1 ... */
                    w.x = __builtin_amdgcn_perm(cw_pk_u8<false>(v[0][1], v[1][1]), cw_pk_u8<false>(v[0][0], v[1][0]), sel_uv);
                    w.y = __builtin_amdgcn_perm(cw_pk_u8<false>(v[0][3], v[1][3]), cw_pk_u8<false>(v[0][2], v[1][2]), sel_uv);
                } else {
                    w.x = __builtin_amdgcn_perm(cw_pk_u8<false>(v[0][2], v[0][3]), cw_pk_u8<false>(v[0][0], v[0][1]), 0x05040100);
                    w.y = __builtin_amdgcn_perm(cw_pk_u8<false>(v[1][2], v[1][3]), cw_pk_u8<false>(v[1][0], v[1][1]), 0x05040100);
                }
                uint8_t *drow = dst + (ptrdiff_t)(yb0 + yy) * J.dstride;
                asm("" : "+s"(drow));
                cw_gptr d = (cw_gptr)drow + cw_opaque(dcol);
                if (X0 + (PAIR ? 4 : 8) <= J.dstW)
                    *(cw_g2)d = w;
                else if (act && !PAIR)
                    *(cw_g1)d = w.x;
            }
        }
        __syncthreads();
        if (yb >= y1)
            break;
        c0 += MF_ADV;
        ya = max(__builtin_amdgcn_readfirstlane(ysb_raw), y0);
        yb = min(__builtin_amdgcn_readfirstlane(ysn_raw), y1);
        ysb_raw = ysn_raw;
        split_rows();
        load_desc(w0);
    }
}

__global__ __launch_bounds__(256) void k_sws_mfma(FFHipMfArgs A)
{
    extern __shared__ __align__(16) uint32_t mf_lds[];
    const uint32_t u0 = blockIdx.x;
    if (u0 >= (uint32_t)A.units_per_frame * (uint32_t)A.nframes)
        return;
    const int f = (int)(u0 / (uint32_t)A.units_per_frame);
    const int u = (int)(u0 - (uint32_t)f * (uint32_t)A.units_per_frame);
    int jj = 0;
    if (A.njobs > 1 && u >= A.job[1].unit_begin) jj = 1;
    if (A.njobs > 2 && u >= A.job[2].unit_begin) jj = 2;
    const FFHipMfJob &J = A.job[jj];
    const int local = u - J.unit_begin;
    const int strip = local / J.ncb, cb = local - strip * J.ncb;
    if (J.pair)
        mf_block<1>(J, f, strip, cb, mf_lds);
    else
        mf_block<0>(J, f, strip, cb, mf_lds);
}

/*
 * Host: tile records of one horizontal bank in MFMA operand order.  Returns the number of tiles, or -1 when
 * the bank does not fit (a tile's taps outside its 32-byte window, or a coefficient whose high byte overflows).
 * pair: the source is a byte-interleaved U/V plane; tile columns 0-15 are U, 16-31 are V samples.
 */
int ffhip_mf_build_tiles(std::vector<uint8_t> *out, const int16_t *hf, const int32_t *hp, int n, int srcW, int pair, int src_swap)
{
    const int cpt = pair ? 16 : 32;
    const int ntiles = cdiv(n, cpt);
    const int rowbytes = pair ? 2 * srcW : srcW;
    if (rowbytes < 32 || (rowbytes & 3) || (n & 3))
        return -1;
    out->assign((size_t)ntiles * MF_REC, 0);
    for (int t = 0; t < ntiles; t++) {
        uint8_t *rec = out->data() + (size_t)t * MF_REC;
        int8_t *bhi = reinterpret_cast<int8_t *>(rec), *blo = reinterpret_cast<int8_t *>(rec + 1024);
        int32_t *bias = reinterpret_cast<int32_t *>(rec + 2048);
        const int c_lo = t * cpt, c_hi = c_lo + cpt < n ? c_lo + cpt : n;
        int first = 1 << 30, last = -1;
        for (int ch = 0; ch < (pair ? 2 : 1); ch++) {
            const int off = pair ? (ch ^ (src_swap ? 1 : 0)) : 0;
            for (int c = c_lo; c < c_hi; c++) {
                const int b0 = pair ? 2 * hp[c] + off : hp[c], b1 = pair ? 2 * (hp[c] + 3) + off : hp[c] + 3;
                if (b0 < first) first = b0;
                if (b1 > last) last = b1;
            }
        }
        int kb = first & ~3;
        if (kb + 32 > rowbytes)
            kb = rowbytes - 32;
        if (first < kb || last >= kb + 32 || kb < 0)
            return -1;
        *reinterpret_cast<int32_t *>(rec + 2304) = kb;
        for (int l = 0; l < 64; l++) {
            const int jx = l & 31, g = l >> 5;
            const int ch = pair ? jx >> 4 : 0, c = c_lo + (pair ? (jx & 15) : jx);
            if (c >= c_hi)
                continue;
            const int off = pair ? (ch ^ (src_swap ? 1 : 0)) : 0;
            if (g == 0) {
                int sum = 0;
                for (int k = 0; k < 4; k++)
                    sum += hf[(size_t)c * 4 + k];
                bias[l] = bias[l + 32] = 128 * sum;
            }
            for (int s = 0; s < 16; s++) {
                const int byte = kb + 16 * g + s;
                int tap;
                if (pair) {
                    if (((byte - off) & 1) || byte < off)
                        continue;
                    tap = (byte - off) / 2 - hp[c];
                } else {
                    tap = byte - hp[c];
                }
                if (tap < 0 || tap > 3)
                    continue;
                const int fv = hf[(size_t)c * 4 + tap];
                const int fl = ((fv + 128) & 255) - 128, fh = (fv - fl) >> 8;
                if (fh < -128 || fh > 127)
                    return -1;
                bhi[l * 16 + s] = (int8_t)fh;
                blo[l * 16 + s] = (int8_t)fl;
            }
        }
    }
    return ntiles;
}

int ffhip_launch_mfma(FFHipMfArgs &A, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    int u = 0;
    for (int i = 0; i < A.njobs; i++) {
        A.job[i].unit_begin = u;
        u += A.job[i].ncb * A.job[i].nstrips;
    }
    A.units_per_frame = u;
    const long long blocks = (long long)u * A.nframes;
    if (blocks >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld workgroups)", blocks);
        return FFHIP_EINVAL;
    }
    hipLaunchKernelGGL(k_sws_mfma, dim3((unsigned)blocks), dim3(256), MF_PR * MF_W * 4, stream, A);
    LAUNCH_CHECK();
    return 0;
}

/* ---- host side ---------------------------------------------------------------------------------- */
/* can a bank pair run on the column walker?  hpos/vpos are host copies. */
int ffhip_cw_bank_ok(const int32_t *hpos, int hsize, int hn, int srcW, const int32_t *vpos, int vsize, int vn, int srcH)
{
    if (hsize != 4 || vsize != 4 || hn <= 0 || (hn & 3) || vn <= 0 || srcW < 8 || srcH < 4)
        return 0;
    for (int x0 = 0; x0 < hn; x0 += 4) {
        int lo = hpos[x0], hi = hpos[x0];
        for (int i = 1; i < 4; i++) {
            const int p = hpos[x0 + i < hn ? x0 + i : hn - 1];
            if (p < lo) lo = p;
            if (p > hi) hi = p;
        }
        if (lo < 0 || hi + 4 > srcW || hi - lo > 4) /* the group's windows must fit one 8-byte span (cw_span_base) */
            return 0;
    }
    for (int y = 0; y < vn; y++) {
        if (vpos[y] < 0 || vpos[y] + 4 > srcH || (y && vpos[y] < vpos[y - 1]))
            return 0;
    }
    return 1;
}

/* 1 when columns 1 and 2 of every 4-column group start at the same source sample with any coefficients (exact 2x
 * up-scaling): the DUP variant then unpacks that window once */
int ffhip_cw_bank_dup12(const int32_t *hpos, int hn)
{
    for (int x0 = 0; x0 + 3 < hn; x0 += 4)
        if (hpos[x0 + 1] != hpos[x0 + 2])
            return 0;
    return (hn & 3) == 0;
}

int ffhip_cw_bank_nowrap(const int16_t *filter, int size, int n)
{
    for (int x = 0; x < n; x++) {
        int neg = 0;
        for (int j = 0; j < size; j++)
            if (filter[(size_t)x * size + j] < 0)
                neg += filter[(size_t)x * size + j];
        if (255 * neg < -32768 * 128)
            return 0;
    }
    return 1;
}

void ffhip_cw_plan_job(FFHipCwJob *j, int groups_per_lane, int strip_target)
{
    const int cols_per_wave = 256 * groups_per_lane;
    j->ncb = cdiv(j->dstW, cols_per_wave);
    /* strips of about strip_target (<= 128) output rows, evened out; each strip re-reads 3 halo source rows */
    if (strip_target > 128) strip_target = 128;
    const int n = cdiv(j->dstH, strip_target > 0 ? strip_target : 1);
    j->strip_rows = cdiv(j->dstH, n);
    j->nstrips = cdiv(j->dstH, j->strip_rows);
}

int ffhip_launch_colwalk(FFHipCwArgs &A, int luma_groups, int depth, hipStream_t stream)
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
    const bool opt = A.flags & 2;
#define CW_LAUNCH(LK, DD, PL, OP) hipLaunchKernelGGL((k_sws_colwalk<LK, DD, PL, OP>), grid, block, 0, stream, A)
    if (A.flags & 1) {
        CW_LAUNCH(0, 3, true, false);
    } else if (luma_groups == 2) {
        if (depth == 6) {
            if (opt && (A.flags & 4)) hipLaunchKernelGGL((k_sws_colwalk<1, 6, false, true, true>), grid, block, 0, stream, A);
            else if (opt) CW_LAUNCH(1, 6, false, true);
            else CW_LAUNCH(1, 6, false, false);
        }
        else            { if (opt) CW_LAUNCH(1, 3, false, true); else CW_LAUNCH(1, 3, false, false); }
    } else {
        if (depth == 6) { if (opt) CW_LAUNCH(0, 6, false, true); else CW_LAUNCH(0, 6, false, false); }
        else            { if (opt) CW_LAUNCH(0, 3, false, true); else CW_LAUNCH(0, 3, false, false); }
    }
#undef CW_LAUNCH
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_colwalk_rgb(FFHipCwRgbArgs &A, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    A.ncb = cdiv(A.dstW, 512);
    /* output rows per strip (a wave owns 512 columns of one): 64 when the batch fills the chip twice over at the kernel's three
     * waves per SIMD; shorter strips for smaller batches — a strip re-filters the rows above it, but ONE 4K frame in 64-row strips
     * is 272 waves on 1,024 SIMDs (measured, yuv420p 1080p -> rgb24 4K: 1 frame 59 -> 31 us at 24 rows, 4 frames 76 -> 51 us;
     * 32 frames are fastest at 64) */
    int rows = 64;
    static const int shorter[] = { 48, 32, 24 };
    for (int i = 0; i < 3 && (long long)A.ncb * cdiv(A.dstH, rows) * A.nframes < 2 * 3072; i++)
        rows = shorter[i];
    if (const char *es = FFHIP_KNOB("FFHIP_CWRGB_STRIP")) /* measured variant */
        rows = atoi(es) >= 8 && atoi(es) <= 64 ? atoi(es) : rows;
    const int n = cdiv(A.dstH, rows);
    A.strip_rows = cdiv(A.dstH, n); /* <= 64 */
    A.nstrips = cdiv(A.dstH, A.strip_rows);
    const long long waves = (long long)A.ncb * A.nstrips * A.nframes;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    const char *et = FFHIP_KNOB("FFHIP_CWRGB_DIRECT"); /* measured variant: per-lane 24-byte stores, no LDS transpose */
    const bool tr = !(et && et[0] == '1');
    const char *el = FFHIP_KNOB("FFHIP_CWRGB_LUT"); /* measured variant: 0 = the closed form in VALU instructions */
    const bool lut = !(el && el[0] == '0');
#define CWR_LAUNCH(S, B) do { if (!lut) hipLaunchKernelGGL((k_sws_colwalk_rgb<S, B, 3, true, false>), grid, block, 0, stream, A); \
                              else if (tr) hipLaunchKernelGGL((k_sws_colwalk_rgb<S, B, 3, true>), grid, block, 0, stream, A); \
                              else hipLaunchKernelGGL((k_sws_colwalk_rgb<S, B, 3, false>), grid, block, 0, stream, A); } while (0)
#define CWR_LAUNCH32(S, B) hipLaunchKernelGGL((k_sws_colwalk_rgb<S, B, 3, false>), grid, block, 0, stream, A)
    if (A.sil) {
        switch (A.bgr) {
        case 0: CWR_LAUNCH(true, 0); break;
        case 1: CWR_LAUNCH(true, 1); break;
        case 2: CWR_LAUNCH32(true, 2); break;
        case 3: CWR_LAUNCH32(true, 3); break;
        case 4: CWR_LAUNCH32(true, 4); break;
        default: CWR_LAUNCH32(true, 5); break;
        }
    } else {
        switch (A.bgr) {
        case 0: CWR_LAUNCH(false, 0); break;
        case 1: CWR_LAUNCH(false, 1); break;
        case 2: CWR_LAUNCH32(false, 2); break;
        case 3: CWR_LAUNCH32(false, 3); break;
        case 4: CWR_LAUNCH32(false, 4); break;
        default: CWR_LAUNCH32(false, 5); break;
        }
    }
#undef CWR_LAUNCH32
#undef CWR_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
