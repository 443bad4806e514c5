/*
 * h264_qpel.hip — H.264 8-bit luma quarter-pel motion compensation, batched.
 *
 * Bit-exact restatement of put/avg_h264_qpel{16,8,4}_mcXY_8_c (libavcodec/h264qpel_template.c:313-459)
 * stated per output sample:
 *   tap6(a..f) = (c+d)*20 - (b+e)*5 + (a+f)                      (:77-305, the 6-tap lowpass)
 *   H = clip_u8((tap6 along x + 16) >> 5), V likewise along y,
 *   J = clip_u8((tap6 along y of the UNCLIPPED horizontal sums + 512) >> 10)     (the "hv" centre)
 *   quarter positions = rnd_avg (a+b+1)>>1 of two of {F, H, V, J} per the 16-entry table,
 *   avg_ variants rnd_avg the result with dst (op_avg, :461).
 *
 * GPU design: one wave per block; lane (y, xg) owns the 4 horizontally adjacent samples
 * x = 4*xg..4*xg+3 of row y (64 lanes = 16x16; 8x8 and 4x4 blocks use 16 / 4 lanes of their wave).
 * A lane pulls the (up to) 6 source rows x 12 bytes it needs as aligned dwords + v_alignbyte — the
 * rows of neighbouring lanes overlap, so the 21x21 reference footprint is fetched from HBM once and
 * re-served by L1 — keeps everything in registers, averages four samples at a time with the
 * packed rnd_avg32 identity (a|b) - (((a^b) & 0xfefefefe) >> 1) (libavcodec/rnd_avg.h), and writes
 * one dword.  mcXY is wave-uniform, so only the taps a position needs are computed.
 * Algorithmic traffic 2 B per sample (reference read once + destination write).
 */
#include <stdlib.h>

#include "common.h"
#include "h264_kernels.h"

/* 12 source bytes starting at p (any alignment): aligned dword loads, funnel-shifted into place */
struct Row12 { uint32_t w[3]; };

__device__ __forceinline__ Row12 load_row12(const uint8_t *p)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    /* bytes sh..sh+9 are used (10 of the 12): the 4th dword is touched only when it holds one */
    const uint32_t d3 = sh == 3 ? q[3] : 0;
    Row12 r;
    r.w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
    r.w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
    r.w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
    return r;
}

/* FFHIP_MC_EMU (include/ffhip.h; h264_mb.c:229-247 -> videodsp_template.c:24-100): footprint sample (x, y) of the reference picture
 * whose (0, 0) is org, read at clamped coordinates — what emulated_edge_mc() leaves in the decoder's edge buffer.  Byte loads: such
 * blocks are the picture's rim, and nothing outside the picture is touched. */
__device__ __forceinline__ uint32_t qp_emu_dword(const uint8_t *org, ptrdiff_t stride, int x, int y, int pw, int ph)
{
    const uint8_t *row = org + (ptrdiff_t)min(max(y, 0), ph - 1) * stride;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        v |= (uint32_t)row[min(max(x + i, 0), pw - 1)] << (8 * i);
    return v;
}
__device__ __forceinline__ Row12 load_row12_emu(const uint8_t *org, ptrdiff_t stride, int x, int y, int pw, int ph)
{
    Row12 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
        r.w[i] = qp_emu_dword(org, stride, x + 4 * i, y, pw, ph);
    return r;
}

__device__ __forceinline__ int rbyte(const Row12 &r, int i) { return (int)((r.w[i >> 2] >> (8 * (i & 3))) & 0xFF); }
__device__ __forceinline__ int tap6(int a, int b, int c, int d, int e, int f) { return (c + d) * 20 - (b + e) * 5 + (a + f); }
/* unclipped horizontal sum at sample i (0..4) of the lane; stream byte 0 is x-2 */
__device__ __forceinline__ int hraw(const Row12 &r, int i)
{
    return tap6(rbyte(r, i), rbyte(r, i + 1), rbyte(r, i + 2), rbyte(r, i + 3), rbyte(r, i + 4), rbyte(r, i + 5));
}
__device__ __forceinline__ uint32_t rnd_avg4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) & 0xFEFEFEFEu) >> 1); }

__global__ __launch_bounds__(256) void k_h264_qpel(uint8_t *dst, const uint8_t *src, ptrdiff_t stride,
                                                   const FFHipQpelBlock *blocks, int n, int pic_w, int pic_h)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= n)
        return;
    const FFHipQpelBlock blk = blocks[b];
    const int size = 16 >> __builtin_amdgcn_readfirstlane((int)blk.size_idx);
    const int mc = __builtin_amdgcn_readfirstlane((int)blk.mcxy) & 15;
    const bool avg = __builtin_amdgcn_readfirstlane((int)blk.avg) != 0;
    const int per_row = size >> 2;
    const int y = lane / per_row, xg = lane - y * per_row;
    if (y >= size)
        return;
    const int mx = mc & 3, my = mc >> 2;
    const uint8_t *s = src + __builtin_amdgcn_readfirstlane(blk.src_offset) + (ptrdiff_t)y * stride + 4 * xg - 2;
    uint8_t *d = dst + __builtin_amdgcn_readfirstlane(blk.dst_offset) + (ptrdiff_t)y * stride + 4 * xg;

    /* which of F/H/V/J the position combines (appendix A.6 table), all wave-uniform */
    const bool useJ = (mx == 2 && my != 0) || (my == 2 && mx != 0);
    const bool useV = (mx != 2 && my != 0) || (mc == 8);       /* V at x (mx 0,1) or x+1 (mx 3) */
    const bool useH = (my != 2 && mx != 0) || (mc == 2);       /* H at y (my 0,1) or y+1 (my 3) */
    const bool vcol1 = mx == 3, hrow1 = my == 3;

    Row12 r[6]; /* rows y-2 .. y+3 */
    if (pic_w > 0 && (__builtin_amdgcn_readfirstlane((int)blk.flags) & FFHIP_MC_EMU)) {
        const uint8_t *org = src + __builtin_amdgcn_readfirstlane(blk.src_offset);
        const int ex = __builtin_amdgcn_readfirstlane((int)blk.src_x) + 4 * xg - 2, ey = __builtin_amdgcn_readfirstlane((int)blk.src_y) + y;
#pragma unroll
        for (int k = 0; k < 6; k++)
            r[k] = load_row12_emu(org, stride, ex, ey + k - 2, pic_w, pic_h);
    } else if (useV || useJ) {
#pragma unroll
        for (int k = 0; k < 6; k++)
            r[k] = load_row12(s + (ptrdiff_t)(k - 2) * stride);
    } else {
        r[2] = load_row12(s);
        r[3] = hrow1 ? load_row12(s + stride) : r[2];
    }

    uint32_t out = 0;
    uint32_t pj = 0, ph = 0, pv = 0, pf = 0;
    if (useJ) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = tap6(hraw(r[0], i), hraw(r[1], i), hraw(r[2], i), hraw(r[3], i), hraw(r[4], i), hraw(r[5], i));
            pj |= (uint32_t)clip_u8((v + 512) >> 10) << (8 * i);
        }
    }
    if (useH) {
        const Row12 &hr = hrow1 ? r[3] : r[2];
#pragma unroll
        for (int i = 0; i < 4; i++)
            ph |= (uint32_t)clip_u8((hraw(hr, i) + 16) >> 5) << (8 * i);
    }
    if (useV) {
        const int c0 = vcol1 ? 3 : 2; /* stream byte of sample 0's column */
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = tap6(rbyte(r[0], c0 + i), rbyte(r[1], c0 + i), rbyte(r[2], c0 + i), rbyte(r[3], c0 + i),
                               rbyte(r[4], c0 + i), rbyte(r[5], c0 + i));
            pv |= (uint32_t)clip_u8((v + 16) >> 5) << (8 * i);
        }
    }
    /* full-pel samples: F, F(+1,0) for mc30, F(0,+1) for mc03 */
    {
        const Row12 &fr = (mc == 12) ? r[3] : r[2];
        const uint32_t sh = (mc == 3) ? 3 : 2;
        pf = __builtin_amdgcn_alignbyte(fr.w[1], fr.w[0], sh);
    }
    switch (mc) {
    case 0:  out = pf; break;
    case 1: case 3:  out = rnd_avg4(pf, ph); break;
    case 2:  out = ph; break;
    case 4: case 12: out = rnd_avg4(pf, pv); break;
    case 5: case 7: case 13: case 15: out = rnd_avg4(ph, pv); break;
    case 6: case 14: out = rnd_avg4(ph, pj); break;
    case 8:  out = pv; break;
    case 9: case 11: out = rnd_avg4(pv, pj); break;
    default: out = pj; break; /* 10 */
    }
    if (!((reinterpret_cast<uintptr_t>(d)) & 3)) {
        uint32_t *dw = reinterpret_cast<uint32_t *>(d);
        if (avg)
            out = rnd_avg4(*dw, out);
        *dw = out;
    } else {
        for (int i = 0; i < 4; i++) {
            const uint32_t v = (out >> (8 * i)) & 0xFF;
            d[i] = (uint8_t)(avg ? (d[i] + v + 1) >> 1 : v);
        }
    }
}

/* ================================================================================================== */
/*
 * k_h264_qpel_l — the same functions with the block's source footprint and its horizontal 6-tap sums SHARED through
 * wave-private LDS (stride % 4 == 0).  In the kernel above every lane filters the six source rows under its samples
 * horizontally by itself: at the centre positions (J) a horizontal sum is recomputed by the six lanes stacked on it
 * and the kernel is VALU-bound on that redundancy (PMC: ~130 instructions per sample at mc22).  Here
 *   1. the wave copies the (size+5) x (size+5) footprint once, as aligned dwords, into LDS (672 B);
 *   2. if the position needs J, its lanes compute each UNCLIPPED horizontal sum once — packed 16-bit arithmetic, two
 *      samples per instruction (|tap6| <= 10710 fits int16) — into an int16 LDS plane (672 B);
 *   3. lane (y, xg) assembles its four samples: J = vertical tap6 over six int16 rows, H from its own row, V from
 *      the raw rows (packed 16-bit again), F by a funnel shift; quarter positions by the packed rnd_avg32 identity.
 * A wave executes its LDS operations in order and shares nothing with other waves: no barrier.
 */
typedef short qp_s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ qp_s2 qp_pair(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_bit_cast(qp_s2, __builtin_amdgcn_perm(hi, lo, sel));
}
__device__ __forceinline__ qp_s2 qp_tap6(qp_s2 a, qp_s2 b, qp_s2 c, qp_s2 d, qp_s2 e, qp_s2 f)
{
    return (c + d) * (short)20 - (b + e) * (short)5 + (a + f);
}
/* unclipped horizontal sums of the 4 samples whose stream (byte 0 = x-2) is r: (s0,s1) and (s2,s3) as int16 pairs */
__device__ __forceinline__ void qp_hraw4(const Row12 &r, qp_s2 &lo, qp_s2 &hi)
{
    const qp_s2 p01 = qp_pair(r.w[1], r.w[0], 0x0c010c00), p12 = qp_pair(r.w[1], r.w[0], 0x0c020c01);
    const qp_s2 p23 = qp_pair(r.w[1], r.w[0], 0x0c030c02), p34 = qp_pair(r.w[1], r.w[0], 0x0c040c03);
    const qp_s2 p45 = qp_pair(r.w[1], r.w[0], 0x0c050c04), p56 = qp_pair(r.w[1], r.w[0], 0x0c060c05);
    const qp_s2 p67 = qp_pair(r.w[1], r.w[0], 0x0c070c06), p78 = qp_pair(r.w[2], r.w[1], 0x0c040c03);
    lo = qp_tap6(p01, p12, p23, p34, p45, p56);
    hi = qp_tap6(p23, p34, p45, p56, p67, p78);
}
/* clip_u8((v + 16) >> 5) of four packed int16 sums -> 4 bytes */
__device__ __forceinline__ uint32_t qp_round5(qp_s2 lo, qp_s2 hi)
{
    const qp_s2 k16 = { 16, 16 }, z = { 0, 0 }, m = { 255, 255 };
    qp_s2 a = (lo + k16) >> (short)5, b = (hi + k16) >> (short)5;
    a = __builtin_elementwise_min(__builtin_elementwise_max(a, z), m);
    b = __builtin_elementwise_min(__builtin_elementwise_max(b, z), m);
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 0x06040200);
}

/* What one block needs from its record, wave-uniform. */
struct QpBlk {
    int size, mc, soff, doff, ndw, rows;
    uint32_t sh;
    bool avg;
    const uint8_t *sa;
    bool emu;    /* FFHIP_MC_EMU: soff is the reference picture's origin, (sx, sy) the block's position in it */
    int sx, sy;
};
__device__ __forceinline__ QpBlk qp_blk(const FFHipQpelBlock *blocks, int b, const uint8_t *src, ptrdiff_t stride, int pic_w)
{
    const FFHipQpelBlock blk = blocks[b];
    QpBlk q;
    q.size = 16 >> __builtin_amdgcn_readfirstlane((int)blk.size_idx);
    q.mc = __builtin_amdgcn_readfirstlane((int)blk.mcxy) & 15;
    q.avg = __builtin_amdgcn_readfirstlane((int)blk.avg) != 0;
    q.soff = __builtin_amdgcn_readfirstlane(blk.src_offset);
    q.doff = __builtin_amdgcn_readfirstlane(blk.dst_offset);
    q.rows = q.size + 5;
    const uint8_t *s0 = src + q.soff - 2 - 2 * stride;
    q.sh = (uint32_t)(reinterpret_cast<uintptr_t>(s0) & 3); /* same for every row: stride % 4 == 0 */
    q.sa = s0 - q.sh;
    q.ndw = (int)((q.sh + q.size + 5 + 3) >> 2);                     /* <= 7 */
    q.emu = pic_w > 0 && (__builtin_amdgcn_readfirstlane((int)blk.flags) & FFHIP_MC_EMU);
    q.sx = __builtin_amdgcn_readfirstlane((int)blk.src_x);
    q.sy = __builtin_amdgcn_readfirstlane((int)blk.src_y);
    if (q.emu) { /* the footprint is assembled byte by byte from its first sample: no shift */
        q.sh = 0;
        q.ndw = (q.size + 5 + 3) >> 2;
    }
    return q;
}
/* the footprint: rows y-2 .. y+size+2, the aligned dwords that hold bytes x-2 .. x+size+2; three dwords per lane at most */
__device__ __forceinline__ void qp_fetch(const QpBlk &q, const uint8_t *src, ptrdiff_t stride, int pic_w, int pic_h, int lane, uint32_t f[3])
{
    if (q.emu) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int t = lane + 64 * i, r = t >> 3, j = t & 7;
            f[i] = (r < q.rows && j < q.ndw) ? qp_emu_dword(src + q.soff, stride, q.sx - 2 + 4 * j, q.sy - 2 + r, pic_w, pic_h) : 0;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int t = lane + 64 * i, r = t >> 3, j = t & 7;
        f[i] = (r < q.rows && j < q.ndw) ? *reinterpret_cast<const uint32_t *>(q.sa + (ptrdiff_t)r * stride + 4 * j) : 0;
    }
}

/* raw: the footprint as aligned dwords, row r at raw + r * rp (rp = 8: the wave's own plane, filled here from f; rp = QW_WP: a view
 * into the workgroup's window, f == nullptr).  ob != nullptr: the lane's four output samples go to ob[lane] instead of memory
 * (16x16 blocks; the caller stores four blocks' rows together). */
__device__ __forceinline__ void qp_block(const QpBlk &q, const uint32_t *f, uint32_t *raw, const int rp, uint32_t *hb, uint8_t *dst, ptrdiff_t stride,
                                         int lane, uint32_t *ob = nullptr)
{
    const int size = q.size, mc = q.mc, rows = q.rows;
    const uint32_t sh = q.sh;
    const bool avg = q.avg;
    const int per_row = size >> 2, lgp = size == 16 ? 2 : size == 8 ? 1 : 0; /* 4-sample groups per row, and their log2 */
    const int mx = mc & 3, my = mc >> 2;
    const bool useJ = (mx == 2 && my != 0) || (my == 2 && mx != 0);
    const bool useV = (mx != 2 && my != 0) || (mc == 8);
    const bool useH = (my != 2 && mx != 0) || (mc == 2);
    const bool vcol1 = mx == 3, hrow1 = my == 3;

    /* ---- 1. footprint -> LDS ---- */
    if (f) {
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (lane + 64 * i < rows * 8)
                raw[lane + 64 * i] = f[i];
        __builtin_amdgcn_wave_barrier();
    }
    auto stream = [&](int row, int xg) { /* 12 bytes from byte x-2+4*xg of a footprint row */
        const uint32_t *p = raw + row * rp + xg;
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
        const uint32_t d3 = sh == 3 ? p[3] : 0;
        Row12 r;
        r.w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
        r.w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
        r.w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
        return r;
    };

    /* ---- 2. horizontal sums of every footprint row, once ---- */
    if (useJ) {
        for (int t = lane; t < rows * per_row; t += 64) {
            const int r = t >> lgp, xg = t & (per_row - 1);
            qp_s2 lo, hi;
            qp_hraw4(stream(r, xg), lo, hi);
            *reinterpret_cast<uint2 *>(hb + r * 8 + 2 * xg) = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
        }
        __builtin_amdgcn_wave_barrier();
    }

    /* ---- 3. my four samples ---- */
    const int y = lane >> lgp, xg = lane & (per_row - 1);
    if (y < size) {
        uint8_t *d = dst + q.doff + (ptrdiff_t)y * stride + 4 * xg;
        uint32_t pj = 0, ph = 0, pv = 0, pf = 0;
        if (useJ) {
            uint2 h[6];
#pragma unroll
            for (int k = 0; k < 6; k++)
                h[k] = *reinterpret_cast<const uint2 *>(hb + (y + k) * 8 + 2 * xg);
            int v[4];
#pragma unroll
            for (int half = 0; half < 2; half++) {
                auto col = [&](int k) { return __builtin_bit_cast(qp_s2, half ? h[k].y : h[k].x); };
                const qp_s2 s23 = col(2) + col(3), s14 = col(1) + col(4), s05 = col(0) + col(5); /* |.| <= 21420: int16 */
                v[2 * half]     = (int)s23.x * 20 - (int)s14.x * 5 + (int)s05.x;
                v[2 * half + 1] = (int)s23.y * 20 - (int)s14.y * 5 + (int)s05.y;
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
                pj |= (uint32_t)clip_u8((v[i] + 512) >> 10) << (8 * i);
            if (useH) {
                const uint2 hr = h[hrow1 ? 3 : 2];
                ph = qp_round5(__builtin_bit_cast(qp_s2, hr.x), __builtin_bit_cast(qp_s2, hr.y));
            }
        } else if (useH) {
            qp_s2 lo, hi;
            qp_hraw4(stream(y + (hrow1 ? 3 : 2), xg), lo, hi);
            ph = qp_round5(lo, hi);
        }
        if (useV) {
            /* the column under sample i: stream byte 2 + i (3 + i for the right-hand neighbour) of rows y-2 .. y+3 */
            const uint32_t o = sh + 2 + (vcol1 ? 1 : 0);
            qp_s2 c01[6], c23[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const uint32_t *p = raw + (y + k) * rp + xg + (o >> 2);
                const uint32_t w = __builtin_amdgcn_alignbyte(p[1], p[0], o & 3);
                c01[k] = qp_pair(0, w, 0x0c010c00);
                c23[k] = qp_pair(0, w, 0x0c030c02);
            }
            pv = qp_round5(qp_tap6(c01[0], c01[1], c01[2], c01[3], c01[4], c01[5]), qp_tap6(c23[0], c23[1], c23[2], c23[3], c23[4], c23[5]));
        }
        {
            const uint32_t o = sh + (mc == 3 ? 3 : 2);
            const uint32_t *p = raw + (y + (mc == 12 ? 3 : 2)) * rp + xg + (o >> 2);
            pf = __builtin_amdgcn_alignbyte(p[1], p[0], o & 3);
        }
        uint32_t out;
        switch (mc) {
        case 0:  out = pf; break;
        case 1: case 3:  out = rnd_avg4(pf, ph); break;
        case 2:  out = ph; break;
        case 4: case 12: out = rnd_avg4(pf, pv); break;
        case 5: case 7: case 13: case 15: out = rnd_avg4(ph, pv); break;
        case 6: case 14: out = rnd_avg4(ph, pj); break;
        case 8:  out = pv; break;
        case 9: case 11: out = rnd_avg4(pv, pj); break;
        default: out = pj; break; /* 10 */
        }
        if (ob) {
            ob[lane] = out;
        } else if (!((reinterpret_cast<uintptr_t>(d)) & 3)) {
            uint32_t *dw = reinterpret_cast<uint32_t *>(d);
            if (avg)
                out = rnd_avg4(*dw, out);
            *dw = out;
        } else {
            for (int i = 0; i < 4; i++) {
                const uint32_t v = (out >> (8 * i)) & 0xFF;
                d[i] = (uint8_t)(avg ? (d[i] + v + 1) >> 1 : v);
            }
        }
    }
    __builtin_amdgcn_wave_barrier(); /* the next block of this wave reuses the planes */
}

/* NB consecutive blocks per wave: their records, then all their footprints, are in flight before the first is computed, so a
 * wave keeps NB x 3 loads outstanding instead of 3 and pays the record -> footprint -> store latency chain once per NB blocks. */
template <int NB>
__global__ __launch_bounds__(256) void k_h264_qpel_l(uint8_t *dst, const uint8_t *src, ptrdiff_t stride,
                                                     const FFHipQpelBlock *blocks, int n, int pic_w, int pic_h)
{
    __shared__ uint32_t lds[4][21 * 8 + 21 * 8];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int b0 = (blockIdx.x * 4 + wave) * NB;
    if (b0 >= n)
        return;
    uint32_t *raw = lds[wave];             /* [row][8] aligned source dwords, row 0 = y-2 */
    uint32_t *hb = lds[wave] + 21 * 8;     /* [row][8] int16 pairs: unclipped horizontal sums of the block's columns */
    QpBlk q[NB];
    uint32_t f[NB][3];
#pragma unroll
    for (int k = 0; k < NB; k++)
        q[k] = qp_blk(blocks, min(b0 + k, n - 1), src, stride, pic_w);
#pragma unroll
    for (int k = 0; k < NB; k++)
        qp_fetch(q[k], src, stride, pic_w, pic_h, lane, f[k]);
#pragma unroll
    for (int k = 0; k < NB; k++)
        if (b0 + k < n)
            qp_block(q[k], f[k], raw, 8, hb, dst, stride, lane);
}

/* ================================================================================================== */
/*
 * k_h264_qpel_t — k_h264_qpel_l<4> with the memory side rebuilt around what tools/ubench/tilecopy measures: at 16x16 blocks the
 * memory pipeline is bound by the NUMBER of row segments a wave requests, not their bytes (the 32-plane copy skeleton: 0.28 ms with
 * the footprint as three dword loads per lane and 16 B destination rows, 0.20 ms with the two changes below, 0.15 ms for plain
 * 64-byte rows).  stride % 16 == 0.
 *   - the footprint arrives as ONE 16-byte load per lane: lane (row, chunk) fetches an ALIGNED 16-byte chunk (2 or 3 per row cover
 *     the size + 5 bytes; an aligned chunk that holds a needed byte never leaves that byte's page, so nothing beyond the documented
 *     footprint can fault), 21 rows x 3 chunks = 63 lanes;
 *   - the wave's four blocks are computed into an LDS tile and leave as one 16-byte store per lane, lane (row, block): x-adjacent
 *     16x16 blocks — consecutive macroblocks — make 64-byte rows, 4 requests per block instead of 16.
 * The arithmetic is qp_block's, unchanged.  (A workgroup-wide window shared by 16 blocks was measured too: fewer requests still,
 * but the load -> barrier -> compute -> store chain of a 28 KB workgroup left it slower, 0.47 vs 0.35 ms on the mixed case.)
 */
typedef uint32_t qp_u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_h264_qpel_t(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                                                     int pic_w, int pic_h)
{
    __shared__ __align__(16) uint32_t rawp[4][21 * 12];
    __shared__ uint32_t hbp[4][21 * 8];
    __shared__ __align__(16) uint32_t obp[4][4 * 64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int b0 = (blockIdx.x * 4 + wave) * 4;
    if (b0 >= n)
        return;
    uint32_t *raw = rawp[wave], *hb = hbp[wave], *ob = obp[wave];
    const int fr = (lane * 171) >> 9, fc = lane - 3 * fr; /* footprint row and 16-byte chunk of this lane: lane / 3, lane % 3 */
    QpBlk q[4];
    qp_u4 f[4];
    uint32_t sh16[4];
    bool tile = true; /* all four 16x16 with dword-aligned destinations: their rows leave together */
#pragma unroll
    for (int k = 0; k < 4; k++) {
        q[k] = qp_blk(blocks, min(b0 + k, n - 1), src, stride, pic_w);
        tile = tile && q[k].size == 16 && b0 + k < n && !((reinterpret_cast<uintptr_t>(dst) + (uintptr_t)(intptr_t)q[k].doff) & 3);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (q[k].emu) { /* the rim of the picture: the chunk is assembled from clamped byte loads, first footprint byte at chunk 0 byte 0 */
            sh16[k] = 0;
            f[k] = (qp_u4){ 0, 0, 0, 0 };
            if (fr < q[k].rows && fc < 2) {
                const uint8_t *org = src + q[k].soff;
                const int ex = q[k].sx - 2 + 16 * fc, ey = q[k].sy - 2 + fr;
                f[k].x = qp_emu_dword(org, stride, ex, ey, pic_w, pic_h);
                f[k].y = qp_emu_dword(org, stride, ex + 4, ey, pic_w, pic_h);
                if (fc == 0) {
                    f[k].z = qp_emu_dword(org, stride, ex + 8, ey, pic_w, pic_h);
                    f[k].w = qp_emu_dword(org, stride, ex + 12, ey, pic_w, pic_h);
                }
            }
            continue;
        }
        const uint8_t *s0 = src + q[k].soff - 2 - 2 * stride;
        sh16[k] = (uint32_t)(reinterpret_cast<uintptr_t>(s0) & 15);
        const int nch = (int)(sh16[k] + q[k].size + 5 + 15) >> 4;
        f[k] = (fr < q[k].rows && fc < nch) ? *reinterpret_cast<const qp_u4 *>(s0 - sh16[k] + (ptrdiff_t)fr * stride + 16 * fc) : (qp_u4){ 0, 0, 0, 0 };
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (b0 + k >= n)
            break;
        if (lane < 63)
            *reinterpret_cast<qp_u4 *>(raw + fr * 12 + 4 * fc) = f[k];
        __builtin_amdgcn_wave_barrier();
        q[k].sh = sh16[k] & 3;
        qp_block(q[k], nullptr, raw + (sh16[k] >> 2), 12, hb, dst, stride, lane, tile ? ob + 64 * k : nullptr);
    }
    if (tile) {
        const int y = lane >> 2, c = lane & 3;
        qp_u4 o = *reinterpret_cast<const qp_u4 *>(ob + 64 * c + 4 * y);
        const int doff = c == 0 ? q[0].doff : c == 1 ? q[1].doff : c == 2 ? q[2].doff : q[3].doff;
        const bool avg = c == 0 ? q[0].avg : c == 1 ? q[1].avg : c == 2 ? q[2].avg : q[3].avg;
        qp_u4 *dp = reinterpret_cast<qp_u4 *>(dst + doff + (ptrdiff_t)y * stride);
        if (avg) {
            const qp_u4 old = *dp;
            o.x = rnd_avg4(old.x, o.x); o.y = rnd_avg4(old.y, o.y); o.z = rnd_avg4(old.z, o.z); o.w = rnd_avg4(old.w, o.w);
        }
        *dp = o;
    }
}

/* ================================================================================================== */
/*
 * k_h264_qpel_m — the 6-tap filters on the MATRIX CORES (round 4; north_star: "MFMA only if the … product genuinely becomes a dense
 * contraction").  Round 3's PMC on k_h264_qpel_t: 315 VALU + 517 SALU wave-instructions per 16x16 block, issue-bound (VALU 53 % busy,
 * issue stalls 44 %) — the arithmetic, not the memory, caps block MC at 0.21 of HBM.  A 6-tap pass over a block IS a small dense
 * product: 21 footprint bytes against a banded 21 x 16 coefficient matrix whose entries (1, -5, 20) fit int8 unsplit.
 *
 *   stage 1  C1[row][col] = sum_k raw'[row][k] * Th[k][col]      v_mfma_i32_16x16x32_i8, A = 8 footprint bytes per lane straight from
 *            an unaligned global_load_dwordx2 (lane = row m, byte group g: bytes 8g .. 8g+7 of footprint row m), B = Th, a per-lane
 *            constant.  raw' = raw ^ 0x80 (signed), the missing 128 * sum(coefficients) rides in as the accumulator's initial value,
 *            so C1 is the exact unclipped horizontal sum.  B = a shifted identity instead gives the raw samples in the same layout.
 *            The 21 footprint rows are two row blocks (0..15, 16..20).
 *   stage 2  C2[y][col] = sum_r Tv[y][r] * X[r][col]             the same instruction with A = Tv (constant) and B = X, the bytes of
 *            what stage 1 left: a lane's eight C1 values ARE eight consecutive K entries of its column once K is numbered to match
 *            (k' = 8g + r <-> row 4g + r of block A, 8g + 4 + r <-> row 16 + 4g + r), so no lane exchanges anything.  X = raw
 *            samples -> V; X = the sums' low and high bytes (two products, (hi << 8) + lo) -> the centre position J; X = the clipped
 *            horizontal samples against a row-selection matrix -> H on the output rows.
 *   then     clip and pack with v_cvt_pk_i16_i32 / v_pk_ashrrev_i16 / v_sat_pk_u8_i16, average planes with the packed rnd_avg32
 *            identity, turn the lane's 4-row column strip into a 4-sample row with two quad-DPP + v_perm steps, average with the
 *            full-sample plane (an unaligned dword load in that layout) and store through the four-block LDS tile as before.
 *
 * ~60 VALU + 3..7 MFMA (a pipe of their own) per block instead of 315 VALU; no workgroup barrier.
 * Bit-exact: every product and sum is an exact int32.  Workgroups are numbered so that each XCD gets one contiguous eighth of the
 * batch (neighbouring blocks share most of their footprints: one L2 instead of eight).
 */
typedef int qm_i4 __attribute__((ext_vector_type(4)));
typedef long qm_l1 __attribute__((aligned(1)));
typedef uint32_t qm_u1 __attribute__((aligned(1)));

struct QmTab { unsigned long long th[64], id2[64], id3[64], tv[64], sel2[64], sel3[64]; };
constexpr int qm_c6(int t) { return (t == 0 || t == 5) ? 1 : (t == 1 || t == 4) ? -5 : (t == 2 || t == 3) ? 20 : 0; }
constexpr QmTab qm_make()
{
    QmTab t{};
    for (int l = 0; l < 64; l++) {
        const int g = l >> 4, n = l & 15;
        unsigned long long th = 0, id2 = 0, id3 = 0, tv = 0, s2 = 0, s3 = 0;
        for (int j = 0; j < 8; j++) {
            const int k = 8 * g + j;                                    /* stage 1: footprint byte k, output column n */
            const int rho = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);   /* stage 2: K slot 8g + j holds footprint row rho; output row n */
            th  |= (unsigned long long)(unsigned char)(signed char)qm_c6(k - n) << (8 * j);
            id2 |= (unsigned long long)(k == n + 2) << (8 * j);
            id3 |= (unsigned long long)(k == n + 3) << (8 * j);
            tv  |= (unsigned long long)(unsigned char)(signed char)qm_c6(rho - n) << (8 * j);
            s2  |= (unsigned long long)(rho == n + 2) << (8 * j);
            s3  |= (unsigned long long)(rho == n + 3) << (8 * j);
        }
        t.th[l] = th; t.id2[l] = id2; t.id3[l] = id3; t.tv[l] = tv; t.sel2[l] = s2; t.sel3[l] = s3;
    }
    return t;
}
__device__ const QmTab qm_tab = qm_make();

__device__ __forceinline__ uint32_t qm_sat_pk_u8(uint32_t pk) /* two int16 -> two uint8, saturating, in the low half */
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(pk));
    return r;
}
/* four int32 (each within int16 after the shift) -> clip_u8(v >> SH) x 4, byte r = value r */
template <int SH>
__device__ __forceinline__ uint32_t qm_pack4(int a, int b, int c, int d, int bias)
{
    qp_s2 lo = __builtin_amdgcn_cvt_pk_i16(a, b), hi = __builtin_amdgcn_cvt_pk_i16(c, d);
    if (bias) {
        const qp_s2 kb = { (short)bias, (short)bias };
        lo += kb;
        hi += kb;
    }
    if (SH) {
        lo = lo >> (short)SH;
        hi = hi >> (short)SH;
    }
    return __builtin_amdgcn_perm(qm_sat_pk_u8(__builtin_bit_cast(uint32_t, hi)), qm_sat_pk_u8(__builtin_bit_cast(uint32_t, lo)), 0x05040100u);
}
__device__ __forceinline__ long qm_long(uint32_t lo, uint32_t hi) { return (long)(((unsigned long)hi << 32) | lo); }
__device__ __forceinline__ qm_i4 qm_splat(int v) { return (qm_i4){ v, v, v, v }; }

struct QmBlk { int size, mc, soff, doff, sx, sy; bool avg, emu; };

/* which planes a quarter-sample position combines, as bit masks over mcXY (libavcodec/h264qpel_template.c:313-459) */
struct QmFlags { bool useJ, useV, useH, wantF, row3, col3; };
__device__ __forceinline__ QmFlags qm_flags(int mc)
{
    /* bit mc of each constant; the table:   mc  0 1 2 3 | 4 5 6 7 | 8 9 10 11 | 12 13 14 15
     *   J (centre)                               . . . . | . . J . | . J J  J  | .  .  J  .
     *   V (vertical half)                        . . . . | V V . V | V V .  V  | V  V  .  V
     *   H (horizontal half)                      . H H H | . H H H | . . .  .  | .  H  H  H
     *   F (full sample)                          F F . F | F . . . | . . .  .  | F  .  .  .      */
    QmFlags f;
    f.useJ  = (0x4E40u >> mc) & 1;
    f.useV  = (0xBBB0u >> mc) & 1;
    f.useH  = (0xE0EEu >> mc) & 1;
    f.wantF = (0x101Bu >> mc) & 1;
    f.row3  = (mc >> 2) == 3;   /* H one row down / F one row down (mc 12) */
    f.col3  = (mc & 3) == 3;    /* V one column right / F one column right (mc 3) */
    return f;
}

/* One block.  TWO: 16 x 16 (two footprint row blocks).  fa / fb: this lane's 8 bytes of footprint rows m and 16 + m; ff: the
 * full-sample dword of the lane's row-layout position.  Returns the lane's four output samples in the row layout. */
template <bool TWO>
__device__ __forceinline__ uint32_t qm_block(int mc, long fa, long fb, uint32_t ff, int lane, long cTh, long cTv, uint32_t selT1, uint32_t selT2)
{
    const QmFlags F = qm_flags(mc);
    const long K80 = (long)0x8080808080808080ull;
    const long a0 = fa ^ K80, a1 = fb ^ K80;
    uint32_t pj = 0, ph = 0, pv = 0;
    if (F.useJ || F.useH) {
        /* exact horizontal sums of the footprint rows: sum(coefficients) = 32, so +128 * 32 undoes the ^0x80 */
        const qm_i4 h0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a0, cTh, qm_splat(4096), 0, 0, 0);
        qm_i4 h1 = qm_splat(0);
        if (TWO)
            h1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a1, cTh, qm_splat(4096), 0, 0, 0);
        if (F.useJ) {
            /* |sum| <= 10710: a low byte (made signed by ^0x80, +128 * 32 in the accumulator) and a signed high byte */
            const uint32_t p0 = __builtin_amdgcn_perm((uint32_t)h0.y, (uint32_t)h0.x, 0x05010400u), p1 = __builtin_amdgcn_perm((uint32_t)h0.w, (uint32_t)h0.z, 0x05010400u);
            const uint32_t p2 = __builtin_amdgcn_perm((uint32_t)h1.y, (uint32_t)h1.x, 0x05010400u), p3 = __builtin_amdgcn_perm((uint32_t)h1.w, (uint32_t)h1.z, 0x05010400u);
            const long blo = qm_long(__builtin_amdgcn_perm(p1, p0, 0x05040100u), __builtin_amdgcn_perm(p3, p2, 0x05040100u)) ^ K80;
            const long bhi = qm_long(__builtin_amdgcn_perm(p1, p0, 0x07060302u), __builtin_amdgcn_perm(p3, p2, 0x07060302u));
            const qm_i4 chi = __builtin_amdgcn_mfma_i32_16x16x32_i8(cTv, bhi, qm_splat(0), 0, 0, 0);
            const qm_i4 clo = __builtin_amdgcn_mfma_i32_16x16x32_i8(cTv, blo, qm_splat(4096 + 512), 0, 0, 0);
            pj = qm_pack4<0>(((chi.x << 8) + clo.x) >> 10, ((chi.y << 8) + clo.y) >> 10, ((chi.z << 8) + clo.z) >> 10, ((chi.w << 8) + clo.w) >> 10, 0);
        }
        if (F.useH) {
            /* clip((sum + 16) >> 5) of every footprint row, then the rows y + 2 (y + 3 for the positions below) by a 0/1 matrix */
            const long bh = qm_long(qm_pack4<5>(h0.x, h0.y, h0.z, h0.w, 16), qm_pack4<5>(h1.x, h1.y, h1.z, h1.w, 16)) ^ K80;
            const qm_i4 hh = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(F.row3 ? qm_tab.sel3[lane] : qm_tab.sel2[lane]), bh, qm_splat(128), 0, 0, 0);
            ph = (uint32_t)hh.x | (uint32_t)hh.y << 8 | (uint32_t)hh.z << 16 | (uint32_t)hh.w << 24;
        }
    }
    if (F.useV) {
        /* the raw samples of column x (x + 1 for the positions to the right) in the stage-2 layout, then the vertical 6 taps */
        const long cId = (long)(F.col3 ? qm_tab.id3[lane] : qm_tab.id2[lane]);
        const qm_i4 r0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a0, cId, qm_splat(128), 0, 0, 0);
        qm_i4 r1 = qm_splat(0);
        if (TWO)
            r1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a1, cId, qm_splat(128), 0, 0, 0);
        const uint32_t x0 = __builtin_amdgcn_perm((uint32_t)r0.y, (uint32_t)r0.x, 0x0c0c0400u), x1 = __builtin_amdgcn_perm((uint32_t)r0.w, (uint32_t)r0.z, 0x0c0c0400u);
        const uint32_t x2 = __builtin_amdgcn_perm((uint32_t)r1.y, (uint32_t)r1.x, 0x0c0c0400u), x3 = __builtin_amdgcn_perm((uint32_t)r1.w, (uint32_t)r1.z, 0x0c0c0400u);
        const long bv = qm_long(__builtin_amdgcn_perm(x1, x0, 0x05040100u), __builtin_amdgcn_perm(x3, x2, 0x05040100u)) ^ K80;
        const qm_i4 vv = __builtin_amdgcn_mfma_i32_16x16x32_i8(cTv, bv, qm_splat(4096 + 16), 0, 0, 0);
        pv = qm_pack4<5>(vv.x, vv.y, vv.z, vv.w, 0);
    }
    /* a position averages the first and the last of the planes it has (one plane: with itself), still 4-row column strips */
    uint32_t c = rnd_avg4(F.useH ? ph : F.useV ? pv : pj, F.useJ ? pj : F.useV ? pv : ph);
    /* 4 x 4 byte transposition inside each lane quad: lane 4q + j gets row 4g + j, columns 4q .. 4q + 3 */
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)c, 0xB1, 0xf, 0xf, true);     /* quad_perm [1,0,3,2] */
    const uint32_t c1 = __builtin_amdgcn_perm(t1, c, selT1);
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)c1, 0x4E, 0xf, 0xf, true);    /* quad_perm [2,3,0,1] */
    c = __builtin_amdgcn_perm(t2, c1, selT2);
    return mc == 0 ? ff : F.wantF ? rnd_avg4(ff, c) : c;
}

/* The memory side is k_h264_qpel_t's, which tools/ubench/tilecopy found to be the cheapest way to move 16 x 16 tiles (the cost follows
 * the number of row segments a wave requests): ONE aligned 16-byte load per lane brings a block's footprint (21 rows x 3 chunks), it
 * rests in a wave-private LDS plane, and the four blocks of a wave leave as 64-byte rows.  (A first version fed the matrix cores
 * straight from unaligned 8-byte global loads — no LDS at all — and was slower: three load instructions per block and byte-misaligned
 * requests cost more in the texture addresser than the LDS round trip saves.)  The lanes then pick their MFMA operands out of LDS:
 * lane (g, m) the 8 bytes 8g .. 8g + 7 of footprint rows m and 16 + m (three dwords and two funnel shifts each). */
/* one group of four consecutive blocks of a wave: records and footprint chunks */
/* (plain arrays of scalars: an array of record structs is not taken apart by the compiler and lands in scratch memory) */
struct QmRec { int size[4], mc[4], doff[4]; bool avg[4]; };
#define QM_GROUP(G) QmRec G##_q; qp_u4 G##_f[4]; uint32_t G##_sh16[4]; bool G##_tile = false; int G##_b0 = 0
#define QM_ARGS(G) G##_q.size, G##_q.mc, G##_q.doff, G##_q.avg, G##_f, G##_sh16, G##_tile, G##_b0
__device__ __forceinline__ void qm_load(int (&Gsize)[4], int (&Gmc)[4], int (&Gdoff)[4], bool (&Gavg)[4], qp_u4 (&Gf)[4], uint32_t (&Gsh16)[4], bool &Gtile,
                                        int &Gb0, int b0, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                                        int pic_w, int pic_h, int fr, int fc)
{
    Gb0 = b0;
    Gtile = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const QpBlk q = qp_blk(blocks, min(b0 + k, n - 1), src, stride, pic_w);
        Gsize[k] = q.size; Gmc[k] = q.mc; Gdoff[k] = q.doff; Gavg[k] = q.avg;
        Gtile = Gtile && q.size == 16 && b0 + k < n && !((reinterpret_cast<uintptr_t>(dst) + (uintptr_t)(intptr_t)q.doff) & 3);
        if (q.emu) { /* the rim of the picture: the chunk is assembled from clamped byte loads, first footprint byte at chunk 0 byte 0 */
            Gsh16[k] = 0;
            Gf[k] = (qp_u4){ 0, 0, 0, 0 };
            if (fr < q.rows && fc < 2) {
                const uint8_t *org = src + q.soff;
                const int ex = q.sx - 2 + 16 * fc, ey = q.sy - 2 + fr;
                Gf[k].x = qp_emu_dword(org, stride, ex, ey, pic_w, pic_h);
                Gf[k].y = qp_emu_dword(org, stride, ex + 4, ey, pic_w, pic_h);
                if (fc == 0) {
                    Gf[k].z = qp_emu_dword(org, stride, ex + 8, ey, pic_w, pic_h);
                    Gf[k].w = qp_emu_dword(org, stride, ex + 12, ey, pic_w, pic_h);
                }
            }
            continue;
        }
        const uint8_t *s0 = src + q.soff - 2 - 2 * stride;
        Gsh16[k] = (uint32_t)(reinterpret_cast<uintptr_t>(s0) & 15);
        /* the part of the footprint the position reads (round 6): a plane filtered vertically (V, J) wants all size + 5 rows, one filtered
         * horizontally (H, J) all size + 5 columns; the others only the block's own rows / columns (+ 1 for the positions that average with
         * the sample one row down / one column right) */
        const QmFlags F = qm_flags(q.mc);
        const bool rows_all = F.useV || F.useJ, cols_all = F.useH || F.useJ;
        const int r_lo = rows_all ? 0 : 2, r_hi = rows_all ? q.rows - 1 : q.size + 2;
        const int c_lo = (int)Gsh16[k] + (cols_all ? 0 : 2), c_hi = (int)Gsh16[k] + (cols_all ? q.size + 4 : q.size + 2);
        Gf[k] = (fr >= r_lo && fr <= r_hi && 16 * fc + 15 >= c_lo && 16 * fc <= c_hi)
                    ? *reinterpret_cast<const qp_u4 *>(s0 - Gsh16[k] + (ptrdiff_t)fr * stride + 16 * fc) : (qp_u4){ 0, 0, 0, 0 };
    }
}

__device__ __forceinline__ void qm_compute(const int (&Gsize)[4], const int (&Gmc)[4], const int (&Gdoff)[4], const bool (&Gavg)[4], const qp_u4 (&Gf)[4],
                                           const uint32_t (&Gsh16)[4], const bool &Gtile, const int &Gb0, uint8_t *dst, ptrdiff_t stride, int n, uint32_t *raw, uint32_t *ob, int lane, int fr, int fc,
                                           long cTh, long cTv, uint32_t selT1, uint32_t selT2)
{
    const int g = lane >> 4, m = lane & 15;
    const int ry = 4 * g + (lane & 3), rxg = (lane >> 2) & 3; /* the row and 4-sample group this lane owns after the transposition */
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (Gb0 + k < n) {
        if (lane < 63)
            *reinterpret_cast<qp_u4 *>(raw + fr * 12 + 4 * fc) = Gf[k];
        __builtin_amdgcn_wave_barrier();
        const int size = Gsize[k], mc = Gmc[k], last = size + 4;
        const QmFlags F = qm_flags(mc);
        const uint32_t sh = Gsh16[k] & 3;
        const uint32_t *r0p = raw + (Gsh16[k] >> 2);   /* the dword that holds footprint byte 0 of row 0 */
        long fa = 0, fb = 0;
        uint32_t ff = 0;
        if (mc) {
            const uint32_t *pa = r0p + min(m, last) * 12 + 2 * g;
            const uint32_t a0 = pa[0], a1 = pa[1], a2 = pa[2];
            fa = qm_long(__builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh));
            if (size == 16) {
                const uint32_t *pb = r0p + min(16 + m, 20) * 12 + 2 * g;
                const uint32_t b0_ = pb[0], b1_ = pb[1], b2_ = pb[2];
                fb = qm_long(__builtin_amdgcn_alignbyte(b1_, b0_, sh), __builtin_amdgcn_alignbyte(b2_, b1_, sh));
            }
        }
        if (F.wantF) {
            const uint32_t o = sh + 2 + (mc == 3);
            const uint32_t *pf = r0p + (min(ry, size - 1) + 2 + (mc == 12)) * 12 + rxg + (o >> 2);
            ff = __builtin_amdgcn_alignbyte(pf[1], pf[0], o & 3);
        }
        uint32_t out = size == 16 ? qm_block<true>(mc, fa, fb, ff, lane, cTh, cTv, selT1, selT2) : qm_block<false>(mc, fa, fb, ff, lane, cTh, cTv, selT1, selT2);
        if (Gtile) {
            ob[64 * k + 4 * ry + rxg] = out;
        } else if (ry < size && 4 * rxg < size) {
            uint8_t *d = dst + Gdoff[k] + (ptrdiff_t)ry * stride + 4 * rxg;
            if (!((reinterpret_cast<uintptr_t>(d)) & 3)) {
                uint32_t *dw = reinterpret_cast<uint32_t *>(d);
                if (Gavg[k])
                    out = rnd_avg4(*dw, out);
                *dw = out;
            } else {
                for (int i = 0; i < 4; i++) {
                    const uint32_t v = (out >> (8 * i)) & 0xFF;
                    d[i] = (uint8_t)(Gavg[k] ? (d[i] + v + 1) >> 1 : v);
                }
            }
        }
        __builtin_amdgcn_wave_barrier(); /* the next block overwrites the plane */
        }
    }
    if (Gtile) {
        const int y = lane >> 2, c = lane & 3;
        qp_u4 o = *reinterpret_cast<const qp_u4 *>(ob + 64 * c + 4 * y);
        const int doff = c == 0 ? Gdoff[0] : c == 1 ? Gdoff[1] : c == 2 ? Gdoff[2] : Gdoff[3];
        const bool avg = c == 0 ? Gavg[0] : c == 1 ? Gavg[1] : c == 2 ? Gavg[2] : Gavg[3];
        qp_u4 *dp = reinterpret_cast<qp_u4 *>(dst + doff + (ptrdiff_t)y * stride);
        if (avg) {
            const qp_u4 old = *dp;
            o.x = rnd_avg4(old.x, o.x); o.y = rnd_avg4(old.y, o.y); o.z = rnd_avg4(old.z, o.z); o.w = rnd_avg4(old.w, o.w);
        }
        *dp = o;
        __builtin_amdgcn_wave_barrier(); /* the next group refills the tile */
    }
}

/* The operands of one block out of a raw plane: the lane's eight footprint bytes of rows m and 16 + m, and its full-sample dword */
struct QmOps { long fa, fb; uint32_t ff; };
__device__ __forceinline__ QmOps qm_operands(const uint32_t *raw, int size, int mc, uint32_t sh16, int g, int m, int ry, int rxg)
{
    QmOps o = { 0, 0, 0 };
    const int last = size + 4;
    const QmFlags F = qm_flags(mc);
    const uint32_t sh = sh16 & 3;
    const uint32_t *r0p = raw + (sh16 >> 2);   /* the dword that holds footprint byte 0 of row 0 */
    if (mc) {
        const uint32_t *pa = r0p + min(m, last) * 12 + 2 * g;
        const uint32_t a0 = pa[0], a1 = pa[1], a2 = pa[2];
        o.fa = qm_long(__builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh));
        if (size == 16) {
            const uint32_t *pb = r0p + min(16 + m, 20) * 12 + 2 * g;
            const uint32_t b0_ = pb[0], b1_ = pb[1], b2_ = pb[2];
            o.fb = qm_long(__builtin_amdgcn_alignbyte(b1_, b0_, sh), __builtin_amdgcn_alignbyte(b2_, b1_, sh));
        }
    }
    if (F.wantF) {
        const uint32_t q = sh + 2 + (mc == 3);
        const uint32_t *pf = r0p + (min(ry, size - 1) + 2 + (mc == 12)) * 12 + rxg + (q >> 2);
        o.ff = __builtin_amdgcn_alignbyte(pf[1], pf[0], q & 3);
    }
    return o;
}

/* qm_compute with the LDS round trip of block k + 1 (footprint chunk out, operands back) issued BEFORE the matrix-core chain of block k:
 * two raw planes per wave, alternating.  One block per wave is a dependent chain — records, footprint, LDS, MFMA, transpose, LDS, store
 * (R4.1) — and this takes the LDS leg out of it. */
__device__ __forceinline__ void qm_compute_p(const int (&Gsize)[4], const int (&Gmc)[4], const int (&Gdoff)[4], const bool (&Gavg)[4], const qp_u4 (&Gf)[4],
                                             const uint32_t (&Gsh16)[4], const bool &Gtile, const int &Gb0, uint8_t *dst, ptrdiff_t stride, int n, uint32_t *raw0, uint32_t *raw1,
                                             uint32_t *ob, int lane, int fr, int fc, long cTh, long cTv, uint32_t selT1, uint32_t selT2)
{
    const int g = lane >> 4, m = lane & 15;
    const int ry = 4 * g + (lane & 3), rxg = (lane >> 2) & 3;
    if (lane < 63)
        *reinterpret_cast<qp_u4 *>(raw0 + fr * 12 + 4 * fc) = Gf[0];
    __builtin_amdgcn_wave_barrier();
    QmOps cur = qm_operands(raw0, Gsize[0], Gmc[0], Gsh16[0], g, m, ry, rxg), nxt = cur;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (Gb0 + k < n) {
            if (k < 3 && Gb0 + k + 1 < n) {
                uint32_t *rn = (k & 1) ? raw0 : raw1;
                if (lane < 63)
                    *reinterpret_cast<qp_u4 *>(rn + fr * 12 + 4 * fc) = Gf[k + 1];
                __builtin_amdgcn_wave_barrier();
                nxt = qm_operands(rn, Gsize[k + 1], Gmc[k + 1], Gsh16[k + 1], g, m, ry, rxg);
            }
            const int size = Gsize[k], mc = Gmc[k];
            uint32_t out = size == 16 ? qm_block<true>(mc, cur.fa, cur.fb, cur.ff, lane, cTh, cTv, selT1, selT2)
                                      : qm_block<false>(mc, cur.fa, cur.fb, cur.ff, lane, cTh, cTv, selT1, selT2);
            if (Gtile) {
                ob[64 * k + 4 * ry + rxg] = out;
            } else if (ry < size && 4 * rxg < size) {
                uint8_t *d = dst + Gdoff[k] + (ptrdiff_t)ry * stride + 4 * rxg;
                if (!((reinterpret_cast<uintptr_t>(d)) & 3)) {
                    uint32_t *dw = reinterpret_cast<uint32_t *>(d);
                    if (Gavg[k])
                        out = rnd_avg4(*dw, out);
                    *dw = out;
                } else {
                    for (int i = 0; i < 4; i++) {
                        const uint32_t v = (out >> (8 * i)) & 0xFF;
                        d[i] = (uint8_t)(Gavg[k] ? (d[i] + v + 1) >> 1 : v);
                    }
                }
            }
            cur = nxt;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (Gtile) {
        const int y = lane >> 2, c = lane & 3;
        qp_u4 o = *reinterpret_cast<const qp_u4 *>(ob + 64 * c + 4 * y);
        const int doff = c == 0 ? Gdoff[0] : c == 1 ? Gdoff[1] : c == 2 ? Gdoff[2] : Gdoff[3];
        const bool avg = c == 0 ? Gavg[0] : c == 1 ? Gavg[1] : c == 2 ? Gavg[2] : Gavg[3];
        qp_u4 *dp = reinterpret_cast<qp_u4 *>(dst + doff + (ptrdiff_t)y * stride);
        if (avg) {
            const qp_u4 old = *dp;
            o.x = rnd_avg4(old.x, o.x); o.y = rnd_avg4(old.y, o.y); o.z = rnd_avg4(old.z, o.z); o.w = rnd_avg4(old.w, o.w);
        }
        *dp = o;
        __builtin_amdgcn_wave_barrier();
    }
}

/* the product's geometry (one group of four blocks per wave) with qm_compute_p */
__global__ __launch_bounds__(256) void k_h264_qpel_mp(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                                                      int pic_w, int pic_h, int per_xcd)
{
    __shared__ __align__(16) uint32_t rawp[4][2][22 * 12];
    __shared__ __align__(16) uint32_t obp[4][4 * 64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int wg = per_xcd ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int b0 = (wg * 4 + wave) * 4;
    if (b0 >= n)
        return;
    const long cTh = (long)qm_tab.th[lane], cTv = (long)qm_tab.tv[lane];
    const uint32_t selT1 = (lane & 1) ? 0x03070105u : 0x06020400u, selT2 = (lane & 2) ? 0x03020706u : 0x05040100u;
    const int fr = (lane * 171) >> 9, fc = lane - 3 * fr;
    QM_GROUP(A);
    qm_load(QM_ARGS(A), b0, dst, src, stride, blocks, n, pic_w, pic_h, fr, fc);
    qm_compute_p(QM_ARGS(A), dst, stride, n, rawp[wave][0], rawp[wave][1], obp[wave], lane, fr, fc, cTh, cTv, selT1, selT2);
}

/* NG groups of four blocks per wave, the next group's records and footprints in flight while this one is computed */
template <int NG>
__global__ __launch_bounds__(256) void k_h264_qpel_m(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                                                     int pic_w, int pic_h, int per_xcd)
{
    __shared__ __align__(16) uint32_t rawp[4][22 * 12];
    __shared__ __align__(16) uint32_t obp[4][4 * 64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    /* workgroup L runs on XCD L % 8 (observed, not promised: speed only): XCD x takes the workgroups x * per_xcd .. of the batch, so
     * that the blocks whose footprints overlap meet in one L2 (PMC: 12 x fewer bytes fetched past the L2s) */
    const int wg = per_xcd ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int b0 = (wg * 4 + wave) * 4 * NG;
    if (b0 >= n)
        return;
    uint32_t *raw = rawp[wave], *ob = obp[wave];
    const long cTh = (long)qm_tab.th[lane], cTv = (long)qm_tab.tv[lane];
    const uint32_t selT1 = (lane & 1) ? 0x03070105u : 0x06020400u, selT2 = (lane & 2) ? 0x03020706u : 0x05040100u;
    const int fr = (lane * 171) >> 9, fc = lane - 3 * fr; /* footprint row and 16-byte chunk of this lane: lane / 3, lane % 3 */
    QM_GROUP(A);
    QM_GROUP(B);
    qm_load(QM_ARGS(A), b0, dst, src, stride, blocks, n, pic_w, pic_h, fr, fc);
#pragma unroll
    for (int gi = 0; gi < NG; gi += 2) {
        if (gi + 1 < NG && b0 + 4 * (gi + 1) < n)
            qm_load(QM_ARGS(B), b0 + 4 * (gi + 1), dst, src, stride, blocks, n, pic_w, pic_h, fr, fc);
        qm_compute(QM_ARGS(A), dst, stride, n, raw, ob, lane, fr, fc, cTh, cTv, selT1, selT2);
        if (gi + 1 >= NG || b0 + 4 * (gi + 1) >= n)
            break;
        if (gi + 2 < NG && b0 + 4 * (gi + 2) < n)
            qm_load(QM_ARGS(A), b0 + 4 * (gi + 2), dst, src, stride, blocks, n, pic_w, pic_h, fr, fc);
        qm_compute(QM_ARGS(B), dst, stride, n, raw, ob, lane, fr, fc, cTh, cTv, selT1, selT2);
        if (gi + 2 >= NG || b0 + 4 * (gi + 2) >= n)
            break;
    }
}

int ffhip_launch_h264_qpel(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                           hipStream_t stream, int pic_w, int pic_h)
{
    if (n <= 0)
        return 0;
    const char *eo = FFHIP_KNOB("FFHIP_QPEL_OLD"); /* measured variant: the register-only kernel */
    const char *en = FFHIP_KNOB("FFHIP_QPEL_NB");  /* measured variant: blocks per wave */
    const int nb = en ? atoi(en) : 4;
    const char *ew = FFHIP_KNOB("FFHIP_QPEL_W");   /* measured variant: 0 = k_h264_qpel_l (dword footprint loads, a store per block row) */
    const char *em = FFHIP_KNOB("FFHIP_QPEL_M");   /* measured variant: 0 = round 3's kernels (VALU filters) */
    if (!(stride & 15) && !(em && em[0] == '0') && !(eo && eo[0] == '1') && !(ew && ew[0] == '0')) {
        /* the matrix-core kernel: aligned 16-byte chunk loads, so the stride must keep a row's alignment */
        const char *ex = FFHIP_KNOB("FFHIP_QPEL_XCD"); /* measured variant: 0 = workgroups in launch order */
        const char *eg = FFHIP_KNOB("FFHIP_QPEL_NG"); /* measured variant: groups of four blocks per wave (1, 2, 4) */
        const int ng = eg ? atoi(eg) : 1 /* measured: 2 and 4 are slower (0.34 / 0.47 ms against 0.31 at 32 planes) */, remap = !(ex && ex[0] == '0');
        const int per_xcd = cdiv(cdiv(n, 16 * ng), 8);
        const char *ep = FFHIP_KNOB("FFHIP_QPEL_PIPE"); /* measured variant: 1 = the LDS leg of block k + 1 ahead of block k's MFMA chain */
        if (ep && ep[0] == '1')
            hipLaunchKernelGGL(k_h264_qpel_mp, dim3(8 * per_xcd), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h, remap ? per_xcd : 0);
        else if (ng == 4)
            hipLaunchKernelGGL(k_h264_qpel_m<4>, dim3(8 * per_xcd), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h, remap ? per_xcd : 0);
        else if (ng == 2)
            hipLaunchKernelGGL(k_h264_qpel_m<2>, dim3(8 * per_xcd), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h, remap ? per_xcd : 0);
        else
            hipLaunchKernelGGL(k_h264_qpel_m<1>, dim3(8 * per_xcd), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h, remap ? per_xcd : 0);
    } else if (!(stride & 15) && n >= 1024 && !(eo && eo[0] == '1') && !(ew && ew[0] == '0')) {
        hipLaunchKernelGGL(k_h264_qpel_t, dim3(cdiv(n, 16)), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h);
    } else if (!(stride & 3) && !(eo && eo[0] == '1')) {
        if (nb >= 4 && n >= 4 * 4096)
            hipLaunchKernelGGL(k_h264_qpel_l<4>, dim3(cdiv(n, 16)), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h);
        else if (nb >= 2 && n >= 2 * 4096)
            hipLaunchKernelGGL(k_h264_qpel_l<2>, dim3(cdiv(n, 8)), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h);
        else
            hipLaunchKernelGGL(k_h264_qpel_l<1>, dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h);
    }
    else
        hipLaunchKernelGGL(k_h264_qpel, dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, src, stride, blocks, n, pic_w, pic_h);
    LAUNCH_CHECK();
    return 0;
}
