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

int ffhip_launch_h264_qpel(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                           hipStream_t stream, int pic_w, int pic_h)
{
    if (n <= 0)
        return 0;
    const char *eo = FFHIP_KNOB("FFHIP_QPEL_OLD"); /* measured variant: the register-only kernel */
    const char *en = FFHIP_KNOB("FFHIP_QPEL_NB");  /* measured variant: blocks per wave */
    const int nb = en ? atoi(en) : 4;
    const char *ew = FFHIP_KNOB("FFHIP_QPEL_W");   /* measured variant: 0 = k_h264_qpel_l (dword footprint loads, a store per block row) */
    if (!(stride & 15) && n >= 1024 && !(eo && eo[0] == '1') && !(ew && ew[0] == '0')) {
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
