/*
 * sws_yuv2rgb.hip — unscaled yuv420p -> rgb24/bgr24, the reference's table-driven converter
 * yuv2rgb_c_24_rgb / yuv2rgb_c_24_bgr (libswscale/yuv2rgb.c:137-228,530-531) with the LUTs of
 * ff_yuv2rgb_c_init_tables (yuv2rgb.c:717-800,901-912) evaluated in closed form:
 *
 *   ramp[k]   = clip_u8((yb0 + k*cy + 0x8000) >> 16),  yb0 = -(384<<16) - 512*cy - oy
 *   R = ramp[yoffs - (crv>>9) + ((V*crv)>>16) + Y]                      (fill_table, :680-691)
 *   B = ramp[yoffs - (cbu>>9) + ((U*cbu)>>16) + Y]
 *   G = ramp[yoffs - (cgu>>9) + ((U*cgu)>>16) - (cgv>>9) + ((V*cgv)>>16) + Y]   (fill_gv_table, :694-703)
 *
 * so no table lives in memory: per chroma sample three bases b = kb + off*cy, per pixel one
 * multiply-add, shift and clamp per channel.  Pure streaming: 4.5 B per pixel, HBM-bound.
 *
 * Work decomposition: one thread = 16 pixels x 2 rows (one chroma row): two 16-B luma loads, two
 * 8-B chroma loads, six 16-B stores; a wave covers 1024 consecutive pixels of a row pair.
 * Pixels written per row: width & ~1 (the reference's 8/4/2-pixel loop never writes an odd tail).
 */
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "sws_kernels.h"

struct Bases { int r, g, b; };

__device__ __forceinline__ Bases chroma_bases(const FFHipYuv2RgbK &k, int U, int V)
{
    Bases o;
    o.r = k.kb + (k.off_r + ((V * k.crv) >> 16)) * k.cy;
    o.b = k.kb + (k.off_b + ((U * k.cbu) >> 16)) * k.cy;
    o.g = k.kb + (k.off_g + ((U * k.cgu) >> 16) + ((V * k.cgv) >> 16)) * k.cy;
    return o;
}

template <bool BGR>
__device__ __forceinline__ void put_px(uint8_t *d, const Bases &b, int ycy)
{
    int r = clip_u8((b.r + ycy) >> 16), g = clip_u8((b.g + ycy) >> 16), bl = clip_u8((b.b + ycy) >> 16);
    d[0] = BGR ? bl : r;
    d[1] = g;
    d[2] = BGR ? r : bl;
}

template <bool BGR, bool VEC>
__global__ __launch_bounds__(256) void k_yuv420p_rgb24(FFHipYuv2RgbArgs a)
{
    const int chunks = (a.wvalid + 15) >> 4;
    const int rowpairs = a.h >> 1;
    const long long total = (long long)chunks * rowpairs * a.nframes;
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total)
        return;
    const int chunk = (int)(id % chunks);
    const int rp = (int)((id / chunks) % rowpairs);
    const int f = (int)(id / ((long long)chunks * rowpairs));
    const int x0 = chunk << 4;

    const uint8_t *py0 = a.y + (size_t)f * a.y_fp + (ptrdiff_t)(2 * rp) * a.y_stride + x0;
    const uint8_t *py1 = py0 + a.y_stride;
    const uint8_t *pu = a.u + (size_t)f * a.u_fp + (ptrdiff_t)rp * a.u_stride + (x0 >> 1);
    const uint8_t *pv = a.v + (size_t)f * a.v_fp + (ptrdiff_t)rp * a.v_stride + (x0 >> 1);
    uint8_t *d0 = a.dst + (size_t)f * a.dst_fp + (ptrdiff_t)(2 * rp + a.dst_y0) * a.dst_stride + 3 * x0;
    uint8_t *d1 = d0 + a.dst_stride;
    const FFHipYuv2RgbK k = a.k;

    if (VEC && x0 + 16 <= a.wvalid) {
        const uint4 y0v = *reinterpret_cast<const uint4 *>(py0);
        const uint4 y1v = *reinterpret_cast<const uint4 *>(py1);
        const uint2 uv = *reinterpret_cast<const uint2 *>(pu);
        const uint2 vv = *reinterpret_cast<const uint2 *>(pv);
        const uint32_t yw0[4] = { y0v.x, y0v.y, y0v.z, y0v.w };
        const uint32_t yw1[4] = { y1v.x, y1v.y, y1v.z, y1v.w };
        const uint32_t uw[2] = { uv.x, uv.y }, vw[2] = { vv.x, vv.y };
        uint32_t o0[12], o1[12];
#pragma unroll
        for (int i = 0; i < 12; i++)
            o0[i] = o1[i] = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int U = (uw[m >> 2] >> (8 * (m & 3))) & 0xFF;
            const int V = (vw[m >> 2] >> (8 * (m & 3))) & 0xFF;
            const Bases b = chroma_bases(k, U, V);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int p = 2 * m + e;
                const int Y0 = (yw0[p >> 2] >> (8 * (p & 3))) & 0xFF;
                const int Y1 = (yw1[p >> 2] >> (8 * (p & 3))) & 0xFF;
                const int c0 = Y0 * k.cy, c1 = Y1 * k.cy;
                const int r0 = clip_u8((b.r + c0) >> 16), g0 = clip_u8((b.g + c0) >> 16), b0 = clip_u8((b.b + c0) >> 16);
                const int r1 = clip_u8((b.r + c1) >> 16), g1 = clip_u8((b.g + c1) >> 16), b1 = clip_u8((b.b + c1) >> 16);
                const int q0 = 3 * p, q1 = 3 * p + 1, q2 = 3 * p + 2;
                o0[q0 >> 2] |= (uint32_t)(BGR ? b0 : r0) << (8 * (q0 & 3));
                o0[q1 >> 2] |= (uint32_t)g0 << (8 * (q1 & 3));
                o0[q2 >> 2] |= (uint32_t)(BGR ? r0 : b0) << (8 * (q2 & 3));
                o1[q0 >> 2] |= (uint32_t)(BGR ? b1 : r1) << (8 * (q0 & 3));
                o1[q1 >> 2] |= (uint32_t)g1 << (8 * (q1 & 3));
                o1[q2 >> 2] |= (uint32_t)(BGR ? r1 : b1) << (8 * (q2 & 3));
            }
        }
        uint4 *s0 = reinterpret_cast<uint4 *>(d0), *s1 = reinterpret_cast<uint4 *>(d1);
        s0[0] = make_uint4(o0[0], o0[1], o0[2], o0[3]);
        s0[1] = make_uint4(o0[4], o0[5], o0[6], o0[7]);
        s0[2] = make_uint4(o0[8], o0[9], o0[10], o0[11]);
        s1[0] = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        s1[1] = make_uint4(o1[4], o1[5], o1[6], o1[7]);
        s1[2] = make_uint4(o1[8], o1[9], o1[10], o1[11]);
    } else {
        const int npairs = (min(16, a.wvalid - x0)) >> 1;
        for (int m = 0; m < npairs; m++) {
            const Bases b = chroma_bases(k, pu[m], pv[m]);
            put_px<BGR>(d0 + 6 * m,     b, py0[2 * m] * k.cy);
            put_px<BGR>(d0 + 6 * m + 3, b, py0[2 * m + 1] * k.cy);
            put_px<BGR>(d1 + 6 * m,     b, py1[2 * m] * k.cy);
            put_px<BGR>(d1 + 6 * m + 3, b, py1[2 * m + 1] * k.cy);
        }
    }
}


/*
 * The streaming kernel proper (VEC planes, any width): one WAVE converts a run of up to 64 x 16
 * pixels of a row pair.  A lane computes its 16 x 2 pixels as before, but the 48 output bytes per row
 * do not go to memory from the lane that made them (that is a 16-byte store every 48 bytes, three
 * partial touches of every 128-byte line): they are transposed through a wave-private 3 KiB LDS tile
 * so that each global_store_dwordx4 of the wave writes 1 KiB of CONTIGUOUS destination.
 * Shift + clamp + pack is v_ashr_pk_u8_i32 (two channels per instruction), byte pairs are merged with
 * v_perm_b32; PLAIN keeps the explicit shift/clamp form of the same arithmetic.
 */
template <bool PLAIN>
__device__ __forceinline__ uint32_t pk16(int a, int b)
{
    uint32_t r;
    if (PLAIN)
        r = (uint32_t)clip_u8(a >> 16) | ((uint32_t)clip_u8(b >> 16) << 8);
    else
        asm("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool BGR, bool PLAIN, bool NTS = true, bool XCD = false, bool NTL = false>
__global__ __launch_bounds__(256) void k_yuv420p_rgb24_t(FFHipYuv2RgbArgs a)
{
    __shared__ uint4 tile[4][192];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int chunks = (a.wvalid + 15) >> 4;
    const int wpr = (chunks + 63) >> 6; /* waves per row pair */
    const int rowpairs = a.h >> 1;
    /* XCD (measured variant): workgroup b runs on XCD b % 8 — number the workgroups so that each XCD converts one contiguous eighth */
    /* workgroup b runs on XCD b % 8 (observed, not promised: speed only).  a.xcd == 1: every XCD converts one contiguous eighth of the
     * launch; > 1 (measured variant, round 6): XCD-contiguous chunks of 1 << (xcd - 1) workgroups dealt round-robin */
    uint32_t bidx = blockIdx.x;
    if (XCD || a.xcd == 1) {
        bidx = (blockIdx.x & 7u) * ((gridDim.x + 7u) >> 3) + (blockIdx.x >> 3);
    } else if (a.xcd > 1) {
        const uint32_t lg = (uint32_t)a.xcd - 1u, C = 1u << lg, full = gridDim.x & ~(8u * C - 1u);
        if (bidx < full) {
            const uint32_t x = bidx & 7u, sl = bidx >> 3;
            bidx = (((sl >> lg) << 3) + x) * C + (sl & (C - 1u));
        }
    }
    const uint32_t gw = bidx * 4u + (uint32_t)wave; /* < 2^31, checked by the launcher */
    /* FLAT (width % 16 == 0, at least 64 chunks per row): the 16-pixel chunks of a frame's row pairs are numbered straight through and
     * a wave takes 64 consecutive ones, across the end of a row pair if need be — at 3840 columns (240 chunks = 3.75 waves) one wave in
     * four would otherwise run three quarters empty.  A wave then touches at most two row pairs: rp0 and rp0 + 1. */
    const bool flat = a.flat != 0;
    int f, rp, x0, nfull, rp0 = 0, bnd = 0, c0 = 0, cpf = 0;
    bool full;
    if (flat) {
        cpf = chunks * rowpairs;
        const uint32_t wpf = ((uint32_t)cpf + 63u) >> 6;
        if (gw >= wpf * (uint32_t)a.nframes)
            return;
        f = (int)(gw / wpf);
        c0 = (int)(gw - (uint32_t)f * wpf) * 64;
        rp0 = c0 / chunks;
        bnd = (rp0 + 1) * chunks;
        const int c = c0 + lane;
        rp = rp0 + (c >= bnd);
        x0 = (c - rp * chunks) << 4;
        full = c < cpf;
        rp = min(rp, rowpairs - 1);
        nfull = min(cpf - c0, 64);
    } else {
        if (gw >= (uint32_t)wpr * (uint32_t)rowpairs * (uint32_t)a.nframes)
            return;
        const int wr = (int)(gw % (uint32_t)wpr);
        rp = (int)((gw / (uint32_t)wpr) % (uint32_t)rowpairs);
        f = (int)(gw / ((uint32_t)wpr * (uint32_t)rowpairs));
        x0 = (wr * 64 + lane) << 4;
        full = x0 + 16 <= a.wvalid;
        nfull = min(max((a.wvalid >> 4) - wr * 64, 0), 64); /* whole 16-pixel chunks of this wave */
        c0 = wr * 64;
    }

    const uint8_t *py0 = a.y + (size_t)f * a.y_fp + (ptrdiff_t)(2 * rp) * a.y_stride;
    const uint8_t *py1 = py0 + a.y_stride;
    const uint8_t *pu = a.u + (size_t)f * a.u_fp + (ptrdiff_t)rp * a.u_stride;
    const uint8_t *pv = a.v + (size_t)f * a.v_fp + (ptrdiff_t)rp * a.v_stride;
    uint8_t *dframe = a.dst + (size_t)f * a.dst_fp + (ptrdiff_t)a.dst_y0 * a.dst_stride;
    uint8_t *d0 = dframe + (ptrdiff_t)(2 * rp) * a.dst_stride;
    uint8_t *d1 = d0 + a.dst_stride;
    const FFHipYuv2RgbK k = a.k;
    uint4 *my = tile[wave];

    uint32_t o0[12], o1[12];
    if (full) {
        typedef uint32_t ld_u4 __attribute__((ext_vector_type(4)));
        typedef uint32_t ld_u2 __attribute__((ext_vector_type(2)));
        ld_u4 y0v, y1v;
        ld_u2 uv, vv;
        if (NTL) {
            y0v = __builtin_nontemporal_load(reinterpret_cast<const ld_u4 *>(py0 + (uint32_t)x0));
            y1v = __builtin_nontemporal_load(reinterpret_cast<const ld_u4 *>(py1 + (uint32_t)x0));
            uv = __builtin_nontemporal_load(reinterpret_cast<const ld_u2 *>(pu + (uint32_t)(x0 >> 1)));
            vv = __builtin_nontemporal_load(reinterpret_cast<const ld_u2 *>(pv + (uint32_t)(x0 >> 1)));
        } else {
            y0v = *reinterpret_cast<const ld_u4 *>(py0 + (uint32_t)x0);
            y1v = *reinterpret_cast<const ld_u4 *>(py1 + (uint32_t)x0);
            uv = *reinterpret_cast<const ld_u2 *>(pu + (uint32_t)(x0 >> 1));
            vv = *reinterpret_cast<const ld_u2 *>(pv + (uint32_t)(x0 >> 1));
        }
        const uint32_t yw0[4] = { y0v.x, y0v.y, y0v.z, y0v.w };
        const uint32_t yw1[4] = { y1v.x, y1v.y, y1v.z, y1v.w };
        const uint32_t uw[2] = { uv.x, uv.y }, vw[2] = { vv.x, vv.y };
        int v0[48], v1[48]; /* pre-shift value of every output byte of the two rows */
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int U = (uw[m >> 2] >> (8 * (m & 3))) & 0xFF;
            const int V = (vw[m >> 2] >> (8 * (m & 3))) & 0xFF;
            const Bases b = chroma_bases(k, U, V);
            const int c0 = BGR ? b.b : b.r, c2 = BGR ? b.r : b.b;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int p = 2 * m + e;
                const int Y0 = (yw0[p >> 2] >> (8 * (p & 3))) & 0xFF;
                const int Y1 = (yw1[p >> 2] >> (8 * (p & 3))) & 0xFF;
                v0[3 * p] = Y0 * k.cy + c0; v0[3 * p + 1] = Y0 * k.cy + b.g; v0[3 * p + 2] = Y0 * k.cy + c2;
                v1[3 * p] = Y1 * k.cy + c0; v1[3 * p + 1] = Y1 * k.cy + b.g; v1[3 * p + 2] = Y1 * k.cy + c2;
            }
        }
#pragma unroll
        for (int d = 0; d < 12; d++) {
            o0[d] = __builtin_amdgcn_perm(pk16<PLAIN>(v0[4 * d + 2], v0[4 * d + 3]), pk16<PLAIN>(v0[4 * d], v0[4 * d + 1]),
                                          0x05040100);
            o1[d] = __builtin_amdgcn_perm(pk16<PLAIN>(v1[4 * d + 2], v1[4 * d + 3]), pk16<PLAIN>(v1[4 * d], v1[4 * d + 1]),
                                          0x05040100);
        }
    }
    const uint32_t run = (uint32_t)c0 * 48u; /* !FLAT: byte offset of this wave's run in a destination row */
#pragma unroll
    for (int row = 0; row < 2; row++) {
        const uint32_t *o = row ? o1 : o0;
        if (full) {
            my[3 * lane + 0] = make_uint4(o[0], o[1], o[2], o[3]);
            my[3 * lane + 1] = make_uint4(o[4], o[5], o[6], o[7]);
            my[3 * lane + 2] = make_uint4(o[8], o[9], o[10], o[11]);
        }
        wave_sync_lds();
        uint8_t *drow = (row ? d1 : d0) + run;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int p = lane + 64 * j;
            if (p < 3 * nfull) {
                if (flat) {
                    /* piece p is third (p % 3) of chunk c0 + p / 3, which lies in row pair rp0 or rp0 + 1 */
                    const int pc = (p * 171) >> 9, cc = c0 + pc, rr = rp0 + (cc >= bnd);
                    *reinterpret_cast<uint4 *>(dframe + (ptrdiff_t)(2 * rr + row) * a.dst_stride + 48u * (uint32_t)(cc - rr * chunks) + 16u * (uint32_t)(p - 3 * pc)) = my[p];   /* (measured variant: plain stores) */
                } else {
                    if (NTS) { typedef uint32_t nt_u4 __attribute__((ext_vector_type(4))); const uint4 t = my[p];
                               __builtin_nontemporal_store(nt_u4{ t.x, t.y, t.z, t.w }, reinterpret_cast<nt_u4 *>(drow + 16u * (uint32_t)p)); }
                    else *reinterpret_cast<uint4 *>(drow + 16u * (uint32_t)p) = my[p];
                }
            }
        }
        wave_sync_lds();
    }
    if (!flat && !full && x0 < a.wvalid) { /* the ragged last chunk of a row: straight to memory */
        const int npairs = (a.wvalid - x0) >> 1;
        for (int m = 0; m < npairs; m++) {
            const Bases b = chroma_bases(k, pu[(x0 >> 1) + m], pv[(x0 >> 1) + m]);
            put_px<BGR>(d0 + 3 * x0 + 6 * m,     b, py0[x0 + 2 * m] * k.cy);
            put_px<BGR>(d0 + 3 * x0 + 6 * m + 3, b, py0[x0 + 2 * m + 1] * k.cy);
            put_px<BGR>(d1 + 3 * x0 + 6 * m,     b, py1[x0 + 2 * m] * k.cy);
            put_px<BGR>(d1 + 3 * x0 + 6 * m + 3, b, py1[x0 + 2 * m + 1] * k.cy);
        }
    }
}

/*
 * 32-bit packed targets (argb / rgba / abgr / bgra, alpha = 255): yuv2rgb_c_32 (libswscale/yuv2rgb.c:522) with its 32-bit
 * tables (yuv2rgb.c:943-966: the same clipped ramp shifted to the component's byte, 255 in the alpha byte) in closed
 * form.  A pixel is a dword, so no transposition is needed: one lane = 4 pixels x 2 rows writes one aligned 16-byte
 * store per row and a wave's store instruction covers 1 KiB of contiguous destination.  5.5 B per pixel, HBM-bound.
 */
template <int LAYOUT> /* 2 argb, 3 rgba, 4 abgr, 5 bgra */
__device__ __forceinline__ uint32_t px32(int r, int g, int b)
{
    /* values are the pre-shift sums: v_ashr_pk_u8_i32 does >> 16 and the clamp; 255 << 16 yields the alpha byte */
    constexpr int A = 255 << 16;
    uint32_t lo, hi;
    if (LAYOUT == 2)      { lo = pk16<false>(A, r); hi = pk16<false>(g, b); }
    else if (LAYOUT == 3) { lo = pk16<false>(r, g); hi = pk16<false>(b, A); }
    else if (LAYOUT == 4) { lo = pk16<false>(A, b); hi = pk16<false>(g, r); }
    else                  { lo = pk16<false>(b, g); hi = pk16<false>(r, A); }
    return __builtin_amdgcn_perm(hi, lo, 0x05040100);
}

template <int LAYOUT>
__device__ __forceinline__ void put_px32(uint8_t *d, const Bases &b, int ycy)
{
    const int r = clip_u8((b.r + ycy) >> 16), g = clip_u8((b.g + ycy) >> 16), bl = clip_u8((b.b + ycy) >> 16);
    if (LAYOUT == 2)      { d[0] = 255; d[1] = (uint8_t)r; d[2] = (uint8_t)g; d[3] = (uint8_t)bl; }
    else if (LAYOUT == 3) { d[0] = (uint8_t)r; d[1] = (uint8_t)g; d[2] = (uint8_t)bl; d[3] = 255; }
    else if (LAYOUT == 4) { d[0] = 255; d[1] = (uint8_t)bl; d[2] = (uint8_t)g; d[3] = (uint8_t)r; }
    else                  { d[0] = (uint8_t)bl; d[1] = (uint8_t)g; d[2] = (uint8_t)r; d[3] = 255; }
}

template <int LAYOUT, bool VEC, bool NTS = true>
__global__ __launch_bounds__(256) void k_yuv420p_rgb32(FFHipYuv2RgbArgs a)
{
    /* one lane = 4 pixels x 2 rows: a dword of each luma row, two bytes of U and of V in, one aligned 16-byte store per
     * row out — every load and store instruction of a wave covers one contiguous run (256 B of luma, 1 KiB of pixels) */
    const int chunks = (a.wvalid + 3) >> 2;
    const int rowpairs = a.h >> 1;
    const long long total = (long long)chunks * rowpairs * a.nframes;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total)
        return;
    const int chunk = (int)(id % chunks);
    const int rp = (int)((id / chunks) % rowpairs);
    const int f = (int)(id / ((long long)chunks * rowpairs));
    const int x0 = chunk << 2;
    const uint8_t *py0 = a.y + (size_t)f * a.y_fp + (ptrdiff_t)(2 * rp) * a.y_stride + x0;
    const uint8_t *py1 = py0 + a.y_stride;
    const uint8_t *pu = a.u + (size_t)f * a.u_fp + (ptrdiff_t)rp * a.u_stride + (x0 >> 1);
    const uint8_t *pv = a.v + (size_t)f * a.v_fp + (ptrdiff_t)rp * a.v_stride + (x0 >> 1);
    uint8_t *d0 = a.dst + (size_t)f * a.dst_fp + (ptrdiff_t)(2 * rp + a.dst_y0) * a.dst_stride + 4 * x0;
    uint8_t *d1 = d0 + a.dst_stride;
    const FFHipYuv2RgbK k = a.k;
    if (VEC && x0 + 4 <= a.wvalid) {
        const uint32_t yw0 = *reinterpret_cast<const uint32_t *>(py0), yw1 = *reinterpret_cast<const uint32_t *>(py1);
        const uint32_t uw = *reinterpret_cast<const uint16_t *>(pu), vw = *reinterpret_cast<const uint16_t *>(pv);
        uint32_t o0[4], o1[4];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const Bases b = chroma_bases(k, (int)((uw >> (8 * m)) & 0xFF), (int)((vw >> (8 * m)) & 0xFF));
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int p = 2 * m + e;
                const int c0 = (int)((yw0 >> (8 * p)) & 0xFF) * k.cy, c1 = (int)((yw1 >> (8 * p)) & 0xFF) * k.cy;
                o0[p] = px32<LAYOUT>(b.r + c0, b.g + c0, b.b + c0);
                o1[p] = px32<LAYOUT>(b.r + c1, b.g + c1, b.b + c1);
            }
        }
        if (NTS) { /* non-temporal: the destination is written once and not read back (3-6 % on the 24-bit kernel, profiles/r05_rgb24_variants.txt) */
            typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(nt_u4{ o0[0], o0[1], o0[2], o0[3] }, reinterpret_cast<nt_u4 *>(d0));
            __builtin_nontemporal_store(nt_u4{ o1[0], o1[1], o1[2], o1[3] }, reinterpret_cast<nt_u4 *>(d1));
        } else {
            *reinterpret_cast<uint4 *>(d0) = make_uint4(o0[0], o0[1], o0[2], o0[3]);
            *reinterpret_cast<uint4 *>(d1) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        }
    } else {
        const int npairs = (min(4, a.wvalid - x0)) >> 1;
        for (int m = 0; m < npairs; m++) {
            const Bases b = chroma_bases(k, pu[m], pv[m]);
            put_px32<LAYOUT>(d0 + 8 * m,     b, py0[2 * m] * k.cy);
            put_px32<LAYOUT>(d0 + 8 * m + 4, b, py0[2 * m + 1] * k.cy);
            put_px32<LAYOUT>(d1 + 8 * m,     b, py1[2 * m] * k.cy);
            put_px32<LAYOUT>(d1 + 8 * m + 4, b, py1[2 * m + 1] * k.cy);
        }
    }
}

/*
 * The converter's remaining forms in one plain kernel: 4:2:2 sources (YUV422FUNC, yuv2rgb.c:238-320: the second luma row of a pair
 * takes the chroma row of its own), a source alpha plane into the alpha byte of a 32-bit target (yuva2rgba_c / yuva2argb_c,
 * yuv2rgb.c:524-529, PUTRGBA :88-93), and the planar target (yuv420p_gbrp_c / yuv422p_gbrp_c, PUTGBRP :127-135).  One lane = 4 pixels
 * x 2 rows, as k_yuv420p_rgb32: a dword of each luma (and alpha) row and two bytes of each chroma row in; 12 / 16 / 3 x 4 bytes per
 * row out, dword or 16-byte stores when VEC (aligned planes), bytes otherwise.
 */
template <int LAYOUT, bool VEC>
__global__ __launch_bounds__(256) void k_yuv2rgb_forms(FFHipYuv2RgbArgs a)
{
    const int chunks = (a.wvalid + 3) >> 2;
    const int rowpairs = a.h >> 1;
    const long long total = (long long)chunks * rowpairs * a.nframes;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total)
        return;
    const int chunk = (int)(id % chunks);
    const int rp = (int)((id / chunks) % rowpairs);
    const int f = (int)(id / ((long long)chunks * rowpairs));
    const int x0 = chunk << 2;
    const int npx = min(4, a.wvalid - x0);
    const FFHipYuv2RgbK k = a.k;
    constexpr int BPP = LAYOUT < 2 ? 3 : LAYOUT < 6 ? 4 : 1;
#pragma unroll
    for (int row = 0; row < 2; row++) {
        const int y = 2 * rp + row;
        const int crow = a.c422 ? y : rp;
        const uint8_t *py = a.y + (size_t)f * a.y_fp + (ptrdiff_t)y * a.y_stride + x0;
        const uint8_t *pu = a.u + (size_t)f * a.u_fp + (ptrdiff_t)crow * a.u_stride + (x0 >> 1);
        const uint8_t *pv = a.v + (size_t)f * a.v_fp + (ptrdiff_t)crow * a.v_stride + (x0 >> 1);
        const uint8_t *pa = a.alpha ? a.alpha + (size_t)f * a.alpha_fp + (ptrdiff_t)y * a.alpha_stride + x0 : nullptr;
        uint8_t *d = a.dst + (size_t)f * a.dst_fp + (ptrdiff_t)(y + a.dst_y0) * a.dst_stride + BPP * x0;
        uint32_t yw = 0, uw = 0, vw = 0, aw = 0xFFFFFFFFu;
        if (VEC && npx == 4) {
            yw = *reinterpret_cast<const uint32_t *>(py);
            uw = *reinterpret_cast<const uint16_t *>(pu);
            vw = *reinterpret_cast<const uint16_t *>(pv);
            if (pa)
                aw = *reinterpret_cast<const uint32_t *>(pa);
        } else {
            for (int p = 0; p < npx; p++) {
                yw |= (uint32_t)py[p] << (8 * p);
                if (pa)
                    aw = (aw & ~(0xFFu << (8 * p))) | ((uint32_t)pa[p] << (8 * p));
            }
            for (int m = 0; m < (npx >> 1); m++) {
                uw |= (uint32_t)pu[m] << (8 * m);
                vw |= (uint32_t)pv[m] << (8 * m);
            }
        }
        uint8_t R[4], G[4], B[4];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const Bases b = chroma_bases(k, (int)((uw >> (8 * m)) & 0xFF), (int)((vw >> (8 * m)) & 0xFF));
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int p = 2 * m + e;
                const int c = (int)((yw >> (8 * p)) & 0xFF) * k.cy;
                R[p] = (uint8_t)clip_u8((b.r + c) >> 16); G[p] = (uint8_t)clip_u8((b.g + c) >> 16); B[p] = (uint8_t)clip_u8((b.b + c) >> 16);
            }
        }
        if (LAYOUT == 6) {
            uint8_t *d1 = a.dst1 + (size_t)f * a.dst1_fp + (ptrdiff_t)(y + a.dst_y0) * a.dst1_stride + x0;
            uint8_t *d2 = a.dst2 + (size_t)f * a.dst2_fp + (ptrdiff_t)(y + a.dst_y0) * a.dst2_stride + x0;
            if (VEC && npx == 4) {
                *reinterpret_cast<uint32_t *>(d)  = G[0] | (G[1] << 8) | (G[2] << 16) | ((uint32_t)G[3] << 24);
                *reinterpret_cast<uint32_t *>(d1) = B[0] | (B[1] << 8) | (B[2] << 16) | ((uint32_t)B[3] << 24);
                *reinterpret_cast<uint32_t *>(d2) = R[0] | (R[1] << 8) | (R[2] << 16) | ((uint32_t)R[3] << 24);
            } else {
                for (int p = 0; p < npx; p++) { d[p] = G[p]; d1[p] = B[p]; d2[p] = R[p]; }
            }
        } else if (LAYOUT >= 2) {
            uint32_t o[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t A = (aw >> (8 * p)) & 0xFF;
                o[p] = LAYOUT == 2 ? A | (R[p] << 8) | (G[p] << 16) | ((uint32_t)B[p] << 24)
                     : LAYOUT == 3 ? R[p] | (G[p] << 8) | (B[p] << 16) | (A << 24)
                     : LAYOUT == 4 ? A | (B[p] << 8) | (G[p] << 16) | ((uint32_t)R[p] << 24)
                                   : B[p] | (G[p] << 8) | (R[p] << 16) | (A << 24);
            }
            if (VEC && npx == 4) {
                *reinterpret_cast<uint4 *>(d) = make_uint4(o[0], o[1], o[2], o[3]);
            } else {
                for (int p = 0; p < npx; p++)
                    for (int j = 0; j < 4; j++)
                        d[4 * p + j] = (uint8_t)(o[p] >> (8 * j));
            }
        } else {
            uint8_t o[12];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                o[3 * p] = LAYOUT == 1 ? B[p] : R[p]; o[3 * p + 1] = G[p]; o[3 * p + 2] = LAYOUT == 1 ? R[p] : B[p];
            }
            if (VEC && npx == 4) {
#pragma unroll
                for (int j = 0; j < 3; j++)
                    reinterpret_cast<uint32_t *>(d)[j] = o[4 * j] | (o[4 * j + 1] << 8) | (o[4 * j + 2] << 16) | ((uint32_t)o[4 * j + 3] << 24);
            } else {
                for (int j = 0; j < 3 * npx; j++)
                    d[j] = o[j];
            }
        }
    }
}

static int launch_forms(const FFHipYuv2RgbArgs &a, int layout, hipStream_t stream)
{
    const long long total4 = (long long)((a.wvalid + 3) >> 2) * (a.h >> 1) * a.nframes;
    if (total4 <= 0)
        return 0;
    if (total4 >= (1LL << 31) * 256) {
        ffhip_set_error("ffhip_sws: batch too large for one launch");
        return FFHIP_EINVAL;
    }
    bool vec = !(((uintptr_t)a.y | (size_t)a.y_stride | a.y_fp | (uintptr_t)a.dst | (size_t)a.dst_stride | a.dst_fp) & 3) &&
               !(((uintptr_t)a.u | (uintptr_t)a.v | (size_t)a.u_stride | (size_t)a.v_stride | a.u_fp | a.v_fp) & 1);
    if (a.alpha)
        vec = vec && !(((uintptr_t)a.alpha | (size_t)a.alpha_stride | a.alpha_fp) & 3);
    if (layout >= 2 && layout < 6)
        vec = vec && !(((uintptr_t)a.dst | (size_t)a.dst_stride | a.dst_fp) & 15);
    if (layout == 6)
        vec = vec && !(((uintptr_t)a.dst1 | (uintptr_t)a.dst2 | (size_t)a.dst1_stride | (size_t)a.dst2_stride | a.dst1_fp | a.dst2_fp) & 3);
    const dim3 grid((unsigned)((total4 + 255) / 256)), block(256);
#define LF(LY) do { if (vec) hipLaunchKernelGGL((k_yuv2rgb_forms<LY, true>), grid, block, 0, stream, a); \
                    else hipLaunchKernelGGL((k_yuv2rgb_forms<LY, false>), grid, block, 0, stream, a); } while (0)
    switch (layout) {
    case 0: LF(0); break;
    case 1: LF(1); break;
    case 2: LF(2); break;
    case 3: LF(3); break;
    case 4: LF(4); break;
    case 5: LF(5); break;
    case 6: LF(6); break;
    default: return FFHIP_EINVAL;
    }
#undef LF
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_yuv420p_rgb24(const FFHipYuv2RgbArgs &a, int layout, hipStream_t stream)
{
    if (a.c422 || a.alpha || layout == 6)
        return launch_forms(a, layout, stream);
    const int bgr = layout == 1;
    const int chunks = (a.wvalid + 15) >> 4;
    const long long total = (long long)chunks * (a.h >> 1) * a.nframes;
    if (total <= 0)
        return 0;
    const bool vec = !(((uintptr_t)a.y | (uintptr_t)a.dst | (size_t)a.y_stride | (size_t)a.dst_stride | a.y_fp |
                        a.dst_fp) & 15) &&
                     !(((uintptr_t)a.u | (uintptr_t)a.v | (size_t)a.u_stride | (size_t)a.v_stride | a.u_fp | a.v_fp) & 7) &&
                     a.dst_stride > 0;
    const dim3 block(256);
    if (layout >= 2) {
        const long long total4 = (long long)((a.wvalid + 3) >> 2) * (a.h >> 1) * a.nframes; /* one thread per 4 pixels x 2 rows */
        const dim3 grid((unsigned)((total4 + 255) / 256));
        if (total4 >= (1LL << 31) * 256) {
            ffhip_set_error("ffhip_sws: batch too large for one launch");
            return FFHIP_EINVAL;
        }
        const char *ev32 = FFHIP_KNOB("FFHIP_YUV2RGB_VARIANT"); /* "st": plain stores (measured variant) */
        const bool pst32 = ev32 && strstr(ev32, "st");
#define L32(LY) do { if (vec && pst32) hipLaunchKernelGGL((k_yuv420p_rgb32<LY, true, false>), grid, block, 0, stream, a); \
                     else if (vec) hipLaunchKernelGGL((k_yuv420p_rgb32<LY, true>), grid, block, 0, stream, a); \
                     else hipLaunchKernelGGL((k_yuv420p_rgb32<LY, false>), grid, block, 0, stream, a); } while (0)
        switch (layout) {
        case 2: L32(2); break;
        case 3: L32(3); break;
        case 4: L32(4); break;
        default: L32(5); break;
        }
#undef L32
        LAUNCH_CHECK();
        return 0;
    }
    const char *ev = FFHIP_KNOB("FFHIP_YUV2RGB_VARIANT"); /* "old": per-lane strided stores; "plain": no v_ashr_pk */
    /* "flat": measured variant (chunks numbered through the row pairs, no wave three quarters empty at 3840 columns): 1.5 % SLOWER than
     * waves that stay inside a row pair (0.4765 / 0.4738 against 0.4681 / 0.4660 ms for 64 4K frames, same box, alternating) — the kernel
     * is bound by the memory system, not by lanes */
    const bool flat = !(a.wvalid & 15) && chunks >= 64 && ev && ev[0] == 'f';
    const long long waves = flat ? (((long long)chunks * (a.h >> 1) + 63) >> 6) * a.nframes : (long long)((chunks + 63) >> 6) * (a.h >> 1) * a.nframes;
    if (vec && waves < (1LL << 31) && !(ev && ev[0] == 'o')) {
        const dim3 grid((unsigned)((waves + 3) / 4));
        const bool plain = ev && ev[0] == 'p';
        FFHipYuv2RgbArgs af = a;
        af.flat = flat;
        /* measured variants (rgb24 only): "st" plain stores (the kernel up to round 4: 3-6 % slower than the non-temporal ones, profiles/
         * r05_rgb24_variants.txt), "xcd" XCD-contiguous numbering, "ntl" non-temporal loads as well */
        const bool pst = ev && strstr(ev, "st"), xcd = ev && strstr(ev, "xcd"), ntl = ev && strstr(ev, "ntl");
        if (xcd)
            af.xcd = 1 + atoi(strstr(ev, "xcd") + 3);          /* "xcd": an eighth per XCD; "xcd4": chunks of 16 workgroups */
        else if (ev)
            af.xcd = 0;                                        /* a named variant overrides the context's tuned numbering */
        if (!bgr && (pst || xcd || ntl)) { /* words: "st", "xcd", "ntl", "st+xcd" (none of them inside "old" / "plain" / "flat") */
            if (pst && xcd) hipLaunchKernelGGL((k_yuv420p_rgb24_t<false, false, false>), grid, block, 0, stream, af);
            else if (pst)   hipLaunchKernelGGL((k_yuv420p_rgb24_t<false, false, false, false>), grid, block, 0, stream, af);
            else if (ntl)   hipLaunchKernelGGL((k_yuv420p_rgb24_t<false, false, true, false, true>), grid, block, 0, stream, af);
            else            hipLaunchKernelGGL((k_yuv420p_rgb24_t<false, false, true>), grid, block, 0, stream, af);
            LAUNCH_CHECK();
            return 0;
        }
        if (bgr) {
            if (plain) hipLaunchKernelGGL((k_yuv420p_rgb24_t<true, true>), grid, block, 0, stream, af);
            else       hipLaunchKernelGGL((k_yuv420p_rgb24_t<true, false>), grid, block, 0, stream, af);
        } else {
            if (plain) hipLaunchKernelGGL((k_yuv420p_rgb24_t<false, true>), grid, block, 0, stream, af);
            else       hipLaunchKernelGGL((k_yuv420p_rgb24_t<false, false>), grid, block, 0, stream, af);
        }
        LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((unsigned)((total + 255) / 256));
    if (bgr) {
        if (vec) hipLaunchKernelGGL((k_yuv420p_rgb24<true, true>), grid, block, 0, stream, a);
        else     hipLaunchKernelGGL((k_yuv420p_rgb24<true, false>), grid, block, 0, stream, a);
    } else {
        if (vec) hipLaunchKernelGGL((k_yuv420p_rgb24<false, true>), grid, block, 0, stream, a);
        else     hipLaunchKernelGGL((k_yuv420p_rgb24<false, false>), grid, block, 0, stream, a);
    }
    LAUNCH_CHECK();
    return 0;
}
