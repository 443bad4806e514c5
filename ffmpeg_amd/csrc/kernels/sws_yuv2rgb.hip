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

int ffhip_launch_yuv420p_rgb24(const FFHipYuv2RgbArgs &a, int bgr, hipStream_t stream)
{
    const int chunks = (a.wvalid + 15) >> 4;
    const long long total = (long long)chunks * (a.h >> 1) * a.nframes;
    if (total <= 0)
        return 0;
    const bool vec = !(((uintptr_t)a.y | (uintptr_t)a.dst | (size_t)a.y_stride | (size_t)a.dst_stride | a.y_fp |
                        a.dst_fp) & 15) &&
                     !(((uintptr_t)a.u | (uintptr_t)a.v | (size_t)a.u_stride | (size_t)a.v_stride | a.u_fp | a.v_fp) & 7) &&
                     a.dst_stride > 0;
    const dim3 block(256), grid((unsigned)((total + 255) / 256));
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
