/*
 * sws_rgbin.hip — packed 8-bit RGB sources of the legacy scaler (rgb24 / bgr24 / rgba / bgra / argb / abgr into a YUV target).
 *
 * In the reference an RGB source is not a format of its own past the first stage: lumToYV12 / chrToYV12 turn every source line into
 * int16 lines — rgb24ToY_c, rgb24ToUV_c, rgb24ToUV_half_c and their bgr / 32-bit twins (libswscale/input.c:264-400,1068-1190: the 32-bit
 * templates carry every term times 256 and shift 8 more, the same integers) — and from there on the context runs the 16-bit path with
 * hScale16To15_c shifting by 13 (swscale.c:100-128: isAnyRGB -> sh = 13), i.e. exactly what it does for a 14-bit planar source, 4:2:2
 * when the chroma is read at half width (chrSrcHSubSample = 1: srcW even, no SWS_FULL_CHR_H_INP, target chroma no wider than half the
 * source, utils.c:1340-1352) and 4:4:4 otherwise — except that swscale does not dither such a context's 8-bit output (should_dither looks
 * at the SOURCE FORMAT's depth, swscale.c:291).  So the hip path is this kernel in front of the context of that 14-bit planar source with
 * a flat dither: one pass over the picture, 3 or 4 bytes in, 4 (4:2:2) or 6 bytes out per pixel, the converters' integers exactly.
 */
#include "common.h"
#include "sws_kernels.h"

/* a lane converts 4 pixels of a row (two chroma samples at half width) */
template <int BPP, bool HALF>
__global__ __launch_bounds__(256) void k_sws_rgb_in(FFHipRgbInArgs a)
{
    const int g = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    const int x0 = 4 * g;
    if (x0 >= a.w)
        return;
    const uint8_t *s = a.src + (size_t)f * a.src_fp + (ptrdiff_t)y * a.src_stride + (size_t)x0 * BPP;
    const int n = min(4, a.w - x0);
    int r[4], gg[4], b[4];
    if (n == 4 && !(reinterpret_cast<uintptr_t>(s) & 3)) {
        /* the lane's 12 / 16 bytes as dwords (a row of whole dwords: every lane's piece starts on one); the component bytes sit at
         * wave-uniform positions of a pixel */
        const uint32_t *q = reinterpret_cast<const uint32_t *>(s);
        if (BPP == 4) {
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
            const uint32_t d[4] = { d0, d1, d2, d3 };
            const int rs = 8 * a.ro, gs = 8 * a.go, bs = 8 * a.bo;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r[i] = (int)((d[i] >> rs) & 255u); gg[i] = (int)((d[i] >> gs) & 255u); b[i] = (int)((d[i] >> bs) & 255u);
            }
        } else {
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
            /* bytes 0..11: pixel i at 3 i; the middle byte is G, the outer two are (R, B) or (B, R) */
            const int e0[4] = { (int)(d0 & 255u), (int)(d0 >> 24), (int)((d1 >> 16) & 255u), (int)((d2 >> 8) & 255u) };
            const int e1[4] = { (int)((d0 >> 8) & 255u), (int)(d1 & 255u), (int)(d1 >> 24), (int)((d2 >> 16) & 255u) };
            const int e2[4] = { (int)((d0 >> 16) & 255u), (int)((d1 >> 8) & 255u), (int)(d2 & 255u), (int)(d2 >> 24) };
            const bool rfirst = a.ro == 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r[i] = rfirst ? e0[i] : e2[i]; gg[i] = e1[i]; b[i] = rfirst ? e2[i] : e0[i];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint8_t *p = s + (i < n ? i : n - 1) * BPP;
            r[i] = p[a.ro]; gg[i] = p[a.go]; b[i] = p[a.bo];
        }
    }
    constexpr int S = 15; /* RGB2YUV_SHIFT */
    uint16_t *Y = reinterpret_cast<uint16_t *>(a.dst[0] + (size_t)f * a.dst_fp[0] + (ptrdiff_t)y * a.dst_stride[0]) + x0;
    uint16_t *U = reinterpret_cast<uint16_t *>(a.dst[1] + (size_t)f * a.dst_fp[1] + (ptrdiff_t)y * a.dst_stride[1]);
    uint16_t *V = reinterpret_cast<uint16_t *>(a.dst[2] + (size_t)f * a.dst_fp[2] + (ptrdiff_t)y * a.dst_stride[2]);
    uint16_t yv[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        yv[i] = (uint16_t)((a.ry * r[i] + a.gy * gg[i] + a.by * b[i] + (32 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
    if (a.y8) {
        /* the luma banks are the identity (a conversion at the source's size into an 8-bit target): hScale16To15_c on the one tap 1 << 14
         * is min((Y * 16384) >> 13, 32767), yuv2plane1 / yuv2planeX on the one tap 1 << 12 with the flat dither is (. + 64) >> 7, clipped
         * (swscale.c:100-128, output.c:468-486) — the target's luma plane is written here and the walker scales the chroma alone */
        uint8_t *d = a.y8 + (size_t)f * a.y8_fp + (ptrdiff_t)y * a.y8_stride + x0;
        uint32_t w = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
            w |= (uint32_t)clip_u8((min(2 * (int)yv[i], 32767) + 64) >> 7) << (8 * i);
        if (n == 4 && !(reinterpret_cast<uintptr_t>(d) & 3)) {
            *reinterpret_cast<uint32_t *>(d) = w;
        } else {
            for (int i = 0; i < n; i++)
                d[i] = (uint8_t)(w >> (8 * i));
        }
    } else if (n == 4) {
        *reinterpret_cast<uint2 *>(Y) = make_uint2(yv[0] | (uint32_t)yv[1] << 16, yv[2] | (uint32_t)yv[3] << 16);
    } else {
        for (int i = 0; i < n; i++)
            Y[i] = yv[i];
    }
    if (HALF) { /* w is even: n is 2 or 4 */
        uint16_t u[2], v[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r2 = r[2 * i] + r[2 * i + 1], g2 = gg[2 * i] + gg[2 * i + 1], b2 = b[2 * i] + b[2 * i + 1];
            u[i] = (uint16_t)((unsigned)(a.ru * r2 + a.gu * g2 + a.bu * b2 + (256 << S) + (1 << (S - 6))) >> (S - 5));
            v[i] = (uint16_t)((unsigned)(a.rv * r2 + a.gv * g2 + a.bv * b2 + (256 << S) + (1 << (S - 6))) >> (S - 5));
        }
        if (n == 4) {
            *reinterpret_cast<uint32_t *>(U + 2 * g) = u[0] | (uint32_t)u[1] << 16;
            *reinterpret_cast<uint32_t *>(V + 2 * g) = v[0] | (uint32_t)v[1] << 16;
        } else {
            U[2 * g] = u[0];
            V[2 * g] = v[0];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < n) {
                U[x0 + i] = (uint16_t)((a.ru * r[i] + a.gu * gg[i] + a.bu * b[i] + (256 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
                V[x0 + i] = (uint16_t)((a.rv * r[i] + a.gv * gg[i] + a.bv * b[i] + (256 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
            }
    }
}

int ffhip_launch_sws_rgb_in(const FFHipRgbInArgs &a, int bpp, int half, int nframes, hipStream_t stream)
{
    if (a.w <= 0 || a.h <= 0 || nframes <= 0)
        return 0;
    const dim3 grid(cdiv(cdiv(a.w, 4), 256), a.h, nframes), block(256);
    if (bpp == 3) {
        if (half) hipLaunchKernelGGL((k_sws_rgb_in<3, true>), grid, block, 0, stream, a);
        else      hipLaunchKernelGGL((k_sws_rgb_in<3, false>), grid, block, 0, stream, a);
    } else {
        if (half) hipLaunchKernelGGL((k_sws_rgb_in<4, true>), grid, block, 0, stream, a);
        else      hipLaunchKernelGGL((k_sws_rgb_in<4, false>), grid, block, 0, stream, a);
    }
    LAUNCH_CHECK();
    return 0;
}

/*
 * k_sws_widen8 — an 8-bit plane as 16-bit samples (round 6): the input side of an 8-bit source into a 9..14-bit target on the 16-bit
 * walker.  hScale8To15_c (swscale.c:128-142) is hScale16To15_c at depth 8 — the same sums, >> 7 — so the walker runs such a context on
 * planes whose samples sit in the low byte of a word; an interleaved (u, v) byte plane becomes an interleaved plane of words.
 * A lane widens 8 bytes of a row.
 */
__global__ __launch_bounds__(256) void k_sws_widen8(const uint8_t *src, ptrdiff_t sstride, size_t sfp, uint8_t *dst, ptrdiff_t dstride, size_t dfp,
                                                    int wbytes)
{
    const int x0 = 8 * (blockIdx.x * 256 + threadIdx.x), y = blockIdx.y, f = blockIdx.z;
    if (x0 >= wbytes)
        return;
    const uint8_t *s = src + (size_t)f * sfp + (ptrdiff_t)y * sstride + x0;
    uint16_t *d = reinterpret_cast<uint16_t *>(dst + (size_t)f * dfp + (ptrdiff_t)y * dstride) + x0;
    if (x0 + 8 <= wbytes && !(reinterpret_cast<uintptr_t>(s) & 7)) {
        const uint2 v = *reinterpret_cast<const uint2 *>(s);
        uint4 o;
        o.x = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u); o.y = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u);
        o.z = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u); o.w = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u);
        *reinterpret_cast<uint4 *>(d) = o;
    } else {
        for (int i = 0; i < 8 && x0 + i < wbytes; i++)
            d[i] = s[i];
    }
}

int ffhip_launch_sws_widen8(const uint8_t *src, ptrdiff_t sstride, size_t sfp, uint8_t *dst, ptrdiff_t dstride, size_t dfp, int wbytes, int rows,
                            int nframes, hipStream_t stream)
{
    if (wbytes <= 0 || rows <= 0 || nframes <= 0)
        return 0;
    hipLaunchKernelGGL(k_sws_widen8, dim3(cdiv(cdiv(wbytes, 8), 256), rows, nframes), dim3(256), 0, stream, src, sstride, sfp, dst, dstride, dfp, wbytes);
    LAUNCH_CHECK();
    return 0;
}
