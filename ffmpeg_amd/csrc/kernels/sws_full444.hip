/*
 * sws_full444.hip — planar 4:4:4 into packed RGB at the source's size (round 5): what sws_scale() runs for yuv444p -> rgb24 / bgra.
 *
 * The reference has no table converter for 4:4:4 sources (ff_yuv2rgb_get_func_ptr() serves 4:2:0 / 4:2:2 only, libswscale/yuv2rgb.c:
 * 680-800) and forces SWS_FULL_CHR_H_INT for them (utils.c:1270-1290), so the conversion goes through the scaler with four one-tap banks
 * and ends in yuv2rgb_full_1_c_template + yuv2rgb_write_full (libswscale/output.c:1998-2040,2256-2306):
 *     hScale8To15_c with the one tap 16384:  (s * 16384) >> 7 = s << 7                          (swscale.c:128-142)
 *     Y = buf0[i] * 4 = y << 9;  U = (ubuf0[i] - (128 << 7)) * 4 = (u << 9) - 65536;  V likewise
 *     Y = (Y - y_offset) * y_coeff + (1 << 21)
 *     R = Y + V * v2r;  G = Y + V * v2g + U * u2g;  B = Y + U * u2b          — 32-bit wrapping arithmetic
 *     clipped to 30 bits when any of the three has a top bit set (a no-op otherwise), >> 22
 * Every product has a factor below 2^23 and is taken modulo 2^32: v_mad_i32_i24 (full rate) is exact for it; v_ashr_pk_u8_i32 by 22 is
 * the clip to 30 bits and the shift in one.  11 + 2 VALU instructions per pixel, 6 (7) bytes of traffic: the kernel streams.
 * Before: the LDS-tiled k_scale_rgb's scalar full-chroma writer, 0.065 of HBM (yuv444p 1080p -> rgb24, 16 frames: 0.38 ms).
 *
 * Geometry and writer as sws_y16rgb.hip: a wave = 64 lanes x 8 pixels of one row, the row segment out through the wave's LDS tile in
 * 8-byte pieces with non-temporal stores, ragged row ends byte by byte.
 */
#include "common.h"
#include "sws_kernels.h"

typedef uint32_t f4_u2 __attribute__((ext_vector_type(2)));
typedef const uint8_t __attribute__((address_space(1))) *f4_gcp;
typedef uint8_t __attribute__((address_space(1))) *f4_gp;
typedef f4_u2 __attribute__((aligned(4))) f4_u2a;
typedef const f4_u2a __attribute__((address_space(1))) *f4_gc2;
typedef f4_u2 __attribute__((address_space(1))) *f4_g2;

__device__ __forceinline__ int f4_mad24(int a, int b, int c) /* b: uniform */
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
/* one dword of four bytes clip_u8(x >> 22): the second instruction writes the high half and keeps the low one */
__device__ __forceinline__ uint32_t f4_pk4(int a, int b, int c, int d)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 22\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 22 op_sel:[0,0,0,1]"
        : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}
__device__ __forceinline__ void f4_wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* LAY: 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra (alpha = 255) */
template <int LAY>
__global__ __launch_bounds__(256) void k_yuv444_rgb_full(FFHipFull444Args A)
{
    constexpr int NW = LAY < 2 ? 6 : 8; /* dwords of a lane's 8 pixels */
    __shared__ __attribute__((aligned(16))) uint32_t tiles[4][64 * NW];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int G = (A.w + 7) >> 3, ncb = (G + 63) >> 6; /* groups of 8 pixels; the last one may be ragged */
    const uint32_t upf = (uint32_t)ncb * (uint32_t)A.h;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave;
    if (gw >= upf * (uint32_t)A.nframes)
        return;
    const int f = (int)(gw / upf);
    const int u = (int)(gw - (uint32_t)f * upf);
    const int row = u / ncb, cb = u - row * ncb;
    const int g = min(cb * 64 + lane, G - 1);
    /* the row's last group may reach past the row: it is fetched where the row's last whole 8 bytes lie and shifted down (w >= 8) */
    const int xb = 8 * g, xl = min(xb, A.w - 8);
    const uint32_t sh = 8u * (uint32_t)(xb - xl);
    uint64_t q[3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const uint8_t *s = A.src[p] + (size_t)f * A.sfp[p] + (ptrdiff_t)row * A.sstride[p];
        const f4_u2 v = *(f4_gc2)((f4_gcp)s + (uint32_t)xl);
        q[p] = (((uint64_t)v.y << 32) | v.x) >> sh;
    }
    const int k0 = __builtin_amdgcn_readfirstlane(A.fk[0]), k1 = __builtin_amdgcn_readfirstlane(A.fk[1]);
    const int k2 = __builtin_amdgcn_readfirstlane(A.fk[2]), k3 = __builtin_amdgcn_readfirstlane(A.fk[3]);
    const int k4 = __builtin_amdgcn_readfirstlane(A.fk[4]), k5 = __builtin_amdgcn_readfirstlane(A.fk[5]);
    int val[24];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int y = (int)((q[0] >> (8 * p)) & 255u), cu = (int)((q[1] >> (8 * p)) & 255u), cv = (int)((q[2] >> (8 * p)) & 255u);
        const int Y = (y << 9) - k1, U = (cu << 9) - 65536, V = (cv << 9) - 65536;
        const int yy = f4_mad24(Y, k0, 1 << 21);
        const int R = f4_mad24(V, k2, yy);
        const int Gn = f4_mad24(U, k4, f4_mad24(V, k3, yy));
        const int B = f4_mad24(U, k5, yy);
        constexpr bool BGR = LAY == 1 || LAY == 4 || LAY == 5;
        val[3 * p] = BGR ? B : R;
        val[3 * p + 1] = Gn;
        val[3 * p + 2] = BGR ? R : B;
    }
    uint32_t w[NW];
    if (LAY >= 2) {
        int alpha = 255 << 22;
        asm("" : "+v"(alpha));
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int x = val[3 * p], y = val[3 * p + 1], z = val[3 * p + 2];
            w[p] = (LAY == 2 || LAY == 4) ? f4_pk4(alpha, x, y, z) : f4_pk4(x, y, z, alpha);
        }
    } else {
#pragma unroll
        for (int d = 0; d < 6; d++)
            w[d] = f4_pk4(val[4 * d], val[4 * d + 1], val[4 * d + 2], val[4 * d + 3]);
    }
    /* transpose through the wave's tile: lane l's 24 / 32 bytes in, 8-byte pieces out — piece i of lane l is bytes 512 i + 8 l of the
     * wave's row segment */
    uint32_t *tile = tiles[wave];
    uint32_t *t = tile + lane * NW;
#pragma unroll
    for (int i = 0; i < NW / 2; i++)
        *reinterpret_cast<uint2 *>(t + 2 * i) = make_uint2(w[2 * i], w[2 * i + 1]);
    f4_wave_sync_lds();
    const int nbytes = (NW / 2) * min(A.w - cb * 512, 512); /* valid bytes of the segment: 3 or 4 per pixel */
    uint8_t *pd = A.dst + (size_t)f * A.dfp + (ptrdiff_t)row * A.dstride;
    f4_gp d = (f4_gp)pd + (uint32_t)(NW * 256) * (uint32_t)cb + 8u * (uint32_t)lane;
#pragma unroll
    for (int i = 0; i < NW / 2; i++) {
        const uint2 qq = *reinterpret_cast<const uint2 *>(tile + i * 128 + lane * 2);
        f4_u2 s;
        s.x = qq.x; s.y = qq.y;
        const int o = i * 512 + lane * 8;
        if (o + 8 <= nbytes) {
            __builtin_nontemporal_store(s, (f4_g2)(d + i * 512));
        } else if (o < nbytes) { /* the row ends inside this piece: its bytes one by one */
            typedef uint8_t __attribute__((address_space(1))) *f4_gb;
            const uint64_t v = (uint64_t)qq.x | (uint64_t)qq.y << 32;
            for (int k = 0; k < nbytes - o; k++)
                ((f4_gb)(d + i * 512))[k] = (uint8_t)(v >> (8 * k));
        }
    }
}

int ffhip_launch_full444(const FFHipFull444Args &a, hipStream_t stream)
{
    if (a.nframes <= 0 || a.h <= 0)
        return 0;
    if (a.w < 8) {
        ffhip_set_error("ffhip_sws: the 4:4:4 -> RGB kernel takes widths from 8 (got %d)", a.w);
        return FFHIP_EINVAL;
    }
    const long long waves = (long long)cdiv(cdiv(a.w, 8), 64) * a.h * a.nframes;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    switch (a.lay) {
    case 0: hipLaunchKernelGGL((k_yuv444_rgb_full<0>), grid, block, 0, stream, a); break;
    case 1: hipLaunchKernelGGL((k_yuv444_rgb_full<1>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((k_yuv444_rgb_full<2>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((k_yuv444_rgb_full<3>), grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((k_yuv444_rgb_full<4>), grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL((k_yuv444_rgb_full<5>), grid, block, 0, stream, a); break;
    default:
        ffhip_set_error("ffhip_sws: packed layout %d is not one of the RGB writer's", a.lay);
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}
