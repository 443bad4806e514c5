/*
 * sws_y16rgb.hip — the SECOND stage of a scaled packed-RGB conversion that has no fused kernel (round 5): every ratio the planar
 * scalers run fast and the RGB writers did not — all down-scaling (4K -> 1080p rgb24 for a display or a network input: banks of 5..16
 * taps, which k_sws_colwalk_rgb does not take and the LDS-tiled k_scale_rgb runs at 0.05 of HBM).
 *
 * What the reference does per output line (yuv2packedX -> yuv2rgb_X_c_template, libswscale/vscale.c:126-170, output.c:1789-1840):
 *     Y = (sum lumSrc[j][x] * lumFilter[j] + (1 << 18)) >> 19            not clipped
 *     U, V = the same sums over the chroma lines, one per PAIR of pixels; the table index clips them to 0..255 (fill_table(), yuv2rgb.c:700-712)
 *     r = table_rV[V][Y] ...                                              (yuv2rgb_write(), output.c:1663-1787)
 * The first two lines are exactly what the planar scaler computes for a target with a chroma line per luma line and half the columns
 * — with ONE difference: yuv2planeX clips Y to 8 bits and yuv2rgb_X does not (the tables have head room and a bicubic overshoot to 260
 * is a brighter pixel than 255).  So the first stage is the wide-bank walker (sws_lwalk.hip) on the context's own four banks with its
 * luma job storing the sums >> 19 as int16 (FFHipLwJob.y16), its chroma jobs as ever; this kernel is the third line: the tables'
 * closed form (sws_yuv2rgb.hip: r = clip8((Y * cy + r(V)) >> 16), chroma terms from LDS tables) on 8 pixels per lane, the row segment
 * out through the wave's LDS tile so that a store instruction covers contiguous bytes — the writer of sws_up2rgb.hip.
 *
 * Traffic: the intermediate (2 + 1 bytes per pixel) is written and read once: 1.65x the algorithmic bytes at 4K -> 1080p.  A fused
 * wide-bank kernel would save that; this one serves every ratio at once.
 */
#include "common.h"
#include "sws_kernels.h"

typedef uint32_t yr_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t yr_u4 __attribute__((ext_vector_type(4)));
typedef const uint8_t __attribute__((address_space(1))) *yr_gcp;
typedef uint8_t __attribute__((address_space(1))) *yr_gp;
typedef const yr_u4 __attribute__((address_space(1))) *yr_gc4;
typedef const uint32_t __attribute__((address_space(1))) *yr_gc1;
typedef yr_u2 __attribute__((address_space(1))) *yr_g2;
typedef yr_u4 __attribute__((address_space(1))) *yr_g4;

__device__ __forceinline__ int yr_mad24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
/* one dword of four clipped bytes (a, b, c, d) >> 16: the second instruction writes the high half and keeps the low one */
__device__ __forceinline__ uint32_t yr_pk4(int a, int b, int c, int d)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 16 op_sel:[0,0,0,1]"
        : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}
__device__ __forceinline__ void yr_wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* LAY: 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra (alpha = 255).  A wave = 64 lanes x 8 pixels of one row. */
/* UVI: the chroma samples arrive as ONE plane of (u, v) byte pairs (the exact-2:1 first stage writes them that way, sws_down2.hip) */
template <int LAY, bool UVI = false>
__global__ __launch_bounds__(256) void k_y16_rgb(FFHipY16RgbArgs A)
{
    constexpr int NW = LAY < 2 ? 6 : 8; /* dwords of a lane's 8 pixels */
    __shared__ __attribute__((aligned(16))) uint32_t tiles[4][64 * NW];
    __shared__ uint2 lut[512]; /* [U] = { b(U), gu(U) }, [256 + V] = { r(V), gv(V) }: the chroma terms, cy-scaled, rounding in */
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
    const int G = (A.w + 7) >> 3, ncb = (G + 63) >> 6; /* groups of 8 pixels; the last one may be ragged (its loads stay inside the planes' padded pitch) */
    const uint32_t upf = (uint32_t)ncb * (uint32_t)A.h;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave;
    if (gw >= upf * (uint32_t)A.nframes)
        return;
    const int f = (int)(gw / upf);
    const int u = (int)(gw - (uint32_t)f * upf);
    const int row = u / ncb, cb = u - row * ncb;
    const int graw = cb * 64 + lane;
    const bool act = graw < G;
    const int g = min(graw, G - 1);

    const uint8_t *py = A.y + (size_t)f * A.yfp + (ptrdiff_t)row * A.ystride;
    const uint8_t *pu = A.u + (size_t)f * A.cfp + (ptrdiff_t)row * A.cstride;
    const uint8_t *pv = A.v + (size_t)f * A.cfp + (ptrdiff_t)row * A.cstride;
    uint8_t *pd = A.dst + (size_t)f * A.dfp + (ptrdiff_t)row * A.dstride;
    const yr_u4 yq = *(yr_gc4)((yr_gcp)py + 16u * (uint32_t)g); /* 8 int16 samples */
    uint32_t uq, vq;
    if (UVI) {
        typedef const yr_u2 __attribute__((address_space(1))) *yr_gc2;
        const yr_u2 p = *(yr_gc2)((yr_gcp)pu + 8u * (uint32_t)g); /* u0 v0 u1 v1 | u2 v2 u3 v3 */
        uq = __builtin_amdgcn_perm(p.y, p.x, 0x06040200u);
        vq = __builtin_amdgcn_perm(p.y, p.x, 0x07050301u);
    } else {
        uq = *(yr_gc1)((yr_gcp)pu + 4u * (uint32_t)g);
        vq = *(yr_gc1)((yr_gcp)pv + 4u * (uint32_t)g);
    }

    const char *lutb = reinterpret_cast<const char *>(lut);
    const int cy = __builtin_amdgcn_readfirstlane(A.k.cy);
    int c0[4], c1[4], c2[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const uint2 tu = *reinterpret_cast<const uint2 *>(lutb + (((uq >> (8 * m)) & 0xffu) << 3));
        const uint2 tv = *reinterpret_cast<const uint2 *>(lutb + 2048 + (((vq >> (8 * m)) & 0xffu) << 3));
        constexpr bool BGR = LAY == 1 || LAY == 4 || LAY == 5;
        c0[m] = (int)(BGR ? tu.x : tv.x);
        c1[m] = (int)(tu.y + tv.y);
        c2[m] = (int)(BGR ? tv.x : tu.x);
    }
    const uint32_t yw[4] = { yq.x, yq.y, yq.z, yq.w };
    int val[24];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int ys = (p & 1) ? (int)yw[p >> 1] >> 16 : (int)(int16_t)(yw[p >> 1] & 0xffffu);
        val[3 * p] = yr_mad24(ys, cy, c0[p >> 1]);
        val[3 * p + 1] = yr_mad24(ys, cy, c1[p >> 1]);
        val[3 * p + 2] = yr_mad24(ys, cy, c2[p >> 1]);
    }
    uint32_t w[NW];
    if (LAY >= 2) {
        int alpha = 255 << 16;
        asm("" : "+v"(alpha));
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int x = val[3 * p], y = val[3 * p + 1], z = val[3 * p + 2]; /* (R, G, B) or, BGR layouts, (B, G, R) */
            w[p] = (LAY == 2 || LAY == 4) ? yr_pk4(alpha, x, y, z) : yr_pk4(x, y, z, alpha);
        }
    } else {
#pragma unroll
        for (int d = 0; d < 6; d++)
            w[d] = yr_pk4(val[4 * d], val[4 * d + 1], val[4 * d + 2], val[4 * d + 3]);
    }
    /* transpose through the wave's tile: lane l's 24 / 32 bytes in, 8-byte pieces out — piece i of lane l is bytes 512 i + 8 l of the
     * wave's row segment */
    uint32_t *tile = tiles[wave];
    uint32_t *t = tile + lane * NW;
#pragma unroll
    for (int i = 0; i < NW / 2; i++)
        *reinterpret_cast<uint2 *>(t + 2 * i) = make_uint2(w[2 * i], w[2 * i + 1]);
    yr_wave_sync_lds();
    const int nbytes = (NW / 2) * min(A.w - cb * 512, 512); /* valid bytes of the segment: 3 or 4 per pixel */
    yr_gp d = (yr_gp)pd + (uint32_t)(NW * 256) * (uint32_t)cb + 8u * (uint32_t)lane;
#pragma unroll
    for (int i = 0; i < NW / 2; i++) {
        const uint2 q = *reinterpret_cast<const uint2 *>(tile + i * 128 + lane * 2);
        yr_u2 s;
        s.x = q.x; s.y = q.y;
        const int o = i * 512 + lane * 8;
        if (o + 8 <= nbytes) {
            __builtin_nontemporal_store(s, (yr_g2)(d + i * 512));
        } else if (o < nbytes) { /* the row ends inside this piece (a width that is not a multiple of 8): its bytes one by one */
            typedef uint8_t __attribute__((address_space(1))) *yr_gb;
            const uint64_t v = (uint64_t)q.x | (uint64_t)q.y << 32;
            for (int k = 0; k < nbytes - o; k++)
                ((yr_gb)(d + i * 512))[k] = (uint8_t)(v >> (8 * k));
        }
    }
    (void)act;
}

int ffhip_launch_y16_rgb(const FFHipY16RgbArgs &a, hipStream_t stream)
{
    if (a.nframes <= 0 || a.h <= 0)
        return 0;
    if (a.w <= 0 || (a.w & 1)) {
        ffhip_set_error("ffhip_sws: the second stage of a scaled RGB target takes even widths (got %d)", a.w);
        return FFHIP_EINVAL;
    }
    const long long waves = (long long)cdiv(cdiv(a.w, 8), 64) * a.h * a.nframes;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define YR_L(L)                                                                                   \
    case L:                                                                                       \
        if (a.uvi) hipLaunchKernelGGL((k_y16_rgb<L, true>), grid, block, 0, stream, a);           \
        else hipLaunchKernelGGL((k_y16_rgb<L, false>), grid, block, 0, stream, a);                \
        break;
    switch (a.lay) {
    YR_L(0) YR_L(1) YR_L(2) YR_L(3) YR_L(4) YR_L(5)
    default:
        ffhip_set_error("ffhip_sws: packed layout %d is not one of the RGB writer's", a.lay);
        return FFHIP_EINVAL;
    }
#undef YR_L
    LAUNCH_CHECK();
    return 0;
}
