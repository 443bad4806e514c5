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
        if (a.c8) {
            /* identity chroma banks: the 8-bit samples as for the luma (min(2 u, 32767) + 64) >> 7, clipped; two bytes per plane */
            uint8_t *u8 = a.dst[1] + (size_t)f * a.dst_fp[1] + (ptrdiff_t)y * a.dst_stride[1] + 2 * g;
            uint8_t *v8 = a.dst[2] + (size_t)f * a.dst_fp[2] + (ptrdiff_t)y * a.dst_stride[2] + 2 * g;
            for (int i = 0; i < n / 2; i++) {
                u8[i] = clip_u8((min(2 * (int)u[i], 32767) + 64) >> 7);
                v8[i] = clip_u8((min(2 * (int)v[i], 32767) + 64) >> 7);
            }
        } else if (n == 4) {
            *reinterpret_cast<uint32_t *>(U + 2 * g) = u[0] | (uint32_t)u[1] << 16;
            *reinterpret_cast<uint32_t *>(V + 2 * g) = v[0] | (uint32_t)v[1] << 16;
        } else {
            U[2 * g] = u[0];
            V[2 * g] = v[0];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < n && a.c8) {
                const int uu = (int)(uint16_t)((a.ru * r[i] + a.gu * gg[i] + a.bu * b[i] + (256 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
                const int vv = (int)(uint16_t)((a.rv * r[i] + a.gv * gg[i] + a.bv * b[i] + (256 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
                (a.dst[1] + (size_t)f * a.dst_fp[1] + (ptrdiff_t)y * a.dst_stride[1])[x0 + i] = clip_u8((min(2 * uu, 32767) + 64) >> 7);
                (a.dst[2] + (size_t)f * a.dst_fp[2] + (ptrdiff_t)y * a.dst_stride[2])[x0 + i] = clip_u8((min(2 * vv, 32767) + 64) >> 7);
            } else if (i < n) {
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

/* ================================================================================================== */
/*
 * k_sws_rgb420 — a packed RGB source into yuv420p / NV12 at the source's size, fused (round 6): what a screen capture or a renderer hands an
 * encoder.  The two-stage form above moves 9.5 bytes per pixel for 5.5 of algorithm (the 14-bit chroma lines go out and come back); here a
 * wave walks down a strip of the picture, a lane owning 4 pixels of every row: the row's luma is written as in k_sws_rgb_in's direct form
 * (identity banks), its two half-width chroma samples per channel — rgb24ToUV_half_c, then hScale16To15_c on the identity bank:
 * min(2 u, 32767) — go into a register ring of vertical pairs T(m) = (row 2m - 1, row 2m), and chroma row y is the 8-tap vertical bank on
 * T(y - 1) .. T(y + 2) with the flat seed 64 << 12, >> 19, clipped (yuv2planeX_8_c / yuv2nv12cX_c, output.c:468-529) — the schedule of
 * sws_down2.hip's vertical half (the bank re-expressed on the windows 2y - 3 .. 2y + 4 of the edge-replicated rows by
 * ffhip_down2_virtual_bank, else this kernel is not used).  Same bytes as the two-stage form (tests/test_gpu_sws_rgbin.py runs both).
 */
typedef uint32_t r4_u4 __attribute__((ext_vector_type(4)));
typedef r4_u4 __attribute__((aligned(4))) r4_u4a;
typedef uint32_t r4_u3 __attribute__((ext_vector_type(3)));
typedef r4_u3 __attribute__((aligned(4))) r4_u3a;
typedef const uint32_t __attribute__((address_space(4))) *r4_cc; /* constant address space: scalar loads */

/* four output bytes: t[i] = seed + T0[i] . c0 + T1[i] . c1 + T2[i] . c2 + T3[i] . c3, clip_u8(t[i] >> 19) */
__device__ __forceinline__ uint32_t r4_v4(const uint32_t (&T0)[4], const uint32_t (&T1)[4], const uint32_t (&T2)[4], const uint32_t (&T3)[4], uint32_t c0,
                                          uint32_t c1, uint32_t c2, uint32_t c3, int seed)
{
    uint32_t out;
    int t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %1, %5, %21, %25\n\t"
        "v_dot2_i32_i16 %2, %6, %21, %25\n\t"
        "v_dot2_i32_i16 %3, %7, %21, %25\n\t"
        "v_dot2_i32_i16 %4, %8, %21, %25\n\t"
        "v_dot2_i32_i16 %1, %9, %22, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %22, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %22, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %22, %4\n\t"
        "v_dot2_i32_i16 %1, %13, %23, %1\n\t"
        "v_dot2_i32_i16 %2, %14, %23, %2\n\t"
        "v_dot2_i32_i16 %3, %15, %23, %3\n\t"
        "v_dot2_i32_i16 %4, %16, %23, %4\n\t"
        "v_dot2_i32_i16 %1, %17, %24, %1\n\t"
        "v_dot2_i32_i16 %2, %18, %24, %2\n\t"
        "v_dot2_i32_i16 %3, %19, %24, %3\n\t"
        "v_dot2_i32_i16 %4, %20, %24, %4\n\t"
        "s_nop 0\n\t"
        "v_ashr_pk_u8_i32 %0, %1, %2, 19\n\t"
        "s_nop 1\n\t"
        "v_ashr_pk_u8_i32 %0, %3, %4, 19 op_sel:[0,0,0,1]"
        : "=&v"(out), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(T0[0]), "v"(T0[1]), "v"(T0[2]), "v"(T0[3]), "v"(T1[0]), "v"(T1[1]), "v"(T1[2]), "v"(T1[3]), "v"(T2[0]), "v"(T2[1]), "v"(T2[2]), "v"(T2[3]),
          "v"(T3[0]), "v"(T3[1]), "v"(T3[2]), "v"(T3[3]), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "v"(seed));
    return out;
}

/* NV: the chroma target is one interleaved plane (cdst[0]: u0 v0 u1 v1 per lane), else two planes (two bytes per lane and plane) */
template <int BPP, bool NV>
__global__ __launch_bounds__(256) void k_sws_rgb420(FFHipRgb420Args A)
{
    const FFHipRgbInArgs &a = A.in;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t upf = (uint32_t)A.ncb * (uint32_t)A.nstrips;
    if (gw >= upf * (uint32_t)A.nframes)
        return;
    const int f = (int)(gw / upf), u = (int)(gw - (uint32_t)f * upf);
    const int strip = u / A.ncb, cb = u - strip * A.ncb;
    const int ngroups = a.w >> 2;
    const int graw = cb * 64 + lane;
    const bool act = graw < ngroups;
    const int g = min(graw, ngroups - 1);
    const int S = A.steps_per_strip;                    /* chroma rows per strip, a multiple of 4 */
    const int ya = strip * S, yb = min(ya + S, A.chrH);
    const int H = a.h;
    const uint8_t *sbase = a.src + (size_t)f * a.src_fp + (size_t)g * (4 * BPP);
    uint8_t *ybase = a.y8 + (size_t)f * a.y8_fp + 4 * (size_t)g;
    uint8_t *cbase0 = A.cdst[0] + (size_t)f * A.cfp + (NV ? 4 : 2) * (size_t)g;
    uint8_t *cbase1 = NV ? cbase0 : A.cdst[1] + (size_t)f * A.cfp + 2 * (size_t)g;
    const int rs = 8 * a.ro, gs = 8 * a.go, bs = 8 * a.bo;
    const bool rfirst = a.ro == 0;
    constexpr int SH = 15; /* RGB2YUV_SHIFT */

    struct Raw { uint32_t q[BPP]; }; /* 4 pixels: 12 or 16 bytes */
    int pr = 2 * ya - 3;               /* next source row to fetch (unclamped) */
    auto load_next = [&](Raw &o) {
        const uint8_t *p = sbase + (ptrdiff_t)min(max(pr, 0), H - 1) * a.src_stride; /* rows above / below the picture replicate the edge row */
        if (BPP == 4) {
            const r4_u4 v = *reinterpret_cast<const r4_u4a *>(p);
            o.q[0] = v.x; o.q[1] = v.y; o.q[2] = v.z; o.q[3 % BPP] = v.w;
        } else {
            const r4_u3 v = *reinterpret_cast<const r4_u3a *>(p);
            o.q[0] = v.x; o.q[1] = v.y; o.q[2] = v.z;
        }
        pr++;
    };
    int cr = 2 * ya - 3;               /* the row the next conversion belongs to */
    /* one source row: its luma (stored when the row is this strip's) and its four chroma samples as 15-bit lines, in the order the target
     * wants them in a dword (NV: u0 v0 u1 v1; planar: u0 u1 v0 v1) */
    auto convert = [&](const Raw &w, int (&h)[4]) {
        int r[4], gg[4], b[4];
        if (BPP == 4) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r[i] = (int)((w.q[i % BPP] >> rs) & 255u); gg[i] = (int)((w.q[i % BPP] >> gs) & 255u); b[i] = (int)((w.q[i % BPP] >> bs) & 255u);
            }
        } else {
            const uint32_t d0 = w.q[0], d1 = w.q[1], d2 = w.q[2];
            const int e0[4] = { (int)(d0 & 255u), (int)(d0 >> 24), (int)((d1 >> 16) & 255u), (int)((d2 >> 8) & 255u) };
            const int e1[4] = { (int)((d0 >> 8) & 255u), (int)(d1 & 255u), (int)(d1 >> 24), (int)((d2 >> 16) & 255u) };
            const int e2[4] = { (int)((d0 >> 16) & 255u), (int)((d1 >> 8) & 255u), (int)(d2 & 255u), (int)(d2 >> 24) };
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r[i] = rfirst ? e0[i] : e2[i]; gg[i] = e1[i]; b[i] = rfirst ? e2[i] : e0[i];
            }
        }
        uint32_t yw = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int yv = (int)(uint16_t)((a.ry * r[i] + a.gy * gg[i] + a.by * b[i] + (32 << (SH - 1)) + (1 << (SH - 7))) >> (SH - 6));
            yw |= (uint32_t)clip_u8((min(2 * yv, 32767) + 64) >> 7) << (8 * i);
        }
        if (act && cr >= 2 * ya && cr < 2 * yb && cr < H)
            *reinterpret_cast<uint32_t *>(ybase + (ptrdiff_t)cr * a.y8_stride) = yw;
        cr++;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r2 = r[2 * i] + r[2 * i + 1], g2 = gg[2 * i] + gg[2 * i + 1], b2 = b[2 * i] + b[2 * i + 1];
            const int uu = (int)(uint16_t)((unsigned)(a.ru * r2 + a.gu * g2 + a.bu * b2 + (256 << SH) + (1 << (SH - 6))) >> (SH - 5));
            const int vv = (int)(uint16_t)((unsigned)(a.rv * r2 + a.gv * g2 + a.bv * b2 + (256 << SH) + (1 << (SH - 6))) >> (SH - 5));
            h[NV ? 2 * i : i] = min(2 * uu, 32767);
            h[NV ? 2 * i + 1 : 2 + i] = min(2 * vv, 32767);
        }
    };
    auto hpair = [&](const Raw &w0, const Raw &w1, uint32_t (&T)[4]) {
        int h0[4], h1[4];
        convert(w0, h0);
        convert(w1, h1);
#pragma unroll
        for (int i = 0; i < 4; i++)
            T[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(h0[i], h1[i]));
    };

    Raw buf[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        load_next(buf[k]);
    /* rows 2a-3 .. 2a+2: the pairs T(a-1), T(a), T(a+1) -> slots 3, 0, 1 (a % 4 == 0) */
    uint32_t ring[4][4];
    hpair(buf[0], buf[1], ring[3]);
    load_next(buf[0]); load_next(buf[1]);
    hpair(buf[2], buf[3], ring[0]);
    load_next(buf[2]); load_next(buf[3]);
    hpair(buf[0], buf[1], ring[1]);
    load_next(buf[0]); load_next(buf[1]);

    const r4_cc vt = (r4_cc)A.vfv;
    for (int y = ya; y < yb; y += 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            /* every row of the trip is computed; the stores alone look at the strip's end */
            Raw &w0 = buf[(2 * k + 2) & 3], &w1 = buf[(2 * k + 3) & 3];
            hpair(w0, w1, ring[(k + 2) & 3]);
            load_next(w0); load_next(w1);
            const int yy = y + k;
            const uint32_t out = r4_v4(ring[(k + 3) & 3], ring[k], ring[(k + 1) & 3], ring[(k + 2) & 3], vt[4 * yy], vt[4 * yy + 1], vt[4 * yy + 2],
                                       vt[4 * yy + 3], 64 << 12);
            if (act && yy < yb) {
                if (NV) {
                    *reinterpret_cast<uint32_t *>(cbase0 + (ptrdiff_t)yy * A.cstride) = out;
                } else {
                    *reinterpret_cast<uint16_t *>(cbase0 + (ptrdiff_t)yy * A.cstride) = (uint16_t)out;
                    *reinterpret_cast<uint16_t *>(cbase1 + (ptrdiff_t)yy * A.cstride) = (uint16_t)(out >> 16);
                }
            }
        }
    }
}

int ffhip_launch_sws_rgb420(FFHipRgb420Args &A, int bpp, int nv, hipStream_t stream)
{
    if (A.in.w <= 0 || A.in.h <= 0 || A.nframes <= 0)
        return 0;
    A.ncb = cdiv(A.in.w >> 2, 64);
    /* strips of 64 chroma rows, shorter until the launch has the waves the chip keeps resident (a strip re-converts six source rows) */
    for (int want = 64; ; want >>= 1) {
        const int n = cdiv(A.chrH, want);
        A.steps_per_strip = cdiv(cdiv(A.chrH, n), 4) * 4;
        A.nstrips = cdiv(A.chrH, A.steps_per_strip);
        if ((long long)A.ncb * A.nstrips * A.nframes >= 8192 || want <= 8)
            break;
    }
    const long long waves = (long long)A.ncb * A.nstrips * A.nframes;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (bpp == 3) {
        if (nv) hipLaunchKernelGGL((k_sws_rgb420<3, true>), grid, block, 0, stream, A);
        else    hipLaunchKernelGGL((k_sws_rgb420<3, false>), grid, block, 0, stream, A);
    } else {
        if (nv) hipLaunchKernelGGL((k_sws_rgb420<4, true>), grid, block, 0, stream, A);
        else    hipLaunchKernelGGL((k_sws_rgb420<4, false>), grid, block, 0, stream, A);
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
