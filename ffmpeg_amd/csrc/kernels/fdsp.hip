/*
 * fdsp.hip — the AVFloatDSPContext vector operations on either side of the MDCT, batched:
 * vector_fmul / vector_fmac_scalar / vector_fmul_scalar / vector_fmul_window / vector_fmul_add / vector_fmul_reverse /
 * butterflies_float (libavutil/float_dsp.c:27-122).  Every output is the reference's one or two IEEE single-precision
 * multiplies and at most one add/sub in the reference's order (library built with -ffp-contract=off: no FMA), so the
 * results are bit-identical.
 *
 * Pure streaming, HBM-bound: one thread per 4 consecutive elements of one vector of the batch (16-byte loads and
 * stores when the operands allow it), blockIdx.y = vector.  The reversed operands (window, fmul_reverse) are read as
 * the mirrored 16-byte group and swizzled in registers, so every access of a wave is one contiguous run.
 */
#include "common.h"
#include "h264_kernels.h"

struct FdspArgs {
    float *dst;
    const float *s0, *s1, *s2;
    size_t pd, p0, p1, p2; /* byte pitches between the vectors of the batch (0: shared) */
    float mul;
    int len;
};

template <bool VEC> struct FdspV;
template <> struct FdspV<true> {
    typedef float4 T;
    static __device__ __forceinline__ T ld(const float *p, int i) { return *reinterpret_cast<const float4 *>(p + i); }
    static __device__ __forceinline__ void st(float *p, int i, T v) { *reinterpret_cast<float4 *>(p + i) = v; }
    /* elements i+3, i+2, i+1, i of the MIRRORED position: x[n-1-i-k], k = 0..3 */
    static __device__ __forceinline__ T ldr(const float *p, int n, int i) { const float4 v = *reinterpret_cast<const float4 *>(p + n - 4 - i); return make_float4(v.w, v.z, v.y, v.x); }
    static __device__ __forceinline__ void str(float *p, int n, int i, T v) { *reinterpret_cast<float4 *>(p + n - 4 - i) = make_float4(v.w, v.z, v.y, v.x); }
};
template <> struct FdspV<false> {
    typedef float4 T; /* only .x is meaningful */
    static __device__ __forceinline__ T ld(const float *p, int i) { return make_float4(p[i], 0, 0, 0); }
    static __device__ __forceinline__ void st(float *p, int i, T v) { p[i] = v.x; }
    static __device__ __forceinline__ T ldr(const float *p, int n, int i) { return make_float4(p[n - 1 - i], 0, 0, 0); }
    static __device__ __forceinline__ void str(float *p, int n, int i, T v) { p[n - 1 - i] = v.x; }
};

#define FD4(expr_x, expr_y, expr_z, expr_w) make_float4(expr_x, expr_y, expr_z, expr_w)

template <int OP, bool VEC>
__global__ __launch_bounds__(256) void k_fdsp(FdspArgs a)
{
    typedef FdspV<VEC> V;
    const int i = (blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1);
    if (i >= a.len)
        return;
    const size_t v = blockIdx.y;
    float *dst = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(a.dst) + v * a.pd);
    const float *s0 = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(a.s0) + v * a.p0);
    const float *s1 = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(a.s1) + v * a.p1);
    const float *s2 = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(a.s2) + v * a.p2);
    const float m = a.mul;
    if (OP == FFHIP_FDSP_FMUL) {
        const float4 x = V::ld(s0, i), y = V::ld(s1, i);
        V::st(dst, i, FD4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w));
    } else if (OP == FFHIP_FDSP_FMAC_SCALAR) {
        const float4 d = V::ld(dst, i), x = V::ld(s0, i);
        V::st(dst, i, FD4(d.x + x.x * m, d.y + x.y * m, d.z + x.z * m, d.w + x.w * m));
    } else if (OP == FFHIP_FDSP_FMUL_SCALAR) {
        const float4 x = V::ld(s0, i);
        V::st(dst, i, FD4(x.x * m, x.y * m, x.z * m, x.w * m));
    } else if (OP == FFHIP_FDSP_FMUL_WINDOW) {
        /* dst[t] = s0[t] w[2n-1-t] - s1[n-1-t] w[t];  dst[2n-1-t] = s0[t] w[t] + s1[n-1-t] w[2n-1-t] */
        const int n = a.len;
        const float4 x = V::ld(s0, i), y = V::ldr(s1, n, i), wi = V::ld(s2, i), wj = V::ldr(s2, 2 * n, i);
        V::st(dst, i, FD4(x.x * wj.x - y.x * wi.x, x.y * wj.y - y.y * wi.y, x.z * wj.z - y.z * wi.z, x.w * wj.w - y.w * wi.w));
        V::str(dst, 2 * n, i, FD4(x.x * wi.x + y.x * wj.x, x.y * wi.y + y.y * wj.y, x.z * wi.z + y.z * wj.z, x.w * wi.w + y.w * wj.w));
    } else if (OP == FFHIP_FDSP_FMUL_ADD) {
        const float4 x = V::ld(s0, i), y = V::ld(s1, i), z = V::ld(s2, i);
        V::st(dst, i, FD4(x.x * y.x + z.x, x.y * y.y + z.y, x.z * y.z + z.z, x.w * y.w + z.w));
    } else if (OP == FFHIP_FDSP_FMUL_REVERSE) {
        const float4 x = V::ld(s0, i), y = V::ldr(s1, a.len, i);
        V::st(dst, i, FD4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w));
    } else { /* butterflies: (v1, v2) = (v1 + v2, v1 - v2) */
        float *v2 = const_cast<float *>(s0);
        const float4 p = V::ld(dst, i), q = V::ld(v2, i);
        V::st(dst, i, FD4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w));
        V::st(v2, i, FD4(p.x - q.x, p.y - q.y, p.z - q.z, p.w - q.w));
    }
}

int ffhip_launch_fdsp(int op, float *dst, size_t pd, const float *s0, size_t p0, const float *s1, size_t p1, const float *s2, size_t p2,
                      float mul, int len, int nvec, hipStream_t stream)
{
    if (len <= 0 || nvec <= 0)
        return 0;
    if (nvec > 65535) { /* gridDim.y: slices of 65,535 vectors */
        for (int v0 = 0; v0 < nvec; v0 += 65535) {
            const int nv = nvec - v0 < 65535 ? nvec - v0 : 65535;
            auto adv = [&](const float *p, size_t pitch) { return p ? reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(p) + (size_t)v0 * pitch) : p; };
            const int r = ffhip_launch_fdsp(op, const_cast<float *>(adv(dst, pd)), pd, adv(s0, p0), p0, adv(s1, p1), p1, adv(s2, p2), p2, mul, len, nv, stream);
            if (r < 0)
                return r;
        }
        return 0;
    }
    FdspArgs a = { dst, s0, s1, s2, pd, p0, p1, p2, mul, len };
    const uintptr_t al = (uintptr_t)dst | pd | (uintptr_t)s0 | p0 | (uintptr_t)(s1 ? s1 : dst) | p1 | (uintptr_t)(s2 ? s2 : dst) | p2;
    const bool vec = !(al & 15) && !(len & 3);
    const dim3 block(256), grid(cdiv(vec ? len / 4 : len, 256), nvec);
#define FD_LAUNCH(OP) do { if (vec) hipLaunchKernelGGL((k_fdsp<OP, true>), grid, block, 0, stream, a); \
                           else hipLaunchKernelGGL((k_fdsp<OP, false>), grid, block, 0, stream, a); } while (0)
    switch (op) {
    case FFHIP_FDSP_FMUL: FD_LAUNCH(FFHIP_FDSP_FMUL); break;
    case FFHIP_FDSP_FMAC_SCALAR: FD_LAUNCH(FFHIP_FDSP_FMAC_SCALAR); break;
    case FFHIP_FDSP_FMUL_SCALAR: FD_LAUNCH(FFHIP_FDSP_FMUL_SCALAR); break;
    case FFHIP_FDSP_FMUL_WINDOW: FD_LAUNCH(FFHIP_FDSP_FMUL_WINDOW); break;
    case FFHIP_FDSP_FMUL_ADD: FD_LAUNCH(FFHIP_FDSP_FMUL_ADD); break;
    case FFHIP_FDSP_FMUL_REVERSE: FD_LAUNCH(FFHIP_FDSP_FMUL_REVERSE); break;
    case FFHIP_FDSP_BUTTERFLIES: FD_LAUNCH(FFHIP_FDSP_BUTTERFLIES); break;
    default:
        ffhip_set_error("ffhip_fdsp: unknown operation %d", op);
        return FFHIP_EINVAL;
    }
#undef FD_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
