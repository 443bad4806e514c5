/*
 * tx_radix.hip — av_tx float FFT / MDCT of 256, 512 and 1024 complex points with the transform held in REGISTERS (round 4).
 *
 * Reference: ff_tx_fft (libavutil/tx_template.c:724-749: out[k] = sum_j in[j] exp(-+2 pi i jk / n), unscaled, either direction),
 * ff_tx_mdct_fwd / ff_tx_mdct_inv (tx_template.c:1268-1342: fold / pre-twiddle, an n = len/2 point FFT, post-twiddle).
 *
 * k_fft_z / k_mdct_z walk the reference's split-radix network level by level through LDS (log2 n round trips of the whole array,
 * butterfly lists and cosine tables fetched from LDS beside the data) and spend two thirds of their wave cycles waiting on LDS
 * (profiles/r03_fft1024_pmc.txt).  Here one wave owns a transform, every lane holds P = n / 64 points, and a pass is a radix-4 / -8 /
 * -16 butterfly computed in registers (Stockham autosort: pass inputs are elements j + t n/R, outputs go to
 * (j / Ns) Ns R + j % Ns + t Ns): 1024 points = 16 x 16 x 4, i.e. TWO trips through LDS, with the inter-pass twiddles kept in
 * registers across the transforms a wave serves.  The first pass reads global memory and the last one writes it, both as
 * coalesced 8-byte accesses (element lane + 64 s in slot s on both sides).
 *
 * The MDCT's fold and post-twiddle pair point k with point n - 1 - k: that is lane 63 - l, slot P - 1 - s of the same wave, so
 * the halves are exchanged with ds_bpermute (no memory) and every input byte is loaded once, every output float2 stored once.
 *
 * Numerics: a different factorisation of the same DFT than the reference's split-radix, with fused multiply-adds; results agree
 * with the reference within the stated tolerance of the float transforms (2^-18 of the transform's largest output,
 * tests/test_gpu_tx.py), not bit for bit.  FFHIP_TX_RADIX=0 selects the split-radix kernels.
 */
#include "common.h"
#include "tx_kernels.h"
#include "tx_radix_core.h"

#pragma clang fp contract(fast)

namespace {

constexpr int FR_WAVES = 4;
__host__ __device__ constexpr size_t fr_z_bytes(int n) { return ((size_t)FR_PAD(n) * 8 + 15) & ~(size_t)15; }


/* ---- 2048 .. 16384 points: a TEAM of T = N / 16 threads (two waves .. the whole 1024-thread workgroup) per transform, 16 points per
 * thread, the work array in the workgroup's LDS with barriers between the passes (16384 = 16 x 16 x 16 x 4: three trips).  The
 * inter-pass twiddles cannot stay in registers at 1024 threads (128 VGPRs each); a butterfly loads W^1, W^2, W^4, W^8 of its
 * (j % Ns) from the table and forms the other eleven by one to three multiplications (error <= 3 ulp of a twiddle). ---- */
template <int T>
__device__ __forceinline__ void frw_sync()
{
    if (T > 64)
        __syncthreads();
    else
        fr_sync();
}

template <int N, int T, int INV, int R, int NS, bool FIRST, bool LAST>
__device__ __forceinline__ void frw_pass(c32 (&v)[N / T], const c32 *wtab, c32 *z, int tid)
{
    constexpr int P = N / T, B = P / R;
#pragma unroll
    for (int b = 0; b < B; b++) {
        c32 a[R];
#pragma unroll
        for (int t = 0; t < R; t++)
            a[t] = v[b + B * t];
        const int j = tid + T * b;
        if (!FIRST) {
            const int k1 = (j & (NS - 1)) * (N / (NS * R));
            auto ld = [&](int k) {
                c32 x = wtab[k];
                if (INV)
                    x.y = -x.y;
                return x;
            };
            c32 w[R];
            w[1] = ld(k1);
            w[2] = ld(2 * k1);
            w[3] = cmul(w[1], w[2]);
            if constexpr (R > 4) {
                w[4] = ld(4 * k1);
#pragma unroll
                for (int t = 5; t < 8; t++)
                    w[t] = cmul(w[4], w[t - 4]);
            }
            if constexpr (R > 8) {
                w[8] = ld(8 * k1);
#pragma unroll
                for (int t = 9; t < 16; t++)
                    w[t] = cmul(w[8], w[t - 8]);
            }
#pragma unroll
            for (int t = 1; t < R; t++)
                a[t] = cmul(a[t], w[t]);
        }
        dft<INV, R>(a);
        if (LAST) {
#pragma unroll
            for (int k = 0; k < R; k++)
                v[b + B * k] = a[k];
        } else {
            const int base = (j / NS) * (NS * R) + (j & (NS - 1));
            c32 *zb = z + FR_PAD(base);
#pragma unroll
            for (int k = 0; k < R; k++)
                zb[k * NS + ((NS * R) % 32 == 0 ? (k * NS) >> 5 : 0)] = a[k]; /* as in fr_pass */
        }
    }
    if (!LAST) {
        frw_sync<T>();
        const c32 *zl = z + FR_PAD(tid);
#pragma unroll
        for (int s = 0; s < P; s++)
            v[s] = zl[(T + T / 32) * s]; /* FR_PAD(tid + T s) */
        frw_sync<T>();
    }
}

template <int LG> struct FrwPlan;
template <> struct FrwPlan<11> { static constexpr int NP = 3; static constexpr int R[4] = { 16, 16, 8, 1 }; };
template <> struct FrwPlan<12> { static constexpr int NP = 3; static constexpr int R[4] = { 16, 16, 16, 1 }; };
template <> struct FrwPlan<13> { static constexpr int NP = 4; static constexpr int R[4] = { 16, 16, 8, 4 }; };
template <> struct FrwPlan<14> { static constexpr int NP = 4; static constexpr int R[4] = { 16, 16, 16, 4 }; };

template <int LG, int INV>
__global__ __launch_bounds__((1 << LG) / 16) void k_fft_rw(const c32 *wtab, const float *in, size_t in_pitch, float *out, size_t out_pitch, int nt)
{
    constexpr int N = 1 << LG, T = N / 16, P = 16;
    using PL = FrwPlan<LG>;
    constexpr int R0 = PL::R[0], R1 = PL::R[1], R2 = PL::R[2], R3 = PL::R[3];
    extern __shared__ __align__(16) uint8_t lds_dyn[];
    c32 *z = reinterpret_cast<c32 *>(lds_dyn);
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const c32 *in2 = reinterpret_cast<const c32 *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
        c32 *out2 = reinterpret_cast<c32 *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
        c32 v[P];
#pragma unroll
        for (int s = 0; s < P; s++)
            v[s] = in2[tid + T * s];
        frw_pass<N, T, INV, R0, 1, true, false>(v, wtab, z, tid);
        frw_pass<N, T, INV, R1, R0, false, false>(v, wtab, z, tid);
        if constexpr (PL::NP == 3) {
            frw_pass<N, T, INV, R2, R0 * R1, false, true>(v, wtab, z, tid);
        } else {
            frw_pass<N, T, INV, R2, R0 * R1, false, false>(v, wtab, z, tid);
            frw_pass<N, T, INV, R3, R0 * R1 * R2, false, true>(v, wtab, z, tid);
        }
#pragma unroll
        for (int s = 0; s < P; s++)
            out2[tid + T * s] = v[s];
    }
}

template <int LG, int INV>
int frw_go(const c32 *wtab, const float *in, size_t in_pitch, float *out, size_t out_pitch, int nt, hipStream_t stream)
{
    constexpr int N = 1 << LG, T = N / 16;
    const size_t lds = fr_z_bytes(N);
    static FFHipPerDeviceOnce attr;
    if (attr.enter()) {
        (void)hipFuncSetAttribute((const void *)k_fft_rw<LG, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr.leave(true);
    }
    int cus = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        cus = prop.multiProcessorCount;
    int per_cu = (int)((160 * 1024) / (((lds + 1279) / 1280) * 1280));
    if (per_cu * (T / 64) > 32) per_cu = 32 / (T / 64);
    if (per_cu < 1) per_cu = 1;
    const int blocks = nt < cus * per_cu ? nt : cus * per_cu;
    hipLaunchKernelGGL((k_fft_rw<LG, INV>), dim3(blocks), dim3(T), lds, stream, wtab, in, in_pitch, out, out_pitch, nt);
    LAUNCH_CHECK();
    return 0;
}

template <int LG, int INV>
__global__ __launch_bounds__(64 * FR_WAVES) void k_fft_r(const c32 *wtab, const float *in, size_t in_pitch, float *out, size_t out_pitch,
                                                         int nt, int waves_total)
{
    constexpr int N = 1 << LG, P = N / 64;
    __shared__ __align__(16) uint8_t lds[FR_WAVES * fr_z_bytes(N)];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    c32 *z = reinterpret_cast<c32 *>(lds + wave * fr_z_bytes(N));
    FrTw<LG> tw;
    fr_load_all_tw<LG, INV>(tw, wtab, lane);
    for (int t = blockIdx.x * FR_WAVES + wave; t < nt; t += waves_total) {
        const c32 *in2 = reinterpret_cast<const c32 *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
        c32 *out2 = reinterpret_cast<c32 *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
        c32 v[P];
#pragma unroll
        for (int s = 0; s < P; s++)
            v[s] = in2[lane + 64 * s];
        fr_core<LG, INV>(v, tw, z, lane);
#pragma unroll
        for (int s = 0; s < P; s++)
            out2[lane + 64 * s] = v[s];
    }
}

/* lane l receives what lane 63 - l holds */
__device__ __forceinline__ float fr_mirror(float x, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((63 - lane) * 4, __builtin_bit_cast(int, x)));
}

/*
 * The MDCT around the same core (n = len / 2 complex points; contiguous rows).  Point k = lane + 64 s and its partner n - 1 - k
 * (lane 63 - l, slot P - 1 - s) — formulas as k_mdct_z (tx_api.hip), which states them per pair:
 *   forward fold   (tx_template.c:1285-1296): lanes compute w[i] and w[n-1-i] for their slots s < P / 2 from four float2 loads
 *                  and hand w[n-1-i] to the partner;
 *   forward post   (:1300-1310): out2[k] = (z[k].re e[k].re + z[k].im e[k].im,  Q[n-1-k]),  Q[k] = z[k].re e[k].im - z[k].im e[k].re;
 *   inverse pre    (:1321-1328): z[k] = (G.y e.re - F.x e.im,  G.y e.im + F.x e.re),  F = in2[k], G = in2[n-1-k];
 *   inverse post   (:1332-1341): out2[k] = (z[k].im e[k].im - z[k].re e[k].re,  Q'[n-1-k]),  Q'[k] = z[k].im e[k].re + z[k].re e[k].im.
 */
template <int LG, int INV>
__global__ __launch_bounds__(64 * FR_WAVES) void k_mdct_r(const c32 *wtab, const c32 *exptab, const float *in, size_t in_pitch, float *out,
                                                          size_t out_pitch, int nt, int waves_total)
{
    constexpr int N = 1 << LG, P = N / 64, Q = N / 2;
    __shared__ __align__(16) uint8_t lds[FR_WAVES * fr_z_bytes(N)];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    c32 *z = reinterpret_cast<c32 *>(lds + wave * fr_z_bytes(N));
    FrTw<LG> tw;
    fr_load_all_tw<LG, INV>(tw, wtab, lane); /* the sub-transform takes the MDCT's direction (ff_tx_mdct_init, tx_template.c:1240) */
    for (int t = blockIdx.x * FR_WAVES + wave; t < nt; t += waves_total) {
        const c32 *in2 = reinterpret_cast<const c32 *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
        c32 *out2 = reinterpret_cast<c32 *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
        c32 v[P];
        if (!INV) {
#pragma unroll
            for (int s = 0; s < P / 2; s++) {
                const int i = lane + 64 * s, j = N - 1 - i;
                const c32 p1 = in2[Q + i], p2 = in2[Q - 1 - i], p3 = in2[3 * Q + i], p4 = in2[3 * Q - 1 - i];
                const c32 e0 = exptab[i], e1 = exptab[j];
                const float re0 = -p1.x + p2.y, im0 = -p3.x - p4.y;
                const float re1 = -p4.x - p3.y, im1 = p2.x - p1.y;
                v[s] = make_float2(re0 * e0.y + im0 * e0.x, re0 * e0.x - im0 * e0.y);
                const c32 wj = make_float2(re1 * e1.y + im1 * e1.x, re1 * e1.x - im1 * e1.y);
                v[P - 1 - s] = make_float2(fr_mirror(wj.x, lane), fr_mirror(wj.y, lane));
            }
        } else {
#pragma unroll
            for (int s = 0; s < P; s++)
                v[s] = in2[lane + 64 * s];
            float gy[P];
#pragma unroll
            for (int s = 0; s < P; s++)
                gy[P - 1 - s] = fr_mirror(v[s].y, lane);
#pragma unroll
            for (int s = 0; s < P; s++) {
                const c32 e = exptab[lane + 64 * s];
                const float fx = v[s].x;
                v[s] = make_float2(gy[s] * e.x - fx * e.y, gy[s] * e.y + fx * e.x);
            }
        }
        fr_core<LG, INV>(v, tw, z, lane);
        float q[P];
#pragma unroll
        for (int s = 0; s < P; s++) {
            const c32 e = exptab[lane + 64 * s], zz = v[s];
            if (!INV) {
                q[P - 1 - s] = fr_mirror(zz.x * e.y - zz.y * e.x, lane);
                v[s].x = zz.x * e.x + zz.y * e.y;
            } else {
                q[P - 1 - s] = fr_mirror(zz.y * e.x + zz.x * e.y, lane);
                v[s].x = zz.y * e.y - zz.x * e.x;
            }
        }
#pragma unroll
        for (int s = 0; s < P; s++)
            out2[lane + 64 * s] = make_float2(v[s].x, q[s]);
    }
}

int fr_blocks(int nt)
{
    int cus = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        cus = prop.multiProcessorCount;
    const int want = (nt + FR_WAVES - 1) / FR_WAVES, cap = cus * 4;
    return want < cap ? want : cap;
}

} // namespace

bool ffhip_tx_radix_ok(int n)
{
    return n == 256 || n == 512 || n == 1024;
}

bool ffhip_tx_radix_fft_ok(int n)
{
    return ffhip_tx_radix_ok(n) || n == 2048 || n == 4096 || n == 8192 || n == 16384;
}

int ffhip_launch_fft_r(int n, int inv, const float2 *wtab, const float *in, size_t in_pitch, float *out, size_t out_pitch, int nt,
                       hipStream_t stream)
{
    const int blocks = fr_blocks(nt);
#define FR_GO(LG_, INV_)                                                                                                              \
    hipLaunchKernelGGL((k_fft_r<LG_, INV_>), dim3(blocks), dim3(64 * FR_WAVES), 0, stream, wtab, in, in_pitch, out, out_pitch, nt,   \
                       blocks * FR_WAVES)
    switch (n) {
    case 256:  if (inv) FR_GO(8, 1); else FR_GO(8, 0); break;
    case 512:  if (inv) FR_GO(9, 1); else FR_GO(9, 0); break;
    case 1024: if (inv) FR_GO(10, 1); else FR_GO(10, 0); break;
    case 2048:  return inv ? frw_go<11, 1>(wtab, in, in_pitch, out, out_pitch, nt, stream) : frw_go<11, 0>(wtab, in, in_pitch, out, out_pitch, nt, stream);
    case 4096:  return inv ? frw_go<12, 1>(wtab, in, in_pitch, out, out_pitch, nt, stream) : frw_go<12, 0>(wtab, in, in_pitch, out, out_pitch, nt, stream);
    case 8192:  return inv ? frw_go<13, 1>(wtab, in, in_pitch, out, out_pitch, nt, stream) : frw_go<13, 0>(wtab, in, in_pitch, out, out_pitch, nt, stream);
    case 16384: return inv ? frw_go<14, 1>(wtab, in, in_pitch, out, out_pitch, nt, stream) : frw_go<14, 0>(wtab, in, in_pitch, out, out_pitch, nt, stream);
    default: return FFHIP_EINVAL;
    }
#undef FR_GO
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_mdct_r(int n, int inv, const float2 *wtab, const float2 *exptab, const float *in, size_t in_pitch, float *out,
                        size_t out_pitch, int nt, hipStream_t stream)
{
    const int blocks = fr_blocks(nt);
#define FR_GO(LG_, INV_)                                                                                                              \
    hipLaunchKernelGGL((k_mdct_r<LG_, INV_>), dim3(blocks), dim3(64 * FR_WAVES), 0, stream, wtab, exptab, in, in_pitch, out,         \
                       out_pitch, nt, blocks * FR_WAVES)
    switch (n) {
    case 256:  if (inv) FR_GO(8, 1); else FR_GO(8, 0); break;
    case 512:  if (inv) FR_GO(9, 1); else FR_GO(9, 0); break;
    case 1024: if (inv) FR_GO(10, 1); else FR_GO(10, 0); break;
    default: return FFHIP_EINVAL;
    }
#undef FR_GO
    LAUNCH_CHECK();
    return 0;
}
