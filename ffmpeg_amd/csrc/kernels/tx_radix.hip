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

#pragma clang fp contract(fast)

namespace {

typedef float2 c32;

#define FR_PAD(i) ((i) + ((i) >> 5))

__device__ __forceinline__ c32 cadd(const c32 a, const c32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 csub(const c32 a, const c32 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c32 cmul(const c32 a, const c32 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
/* a * (cr -+ i ci): the forward transform's constants are exp(-i phi) */
template <int INV>
__device__ __forceinline__ c32 cmulc(const c32 a, const float cr, const float ci)
{
    return INV ? make_float2(a.x * cr - a.y * ci, a.x * ci + a.y * cr) : make_float2(a.x * cr + a.y * ci, a.y * cr - a.x * ci);
}
/* a * (-+i) */
template <int INV>
__device__ __forceinline__ c32 cmuli(const c32 a)
{
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

__device__ __forceinline__ void fr_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int INV>
__device__ __forceinline__ void dft4(c32 &a0, c32 &a1, c32 &a2, c32 &a3)
{
    const c32 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmuli<INV>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

template <int INV, int R>
__device__ __forceinline__ void dft(c32 (&a)[R])
{
    constexpr float H = 0.70710678118654752440f, C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
    if constexpr (R == 4) {
        dft4<INV>(a[0], a[1], a[2], a[3]);
    } else if constexpr (R == 8) {
        /* n = n0 + 2 n1, k = k1 + 4 k0: W8^(nk) = W2^(n0 k0) W8^(n0 k1) W4^(n1 k1) */
        dft4<INV>(a[0], a[2], a[4], a[6]);
        dft4<INV>(a[1], a[3], a[5], a[7]);
        const c32 o1 = cmulc<INV>(a[3], H, H), o2 = cmuli<INV>(a[5]), o3 = cmulc<INV>(a[7], -H, H);
        const c32 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6], o0 = a[1];
        a[0] = cadd(e0, o0); a[4] = csub(e0, o0);
        a[1] = cadd(e1, o1); a[5] = csub(e1, o1);
        a[2] = cadd(e2, o2); a[6] = csub(e2, o2);
        a[3] = cadd(e3, o3); a[7] = csub(e3, o3);
    } else {
        static_assert(R == 16, "radix");
        /* n = n0 + 4 n1, k = k1 + 4 k0: W16^(nk) = W4^(n0 k0) W16^(n0 k1) W4^(n1 k1) */
#pragma unroll
        for (int n0 = 0; n0 < 4; n0++)
            dft4<INV>(a[n0], a[n0 + 4], a[n0 + 8], a[n0 + 12]); /* a[n0 + 4 k1] = A[n0][k1] */
        a[1 + 4] = cmulc<INV>(a[1 + 4], C1, S1);   /* W16^1 */
        a[1 + 8] = cmulc<INV>(a[1 + 8], H, H);     /* W16^2 */
        a[1 + 12] = cmulc<INV>(a[1 + 12], S1, C1); /* W16^3 */
        a[2 + 4] = cmulc<INV>(a[2 + 4], H, H);     /* W16^2 */
        a[2 + 8] = cmuli<INV>(a[2 + 8]);           /* W16^4 */
        a[2 + 12] = cmulc<INV>(a[2 + 12], -H, H);  /* W16^6 */
        a[3 + 4] = cmulc<INV>(a[3 + 4], S1, C1);   /* W16^3 */
        a[3 + 8] = cmulc<INV>(a[3 + 8], -H, H);    /* W16^6 */
        a[3 + 12] = cmulc<INV>(a[3 + 12], -C1, -S1); /* W16^9 */
        c32 x[16];
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) {
            c32 b0 = a[4 * k1], b1 = a[1 + 4 * k1], b2 = a[2 + 4 * k1], b3 = a[3 + 4 * k1];
            dft4<INV>(b0, b1, b2, b3);
            x[k1] = b0; x[k1 + 4] = b1; x[k1 + 8] = b2; x[k1 + 12] = b3;
        }
#pragma unroll
        for (int k = 0; k < 16; k++)
            a[k] = x[k];
    }
}

/* the passes of an N-point transform on 64 lanes */
template <int LG> struct FrPlan;
template <> struct FrPlan<8>  { static constexpr int NP = 4; static constexpr int R[4] = { 4, 4, 4, 4 }; };
template <> struct FrPlan<9>  { static constexpr int NP = 3; static constexpr int R[4] = { 8, 8, 8, 1 }; };
template <> struct FrPlan<10> { static constexpr int NP = 3; static constexpr int R[4] = { 16, 16, 4, 1 }; };

template <int LG>
struct FrTw { /* the inter-pass twiddles of passes 1 .. NP-1: P / R butterflies x (R - 1) factors each */
    static constexpr int P = (1 << LG) / 64;
    static constexpr int cnt(int p) { return (P / FrPlan<LG>::R[p]) * (FrPlan<LG>::R[p] - 1); }
    c32 w1[cnt(1)];
    c32 w2[cnt(2)];
    c32 w3[FrPlan<LG>::NP > 3 ? cnt(3) : 1];
};

/* factor t of butterfly j = lane + 64 b in a pass of radix R behind Ns points: exp(-+2 pi i (j % Ns) t / (Ns R)) */
template <int LG, int INV, int R, int NS, int CNT>
__device__ __forceinline__ void fr_load_tw(c32 (&w)[CNT], const c32 *wtab, int lane)
{
    constexpr int N = 1 << LG, B = (N / 64) / R;
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int t = 1; t < R; t++) {
            const int k = ((lane + 64 * b) & (NS - 1)) * t * (N / (NS * R));
            c32 v = wtab[k];
            if (INV)
                v.y = -v.y;
            w[b * (R - 1) + t - 1] = v;
        }
}

template <int LG, int INV>
__device__ __forceinline__ void fr_load_all_tw(FrTw<LG> &tw, const c32 *wtab, int lane)
{
    using PL = FrPlan<LG>;
    fr_load_tw<LG, INV, PL::R[1], PL::R[0]>(tw.w1, wtab, lane);
    fr_load_tw<LG, INV, PL::R[2], PL::R[0] * PL::R[1]>(tw.w2, wtab, lane);
    if constexpr (PL::NP > 3)
        fr_load_tw<LG, INV, PL::R[3], PL::R[0] * PL::R[1] * PL::R[2]>(tw.w3, wtab, lane);
}

/* one pass: v[s] is element lane + 64 s of the pass's input on entry and of its output (the next pass's input) on return */
template <int LG, int INV, int R, int NS, bool FIRST, bool LAST, int CNT>
__device__ __forceinline__ void fr_pass(c32 (&v)[(1 << LG) / 64], const c32 (&w)[CNT], c32 *z, int lane)
{
    constexpr int N = 1 << LG, P = N / 64, B = P / R;
#pragma unroll
    for (int b = 0; b < B; b++) {
        c32 a[R];
#pragma unroll
        for (int t = 0; t < R; t++)
            a[t] = v[b + B * t];
        if (!FIRST) {
#pragma unroll
            for (int t = 1; t < R; t++)
                a[t] = cmul(a[t], w[b * (R - 1) + t - 1]);
        }
        dft<INV, R>(a);
        if (LAST) {
#pragma unroll
            for (int k = 0; k < R; k++)
                v[b + B * k] = a[k];
        } else {
            const int j = lane + 64 * b;
            const int base = (j / NS) * (NS * R) + (j & (NS - 1));
            /* the padding of base + k Ns is the padding of base plus a constant: base is a multiple of Ns R plus less than Ns, and
             * either 32 divides Ns R or Ns R divides 32 (the butterfly's outputs stay inside one 32-element row) */
            c32 *zb = z + FR_PAD(base);
#pragma unroll
            for (int k = 0; k < R; k++)
                zb[k * NS + ((NS * R) % 32 == 0 ? (k * NS) >> 5 : 0)] = a[k];
        }
    }
    if (!LAST) {
        fr_sync();
        const c32 *zl = z + FR_PAD(lane);
#pragma unroll
        for (int s = 0; s < P; s++)
            v[s] = zl[66 * s]; /* FR_PAD(lane + 64 s) */
        fr_sync();
    }
}

template <int LG, int INV>
__device__ __forceinline__ void fr_core(c32 (&v)[(1 << LG) / 64], const FrTw<LG> &tw, c32 *z, int lane)
{
    using PL = FrPlan<LG>;
    constexpr int R0 = PL::R[0], R1 = PL::R[1], R2 = PL::R[2], R3 = PL::R[3];
    const c32 none[1] = { make_float2(0.f, 0.f) };
    fr_pass<LG, INV, R0, 1, true, false>(v, none, z, lane);
    fr_pass<LG, INV, R1, R0, false, false>(v, tw.w1, z, lane);
    if constexpr (PL::NP == 3) {
        fr_pass<LG, INV, R2, R0 * R1, false, true>(v, tw.w2, z, lane);
    } else {
        fr_pass<LG, INV, R2, R0 * R1, false, false>(v, tw.w2, z, lane);
        fr_pass<LG, INV, R3, R0 * R1 * R2, false, true>(v, tw.w3, z, lane);
    }
}

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
