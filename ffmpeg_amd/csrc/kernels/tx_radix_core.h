/*
 * tx_radix_core.h — the register-resident radix-4 / -8 / -16 FFT core of kernels/tx_radix.hip (one wave per transform, n = 256, 512 or
 * 1024 complex points, P = n / 64 per lane), shared with the RDFT / DCT kernels of tx_api.hip.  See tx_radix.hip for the design.
 * The helpers that multiply-and-add carry their own `fp contract(fast)`: tx_api.hip is otherwise built without contraction.
 */
#ifndef FFHIP_TX_RADIX_CORE_H
#define FFHIP_TX_RADIX_CORE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float2 c32;

#define FR_PAD(i) ((i) + ((i) >> 5))

__device__ __forceinline__ c32 cadd(const c32 a, const c32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 csub(const c32 a, const c32 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c32 cmul(const c32 a, const c32 w)
{
#pragma clang fp contract(fast)
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}
/* a * (cr -+ i ci): the forward transform's constants are exp(-i phi) */
template <int INV>
__device__ __forceinline__ c32 cmulc(const c32 a, const float cr, const float ci)
{
#pragma clang fp contract(fast)
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


/* the same transform for a kernel that keeps its data in the wave's (padded: FR_PAD == TX_PAD) LDS work array in natural order */
template <int LG, int INV>
__device__ __forceinline__ void fr_fft_lds(c32 *z, const FrTw<LG> &tw, int lane)
{
    constexpr int P = (1 << LG) / 64;
    c32 v[P];
    fr_sync();
    const c32 *zl = z + FR_PAD(lane);
#pragma unroll
    for (int s = 0; s < P; s++)
        v[s] = zl[66 * s];
    fr_sync();
    fr_core<LG, INV>(v, tw, z, lane);
    c32 *zw = z + FR_PAD(lane);
#pragma unroll
    for (int s = 0; s < P; s++)
        zw[66 * s] = v[s];
    fr_sync();
}

#endif
