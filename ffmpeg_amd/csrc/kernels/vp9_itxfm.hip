/*
 * vp9_itxfm.hip — VP9 inverse transforms + add, 8 bits, batched: VP9DSPContext.itxfm_add[tx][txtp]
 * (libavcodec/vp9dsp_template.c:1155-1776): DCT / ADST in either pass for 4x4, 8x8, 16x16, DCT 32x32, the lossless 4x4 WHT,
 * and the dc-only shortcut of DCT_DCT.
 *
 * Every multiplication of the reference's butterfly network is followed by its own rounding ((x + 2^13) >> 14) and
 * intermediates wrap in 32 bits, so the network is the definition; it is built here from its structure — an N-point inverse
 * DCT is the N/2-point one on the even inputs plus an odd part of plane rotations, out[i] = E[i] + O[i], out[N-1-i] = E[i] -
 * O[i] — as templates that unroll completely into registers.
 *
 * GPU shape (as k_hevc_idct): a transform unit is N lanes, 64 / N units per wave, the block staged in wave-private LDS.  Lane i
 * transforms column i in place; the reference's second pass on "tmp + i" then is row i of that matrix, and its outputs belong
 * to picture column i, so they are written back transposed; finally lane i adds row i of the residual to the picture with
 * packed byte stores.  Values are stored as int16 between and after the passes, as the reference's dctcoef arrays do.
 */
#include "common.h"
#include "h264_kernels.h"

static_assert(sizeof(FFHipVp9TU) == 12, "FFHipVp9TU is a 12-byte record");

__device__ __forceinline__ int vp_r14(uint32_t x) { return (int)(x + (1u << 13)) >> 14; }
/* (a c - b s, a s + b c), each rounded */
__device__ __forceinline__ void vp_rot(int a, int b, uint32_t c, uint32_t s, int &lo, int &hi)
{
    lo = vp_r14((uint32_t)a * c - (uint32_t)b * s);
    hi = vp_r14((uint32_t)a * s + (uint32_t)b * c);
}
/* (-(a s + b c), a c - b s): the negation happens before the rounding */
__device__ __forceinline__ void vp_nrot(int a, int b, uint32_t c, uint32_t s, int &lo, int &hi)
{
    lo = vp_r14(0u - ((uint32_t)a * s + (uint32_t)b * c));
    hi = vp_r14((uint32_t)a * c - (uint32_t)b * s);
}
/* ((a - b), (a + b)) / sqrt 2 */
__device__ __forceinline__ void vp_half(int a, int b, int &lo, int &hi)
{
    const int l = vp_r14(((uint32_t)a - (uint32_t)b) * 11585u), h = vp_r14(((uint32_t)a + (uint32_t)b) * 11585u);
    lo = l;
    hi = h;
}

/* odd parts: x = the odd-indexed inputs in order, o[] ordered so that out[i] = e[i] + o[i] */
__device__ __forceinline__ void vp_odd(const int (&x)[2], int (&o)[2]) /* idct4: inputs 1, 3 */
{
    vp_rot(x[0], x[1], 6270, 15137, o[1], o[0]);
}
__device__ __forceinline__ void vp_odd(const int (&x)[4], int (&o)[4]) /* idct8: inputs 1, 3, 5, 7 */
{
    int a4, a7, a5, a6;
    vp_rot(x[0], x[3], 3196, 16069, a4, a7);
    vp_rot(x[2], x[1], 13623, 9102, a5, a6);
    o[3] = a4 + a5;
    o[0] = a7 + a6;
    vp_half(a7 - a6, a4 - a5, o[2], o[1]);
}
__device__ __forceinline__ void vp_odd(const int (&x)[8], int (&o)[8]) /* idct16: inputs 1, 3, ..., 15 */
{
    int a[8], t[8];
    vp_rot(x[0], x[7], 1606, 16305, a[0], a[7]);
    vp_rot(x[4], x[3], 12665, 10394, a[1], a[6]);
    vp_rot(x[2], x[5], 7723, 14449, a[2], a[5]);
    vp_rot(x[6], x[1], 15679, 4756, a[3], a[4]);
    t[0] = a[0] + a[1]; t[1] = a[0] - a[1]; t[2] = a[3] - a[2]; t[3] = a[3] + a[2];
    t[4] = a[4] + a[5]; t[5] = a[4] - a[5]; t[6] = a[7] - a[6]; t[7] = a[7] + a[6];
    vp_rot(t[6], t[1], 6270, 15137, a[1], a[6]);
    vp_nrot(t[5], t[2], 6270, 15137, a[2], a[5]);
    a[0] = t[0] + t[3]; a[3] = t[0] - t[3];
    t[1] = a[1] + a[2]; t[2] = a[1] - a[2];
    a[4] = t[7] - t[4]; a[7] = t[7] + t[4];
    t[5] = a[6] - a[5]; t[6] = a[6] + a[5];
    vp_half(t[5], t[2], a[2], a[5]);
    vp_half(a[4], a[3], t[3], t[4]);
    o[0] = a[7]; o[1] = t[6]; o[2] = a[5]; o[3] = t[4]; o[4] = t[3]; o[5] = a[2]; o[6] = t[1]; o[7] = a[0];
}
__device__ __forceinline__ void vp_odd(const int (&x)[16], int (&o)[16]) /* idct32: inputs 1, 3, ..., 31 */
{
    int a[16], t[16];
    vp_rot(x[0], x[15], 804, 16364, a[0], a[15]);
    vp_rot(x[8], x[7], 12140, 11003, a[1], a[14]);
    vp_rot(x[4], x[11], 7005, 14811, a[2], a[13]);
    vp_rot(x[12], x[3], 15426, 5520, a[3], a[12]);
    vp_rot(x[2], x[13], 3981, 15893, a[4], a[11]);
    vp_rot(x[10], x[5], 14053, 8423, a[5], a[10]);
    vp_rot(x[6], x[9], 9760, 13160, a[6], a[9]);
    vp_rot(x[14], x[1], 16207, 2404, a[7], a[8]);
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
        t[k] = a[k] + a[k + 1];
        t[k + 1] = a[k] - a[k + 1];
        t[k + 2] = a[k + 3] - a[k + 2];
        t[k + 3] = a[k + 3] + a[k + 2];
    }
    vp_rot(t[14], t[1], 3196, 16069, a[1], a[14]);
    vp_nrot(t[13], t[2], 3196, 16069, a[2], a[13]);
    vp_rot(t[10], t[5], 13623, 9102, a[5], a[10]);
    vp_nrot(t[9], t[6], 13623, 9102, a[6], a[9]);
    a[0] = t[0] + t[3];    a[3] = t[0] - t[3];
    t[1] = a[1] + a[2];    t[2] = a[1] - a[2];
    a[4] = t[7] - t[4];    a[7] = t[7] + t[4];
    t[5] = a[6] - a[5];    t[6] = a[6] + a[5];
    a[8] = t[8] + t[11];   a[11] = t[8] - t[11];
    t[9] = a[9] + a[10];   t[10] = a[9] - a[10];
    a[12] = t[15] - t[12]; a[15] = t[15] + t[12];
    t[13] = a[14] - a[13]; t[14] = a[14] + a[13];
    vp_rot(t[13], t[2], 6270, 15137, a[2], a[13]);
    vp_rot(a[12], a[3], 6270, 15137, t[3], t[12]);
    vp_nrot(a[11], a[4], 6270, 15137, t[4], t[11]);
    vp_nrot(t[10], t[5], 6270, 15137, a[5], a[10]);
    int n[16];
    n[0] = a[0] + a[7];    n[7] = a[0] - a[7];
    n[1] = t[1] + t[6];    n[6] = t[1] - t[6];
    n[2] = a[2] + a[5];    n[5] = a[2] - a[5];
    n[3] = t[3] + t[4];    n[4] = t[3] - t[4];
    n[8] = a[15] - a[8];   n[15] = a[15] + a[8];
    n[9] = t[14] - t[9];   n[14] = t[14] + t[9];
    n[10] = a[13] - a[10]; n[13] = a[13] + a[10];
    n[11] = t[12] - t[11]; n[12] = t[12] + t[11];
    vp_half(n[11], n[4], n[4], n[11]);
    vp_half(n[10], n[5], n[5], n[10]);
    vp_half(n[9], n[6], n[6], n[9]);
    vp_half(n[8], n[7], n[7], n[8]);
#pragma unroll
    for (int k = 0; k < 16; k++)
        o[k] = n[15 - k];
}

template <int N>
__device__ __forceinline__ void vp_idct(const int (&x)[N], int (&out)[N])
{
    int ev[N / 2], od[N / 2], e[N / 2], o[N / 2];
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
        ev[k] = x[2 * k];
        od[k] = x[2 * k + 1];
    }
    if constexpr (N == 4)
        vp_half(ev[0], ev[1], e[1], e[0]);
    else
        vp_idct<N / 2>(ev, e);
    vp_odd(od, o);
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
        out[k] = e[k] + o[k];
        out[N - 1 - k] = e[k] - o[k];
    }
}

__device__ __forceinline__ void vp_iadst(const int (&x)[4], int (&out)[4])
{
    const uint32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    const uint32_t t0 = 5283u * x0 + 15212u * x2 + 9929u * x3, t1 = 9929u * x0 - 5283u * x2 - 15212u * x3;
    const uint32_t t2 = 13377u * (x0 - x2 + x3), t3 = 13377u * x1;
    out[0] = vp_r14(t0 + t3);
    out[1] = vp_r14(t1 + t3);
    out[2] = vp_r14(t2);
    out[3] = vp_r14(t0 + t1 - t3);
}
__device__ __forceinline__ void vp_iadst(const int (&x)[8], int (&out)[8])
{
    constexpr uint32_t c[4][2] = { { 16305, 1606 }, { 14449, 7723 }, { 10394, 12665 }, { 4756, 15679 } };
    uint32_t p[8];
    int t[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t a = x[7 - 2 * k], b = x[2 * k];
        p[2 * k] = c[k][0] * a + c[k][1] * b;
        p[2 * k + 1] = c[k][1] * a - c[k][0] * b;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        t[k] = vp_r14(p[k] + p[k + 4]);
        t[k + 4] = vp_r14(p[k] - p[k + 4]);
    }
    const uint32_t q4 = 15137u * (uint32_t)t[4] + 6270u * (uint32_t)t[5], q5 = 6270u * (uint32_t)t[4] - 15137u * (uint32_t)t[5];
    const uint32_t q6 = 15137u * (uint32_t)t[7] - 6270u * (uint32_t)t[6], q7 = 6270u * (uint32_t)t[7] + 15137u * (uint32_t)t[6];
    const int s2 = t[0] - t[2], s3 = t[1] - t[3], s6 = vp_r14(q4 - q6), s7 = vp_r14(q5 - q7);
    out[0] = t[0] + t[2];
    out[7] = -(t[1] + t[3]);
    out[1] = -vp_r14(q4 + q6);
    out[6] = vp_r14(q5 + q7);
    out[3] = -vp_r14(((uint32_t)s2 + (uint32_t)s3) * 11585u);
    out[4] = vp_r14(((uint32_t)s2 - (uint32_t)s3) * 11585u);
    out[2] = vp_r14(((uint32_t)s6 + (uint32_t)s7) * 11585u);
    out[5] = -vp_r14(((uint32_t)s6 - (uint32_t)s7) * 11585u);
}
__device__ __forceinline__ void vp_iadst(const int (&x)[16], int (&out)[16])
{
    constexpr uint32_t c[8][2] = { { 16364, 804 }, { 15893, 3981 }, { 14811, 7005 }, { 13160, 9760 },
                                   { 11003, 12140 }, { 8423, 14053 }, { 5520, 15426 }, { 2404, 16207 } };
    uint32_t p[16], q[16];
    int a[16], t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t u = x[15 - 2 * k], v = x[2 * k];
        p[2 * k] = c[k][0] * u + c[k][1] * v;
        p[2 * k + 1] = c[k][1] * u - c[k][0] * v;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        a[k] = vp_r14(p[k] + p[k + 8]);
        a[k + 8] = vp_r14(p[k] - p[k + 8]);
    }
    q[8] = (uint32_t)a[8] * 16069u + (uint32_t)a[9] * 3196u;
    q[9] = (uint32_t)a[8] * 3196u - (uint32_t)a[9] * 16069u;
    q[10] = (uint32_t)a[10] * 9102u + (uint32_t)a[11] * 13623u;
    q[11] = (uint32_t)a[10] * 13623u - (uint32_t)a[11] * 9102u;
    q[12] = (uint32_t)a[13] * 16069u - (uint32_t)a[12] * 3196u;
    q[13] = (uint32_t)a[13] * 3196u + (uint32_t)a[12] * 16069u;
    q[14] = (uint32_t)a[15] * 9102u - (uint32_t)a[14] * 13623u;
    q[15] = (uint32_t)a[15] * 13623u + (uint32_t)a[14] * 9102u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        t[k] = a[k] + a[k + 4];
        t[k + 4] = a[k] - a[k + 4];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        a[k + 8] = vp_r14(q[k + 8] + q[k + 12]);
        a[k + 12] = vp_r14(q[k + 8] - q[k + 12]);
    }
    const uint32_t r4 = (uint32_t)t[4] * 15137u + (uint32_t)t[5] * 6270u, r5 = (uint32_t)t[4] * 6270u - (uint32_t)t[5] * 15137u;
    const uint32_t r6 = (uint32_t)t[7] * 15137u - (uint32_t)t[6] * 6270u, r7 = (uint32_t)t[7] * 6270u + (uint32_t)t[6] * 15137u;
    const uint32_t r12 = (uint32_t)a[12] * 15137u + (uint32_t)a[13] * 6270u, r13 = (uint32_t)a[12] * 6270u - (uint32_t)a[13] * 15137u;
    const uint32_t r14 = (uint32_t)a[15] * 15137u - (uint32_t)a[14] * 6270u, r15 = (uint32_t)a[15] * 6270u + (uint32_t)a[14] * 15137u;
    const int s2 = t[0] - t[2], s3 = t[1] - t[3], s6 = vp_r14(r4 - r6), s7 = vp_r14(r5 - r7);
    const int s10 = a[8] - a[10], s11 = a[9] - a[11], s14 = vp_r14(r12 - r14), s15 = vp_r14(r13 - r15);
    out[0] = t[0] + t[2];
    out[15] = -(t[1] + t[3]);
    out[3] = -vp_r14(r4 + r6);
    out[12] = vp_r14(r5 + r7);
    out[1] = -(a[8] + a[10]);
    out[14] = a[9] + a[11];
    out[2] = vp_r14(r12 + r14);
    out[13] = -vp_r14(r13 + r15);
    out[7] = vp_r14((0u - ((uint32_t)s2 + (uint32_t)s3)) * 11585u);
    out[8] = vp_r14(((uint32_t)s2 - (uint32_t)s3) * 11585u);
    out[4] = vp_r14(((uint32_t)s7 + (uint32_t)s6) * 11585u);
    out[11] = vp_r14(((uint32_t)s7 - (uint32_t)s6) * 11585u);
    out[6] = vp_r14(((uint32_t)s11 + (uint32_t)s10) * 11585u);
    out[9] = vp_r14(((uint32_t)s11 - (uint32_t)s10) * 11585u);
    out[5] = vp_r14((0u - ((uint32_t)s14 + (uint32_t)s15)) * 11585u);
    out[10] = vp_r14(((uint32_t)s14 - (uint32_t)s15) * 11585u);
}
/* 32x32 has no ADST */
__device__ __forceinline__ void vp_iadst(const int (&x)[32], int (&out)[32]) { vp_idct<32>(x, out); }

/* lossless mode: the Walsh-Hadamard transform; pass 0 scales its inputs down by 4 */
__device__ __forceinline__ void vp_iwht(const int (&x)[4], int (&out)[4], bool first)
{
    int t0 = x[0], t1 = x[3], t2 = x[1], t3 = x[2];
    if (first) {
        t0 >>= 2; t1 >>= 2; t2 >>= 2; t3 >>= 2;
    }
    t0 += t2;
    t3 -= t1;
    const int t4 = (t0 - t3) >> 1;
    t1 = t4 - t1;
    t2 = t4 - t2;
    t0 -= t1;
    t3 += t2;
    out[0] = t0; out[1] = t1; out[2] = t2; out[3] = t3;
}

__device__ __forceinline__ void vp_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int LOG2, bool WHT>
__global__ __launch_bounds__(256) void k_vp9_itxfm(int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n)
{
    constexpr int N = 1 << LOG2, UPW = 64 / N, BITS = WHT ? 0 : LOG2 == 2 ? 4 : LOG2 == 3 ? 5 : 6;
    __shared__ __align__(16) int16_t lds[4][64 * N];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int u0 = (blockIdx.x * 4 + wave) * UPW;
    if (u0 >= n)
        return;
    int16_t *blk = lds[wave];
    const int ul = lane / N, i = lane % N;
    const int u = u0 + ul;
    const bool live = u < n;
    const FFHipVp9TU tu = tus[live ? u : u0];
    /* ---- stage the wave's units ---- */
    constexpr int DW = N * N / 2;
    for (int t = lane; t < UPW * DW; t += 64) {
        const int b = t / DW, w = t % DW;
        if (u0 + b < n)
            reinterpret_cast<uint32_t *>(blk)[t] = reinterpret_cast<const uint32_t *>(coeffs + tus[u0 + b].coeff_offset)[w];
    }
    vp_wave_sync();
    int16_t *mine = blk + ul * N * N;
    const bool adst1 = !WHT && LOG2 < 5 && (tu.txtp == 1 || tu.txtp == 3), adst2 = !WHT && LOG2 < 5 && (tu.txtp == 2 || tu.txtp == 3);
    const bool dc_only = !WHT && tu.dc_only && !adst1 && !adst2;
    int dcv = 0;
    if (dc_only)
        dcv = vp_r14((uint32_t)vp_r14((uint32_t)(int)mine[0] * 11585u) * 11585u);
    int x[N], o[N];
    /* first pass: column i, in place */
    if (!dc_only) {
#pragma unroll
        for (int k = 0; k < N; k++)
            x[k] = mine[k * N + i];
        if constexpr (WHT)
            vp_iwht(x, o, true);
        else if (adst1)
            vp_iadst(x, o);
        else
            vp_idct<N>(x, o);
    }
    vp_wave_sync();
    if (!dc_only) {
#pragma unroll
        for (int k = 0; k < N; k++)
            mine[k * N + i] = (int16_t)o[k];
    }
    vp_wave_sync();
    /* second pass: row i of that matrix; its outputs are picture column i, so they go back transposed */
    if (!dc_only) {
#pragma unroll
        for (int k = 0; k < N; k++)
            x[k] = mine[i * N + k];
        if constexpr (WHT)
            vp_iwht(x, o, false);
        else if (adst2)
            vp_iadst(x, o);
        else
            vp_idct<N>(x, o);
    }
    vp_wave_sync();
#pragma unroll
    for (int k = 0; k < N; k++)
        mine[k * N + i] = dc_only ? (int16_t)dcv : (int16_t)o[k];
    vp_wave_sync();
    /* ---- picture += (residual + round) >> bits, row i of my unit ---- */
    if (live) {
        uint8_t *d = dst + tu.dst_offset + (ptrdiff_t)i * stride;
        const int16_t *r = mine + i * N;
        auto res = [&](int v) { return BITS ? (int)((uint32_t)v + (1u << (BITS ? BITS - 1 : 0))) >> BITS : v; };
#pragma unroll
        for (int c = 0; c < N; c += 4) {
            if (!(((uintptr_t)d) & 3)) {
                const uint32_t p = *reinterpret_cast<const uint32_t *>(d + c);
                const uint32_t q = pack4(clip_u8((int)(p & 0xFF) + res(r[c])), clip_u8((int)((p >> 8) & 0xFF) + res(r[c + 1])),
                                         clip_u8((int)((p >> 16) & 0xFF) + res(r[c + 2])), clip_u8((int)(p >> 24) + res(r[c + 3])));
                *reinterpret_cast<uint32_t *>(d + c) = q;
            } else {
                for (int e = 0; e < 4; e++)
                    d[c + e] = (uint8_t)clip_u8((int)d[c + e] + res(r[c + e]));
            }
        }
    }
    /* ---- the block is consumed: zeroed (dc-only: block[0] alone, as the reference leaves the rest untouched) ---- */
    for (int t = lane; t < UPW * DW; t += 64) {
        const int b = t / DW, w = t % DW;
        if (u0 + b < n) {
            const FFHipVp9TU tb = tus[u0 + b];
            uint32_t *cw = reinterpret_cast<uint32_t *>(coeffs + tb.coeff_offset);
            const bool dcb = !WHT && tb.dc_only && (LOG2 == 5 || tb.txtp == 0);
            if (!dcb)
                cw[w] = 0;
            else if (w == 0)
                cw[0] &= 0xFFFF0000u;
        }
    }
}

int ffhip_launch_vp9_itxfm(int tx, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    const int log2 = tx == 4 ? 2 : tx + 2, upw = 64 >> log2;
    const dim3 grid(cdiv(n, 4 * upw)), block(256);
    switch (tx) {
    case 0: hipLaunchKernelGGL((k_vp9_itxfm<2, false>), grid, block, 0, stream, coeffs, dst, stride, tus, n); break;
    case 1: hipLaunchKernelGGL((k_vp9_itxfm<3, false>), grid, block, 0, stream, coeffs, dst, stride, tus, n); break;
    case 2: hipLaunchKernelGGL((k_vp9_itxfm<4, false>), grid, block, 0, stream, coeffs, dst, stride, tus, n); break;
    case 3: hipLaunchKernelGGL((k_vp9_itxfm<5, false>), grid, block, 0, stream, coeffs, dst, stride, tus, n); break;
    case 4: hipLaunchKernelGGL((k_vp9_itxfm<2, true>), grid, block, 0, stream, coeffs, dst, stride, tus, n); break;
    default:
        ffhip_set_error("ffhip_vp9_itxfm: tx %d outside 0..4", tx);
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}
