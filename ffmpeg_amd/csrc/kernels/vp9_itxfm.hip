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
#include <type_traits>

#include "common.h"
#include "h264_kernels.h"

static_assert(sizeof(FFHipVp9TU) == 12, "FFHipVp9TU is a 12-byte record");

namespace vp32 {
#define VP_ST int
#define VP_UT uint32_t
#include "vp9_itxfm_net.inc"
#undef VP_ST
#undef VP_UT
} // namespace vp32
namespace vp64 {
#define VP_ST long long
#define VP_UT unsigned long long
#include "vp9_itxfm_net.inc"
#undef VP_ST
#undef VP_UT
} // namespace vp64

__device__ __forceinline__ void vp_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* HBD: coefficients are int32 (dctcoef), the networks run in 64 bits (namespace vp64), samples are uint16_t clipped to (1 << bd) - 1 */
template <int LOG2, bool WHT, bool HBD>
__global__ __launch_bounds__(256) void k_vp9_itxfm(void *coeffs_, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n, int bd)
{
    using COEF = typename std::conditional<HBD, int32_t, int16_t>::type;
    using ST = typename std::conditional<HBD, long long, int>::type;
    using UT = typename std::conditional<HBD, unsigned long long, uint32_t>::type;
    constexpr int N = 1 << LOG2, UPW = 64 / N, BITS = WHT ? 0 : LOG2 == 2 ? 4 : LOG2 == 3 ? 5 : 6;
    __shared__ __align__(16) COEF lds[4][64 * N];
    COEF *coeffs = static_cast<COEF *>(coeffs_);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int u0 = (blockIdx.x * 4 + wave) * UPW;
    if (u0 >= n)
        return;
    COEF *blk = lds[wave];
    const int ul = lane / N, i = lane % N;
    const int u = u0 + ul;
    const bool live = u < n;
    const FFHipVp9TU tu = tus[live ? u : u0];
    /* ---- stage the wave's units: 16 bytes per lane and step where the block is 16-byte aligned, else as dwords ---- */
    constexpr int DW = N * N * (int)sizeof(COEF) / 4, Q4 = DW / 4;
    for (int t = lane; t < UPW * Q4; t += 64) {
        const int b = t / Q4, w = t % Q4;
        if (u0 + b < n) {
            const COEF *g = coeffs + tus[u0 + b].coeff_offset;
            if (!(reinterpret_cast<uintptr_t>(g) & 15)) {
                reinterpret_cast<uint4 *>(blk)[t] = reinterpret_cast<const uint4 *>(g)[w];
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    reinterpret_cast<uint32_t *>(blk)[4 * t + k] = reinterpret_cast<const uint32_t *>(g)[4 * w + k];
            }
        }
    }
    vp_wave_sync();
    COEF *mine = blk + ul * N * N;
    const bool adst1 = !WHT && LOG2 < 5 && (tu.txtp == 1 || tu.txtp == 3), adst2 = !WHT && LOG2 < 5 && (tu.txtp == 2 || tu.txtp == 3);
    const bool dc_only = !WHT && tu.dc_only && !adst1 && !adst2;
    auto r14 = [](UT x) { return (ST)(x + ((UT)1 << 13)) >> 14; };
    int dcv = 0;
    if (dc_only)
        dcv = (int)r14((UT)r14((UT)(ST)mine[0] * 11585u) * 11585u);
    ST x[N], o[N];
    auto run = [&](bool adst, bool first) {
        if constexpr (HBD) {
            if constexpr (WHT) vp64::vp_iwht(x, o, first);
            else if (adst) vp64::vp_iadst(x, o);
            else vp64::vp_idct<N>(x, o);
        } else {
            if constexpr (WHT) vp32::vp_iwht(x, o, first);
            else if (adst) vp32::vp_iadst(x, o);
            else vp32::vp_idct<N>(x, o);
        }
    };
    /* first pass: column i, in place */
    if (!dc_only) {
#pragma unroll
        for (int k = 0; k < N; k++)
            x[k] = mine[k * N + i];
        run(adst1, true);
    }
    vp_wave_sync();
    if (!dc_only) {
#pragma unroll
        for (int k = 0; k < N; k++)
            mine[k * N + i] = (COEF)o[k];
    }
    vp_wave_sync();
    /* second pass: row i of that matrix; its outputs are picture column i, so they go back transposed */
    if (!dc_only) {
#pragma unroll
        for (int k = 0; k < N; k++)
            x[k] = mine[i * N + k];
        run(adst2, false);
    }
    vp_wave_sync();
#pragma unroll
    for (int k = 0; k < N; k++)
        mine[k * N + i] = dc_only ? (COEF)dcv : (COEF)o[k];
    vp_wave_sync();
    /* ---- picture += (residual + round) >> bits, row i of my unit ---- */
    if (live) {
        const COEF *r = mine + i * N;
        int z[N];
#pragma unroll
        for (int c = 0; c < N; c++)
            z[c] = BITS ? (int)((uint32_t)(int)r[c] + (1u << (BITS ? BITS - 1 : 0))) >> BITS : (int)r[c];
        ffhip_add_row<N>(dst + tu.dst_offset + (ptrdiff_t)i * stride, z, HBD ? bd : 8);
    }
    /* ---- the block is consumed: zeroed (dc-only: block[0] alone, as the reference leaves the rest untouched) ---- */
    for (int t = lane; t < UPW * Q4; t += 64) {
        const int b = t / Q4, w = t % Q4;
        if (u0 + b < n) {
            const FFHipVp9TU tb = tus[u0 + b];
            COEF *g = coeffs + tb.coeff_offset;
            uint32_t *cw = reinterpret_cast<uint32_t *>(g);
            const bool dcb = !WHT && tb.dc_only && (LOG2 == 5 || tb.txtp == 0);
            if (!dcb) {
                if (!(reinterpret_cast<uintptr_t>(g) & 15)) {
                    reinterpret_cast<uint4 *>(g)[w] = make_uint4(0, 0, 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        cw[4 * w + k] = 0;
                }
            } else if (w == 0) {
                cw[0] = HBD ? 0u : cw[0] & 0xFFFF0000u;
            }
        }
    }
}

int ffhip_launch_vp9_itxfm(int tx, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n, hipStream_t stream)
{
    return ffhip_launch_vp9_itxfm_bd(8, tx, coeffs, dst, stride, tus, n, stream);
}

/* bd 8: coeffs are int16; bd 10 / 12: int32 (the reference's dctcoef), coeff_offset counts coefficients either way */
int ffhip_launch_vp9_itxfm_bd(int bd, int tx, void *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (tx < 0 || tx > 4) {
        ffhip_set_error("ffhip_vp9_itxfm: tx %d outside 0..4", tx);
        return FFHIP_EINVAL;
    }
    if (bd != 8 && !((bd == 10 || bd == 12) && !(((uintptr_t)dst | (size_t)stride) & 1) && !((uintptr_t)coeffs & 3))) {
        ffhip_set_error("ffhip_vp9_itxfm: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte, int32 coefficients 4-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    const int log2 = tx == 4 ? 2 : tx + 2, upw = 64 >> log2;
    const dim3 grid(cdiv(n, 4 * upw)), block(256);
#define VT_CASE(T, LG, W) case T: if (bd == 8) hipLaunchKernelGGL((k_vp9_itxfm<LG, W, false>), grid, block, 0, stream, coeffs, dst, stride, tus, n, 8); \
                                  else hipLaunchKernelGGL((k_vp9_itxfm<LG, W, true>), grid, block, 0, stream, coeffs, dst, stride, tus, n, bd); break;
    switch (tx) {
    VT_CASE(0, 2, false) VT_CASE(1, 3, false) VT_CASE(2, 4, false) VT_CASE(3, 5, false)
    default:
    VT_CASE(4, 2, true)
    }
#undef VT_CASE
    LAUNCH_CHECK();
    return 0;
}
