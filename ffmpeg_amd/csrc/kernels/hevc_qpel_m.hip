/*
 * hevc_qpel_m.hip — put_hevc_qpel_uni_{pixels,h,v,hv} for 16 x 16 luma blocks on the MATRIX CORES (8 bits), the skeleton of
 * k_h264_qpel_m (h264_qpel.hip) with HEVC's 8-tap filters (libavcodec/h26x/h2656_inter_template.c:97-245 put_uni_*;
 * libavcodec/hevc/dsp_template.c:300-420 QPEL_FILTER; ff_hevc_qpel_filters, libavcodec/hevc/filter tables in dsp.c:40-45).
 *
 * A 16 x 16 block with its 23 x 23 footprint is two dense products:
 *   stage 1  C1[row][col] = sum_k raw'[row][k] * Th[mx][k][col]   v_mfma_i32_16x16x32_i8 twice (footprint rows 0..15 and 16..22);
 *            A = 8 footprint bytes per lane (lane = row m, byte group g), B = the banded 8-tap matrix of the block's mx — a per-lane
 *            constant read from a 2 KiB table — or, for mx = 0, the shifted identity that hands the raw samples through.
 *            raw' = raw ^ 0x80; the missing 128 * sum(taps) = 8192 (128 for the identity) is the accumulator's initial value.
 *   stage 2  C2[y][col] = sum_r Tv[my][y][r] * X[r][col]          the same instruction with A = Tv and B = the lane's own eight C1
 *            values (they are eight consecutive K slots of its column once K is numbered to match): the raw bytes (mx = 0), or the
 *            sums' low and high bytes — two products, (hi << 8) + lo — since a horizontal sum needs 16 bits (-6120 .. 22440).
 *            my = 0: no second stage — the lanes feed footprint rows 3 .. 18 to stage 1, whose C layout then IS the output block.
 *   then     the reference's rounding: (sum + 32) >> 6, or ((sum >> 6) + 32) >> 6 behind both passes, saturating packs, the 4 x 4 byte
 *            transposition on DPP, and the four blocks of a wave leave through an LDS tile as 16-byte row stores.
 * Every product and sum is an exact int32: bit-exact.  The memory side is k_h264_qpel_m's: aligned 16-byte footprint chunks (23 rows x
 * 3 chunks = 69: one load per lane and a second one on five lanes), a wave-private LDS plane, workgroups numbered so that an XCD takes a
 * contiguous eighth of the batch.
 * The other output stages (MODE 0 put -> int16, 2 uni_w, 3 bi, 4 bi_w: hevc/dsp_template.c:368-420,432-625,630-815) take the 14-bit
 * intermediate instead of the rounded sample: the lane's four column-strip values are transposed as two byte planes (low bytes, high
 * bytes: the same two DPP steps each) into four int16 of a row, and the stage — the weights, the other list's int16 block read as
 * 8-byte row pieces — runs in the row layout.  Blocks of any other size and chroma stay with k_hevc_mc (hevc_mc.hip), which skips what
 * this kernel took.
 */
#include <type_traits>

#include "common.h"
#include "h264_kernels.h"

typedef int hq_i4 __attribute__((ext_vector_type(4)));
typedef short hq_s2 __attribute__((ext_vector_type(2)));
typedef uint32_t hq_u4 __attribute__((ext_vector_type(4)));

struct HqTab { unsigned long long th[4][64], tv[4][64]; };
constexpr int hq_tap(int f, int t)
{
    constexpr int c[4][8] = { { 0, 0, 0, 1, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
    return (t < 0 || t > 7) ? 0 : c[f][t];
}
constexpr HqTab hq_make()
{
    HqTab t{};
    for (int f = 0; f < 4; f++)
        for (int l = 0; l < 64; l++) {
            const int g = l >> 4, n = l & 15;
            unsigned long long th = 0, tv = 0;
            for (int j = 0; j < 8; j++) {
                const int k = 8 * g + j;                                    /* stage 1: footprint byte k, output column n */
                const int rho = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);   /* stage 2: K slot 8g + j holds footprint row rho; output row n */
                th |= (unsigned long long)(unsigned char)(signed char)hq_tap(f, k - n) << (8 * j);
                tv |= (unsigned long long)(unsigned char)(signed char)hq_tap(f, rho - n) << (8 * j);
            }
            t.th[f][l] = th; t.tv[f][l] = tv;
        }
    return t;
}
__device__ const HqTab hq_tab = hq_make();

__device__ __forceinline__ uint32_t hq_sat_pk_u8(uint32_t pk) /* two int16 -> two uint8, saturating, in the low half */
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(pk));
    return r;
}
/* four int32 (each within int16) -> clip_u8((v + 32) >> 6) x 4, byte r = value r */
__device__ __forceinline__ uint32_t hq_round6(int a, int b, int c, int d)
{
    const hq_s2 k32 = { 32, 32 };
    hq_s2 lo = __builtin_amdgcn_cvt_pk_i16(a, b), hi = __builtin_amdgcn_cvt_pk_i16(c, d);
    lo = (lo + k32) >> (short)6;
    hi = (hi + k32) >> (short)6;
    return __builtin_amdgcn_perm(hq_sat_pk_u8(__builtin_bit_cast(uint32_t, hi)), hq_sat_pk_u8(__builtin_bit_cast(uint32_t, lo)), 0x05040100u);
}
__device__ __forceinline__ long hq_long(uint32_t lo, uint32_t hi) { return (long)(((unsigned long)hi << 32) | lo); }
__device__ __forceinline__ hq_i4 hq_splat(int v) { return (hq_i4){ v, v, v, v }; }

/* 4 x 4 byte transposition inside each lane quad: lane 4q + j gets row 4g + j, columns 4q .. 4q + 3 */
__device__ __forceinline__ uint32_t hq_transpose(uint32_t c, uint32_t selT1, uint32_t selT2)
{
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)c, 0xB1, 0xf, 0xf, true);     /* quad_perm [1,0,3,2] */
    const uint32_t c1 = __builtin_amdgcn_perm(t1, c, selT1);
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)c1, 0x4E, 0xf, 0xf, true);    /* quad_perm [2,3,0,1] */
    return __builtin_amdgcn_perm(t2, c1, selT2);
}

/* MODE 0 put (int16), 1 uni, 2 uni_w, 3 bi, 4 bi_w; modes 2..4 read the 24-byte weighted record */
template <int MODE>
__global__ __launch_bounds__(256) void k_hevc_qpel_m(void *dst_, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                                                     const void *blocks_, int n, int per_xcd, int full)
{
    using Rec = typename std::conditional<(MODE >= 2), FFHipHevcMcWBlock, FFHipHevcMcBlock>::type;
    __shared__ __align__(16) uint32_t rawp[4][24 * 12];
    __shared__ __align__(16) uint32_t obp[4][4 * 64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int wg = per_xcd ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int b0 = (wg * 4 + wave) * 4;
    if (b0 >= n)
        return;
    uint8_t *dst = static_cast<uint8_t *>(dst_);
    uint32_t *raw = rawp[wave], *ob = obp[wave];
    const int fr0 = (lane * 171) >> 9, fc0 = lane - 3 * fr0;          /* chunk `lane`: footprint row lane / 3, 16-byte chunk lane % 3 */
    const int fr1 = (64 + lane) / 3, fc1 = 64 + lane - 3 * fr1;        /* chunk 64 + lane (lanes 0..4: rows 21, 22) */
    const uint32_t selT1 = (lane & 1) ? 0x03070105u : 0x06020400u, selT2 = (lane & 2) ? 0x03020706u : 0x05040100u;
    const int g = lane >> 4, m = lane & 15;
    const int ry = 4 * g + (lane & 3), rxg = (lane >> 2) & 3;          /* the row and 4-sample group this lane owns after the transposition */
    const long K80 = (long)0x8080808080808080ull;

    /* records and footprints of the wave's four blocks, all in flight before the first is used */
    int Gmx[4], Gmy[4], Gdoff[4], Gwx0[4], Gwx1[4], Gox[4], Gwsh[4], Gs2[4];
    bool Gel[4], tile = MODE != 0;
    uint32_t Gsh16[4];
    hq_u4 Gf0[4], Gf1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const Rec rec = static_cast<const Rec *>(blocks_)[min(b0 + k, n - 1)];
        const int w = __builtin_amdgcn_readfirstlane((int)rec.width), h = __builtin_amdgcn_readfirstlane((int)rec.height);
        Gmx[k] = __builtin_amdgcn_readfirstlane((int)rec.mx) & 3;
        Gmy[k] = __builtin_amdgcn_readfirstlane((int)rec.my) & 3;
        Gdoff[k] = __builtin_amdgcn_readfirstlane(rec.dst_offset);
        Gwx0[k] = Gwx1[k] = Gox[k] = Gwsh[k] = Gs2[k] = 0;
        if constexpr (MODE >= 2) {
            Gwx0[k] = __builtin_amdgcn_readfirstlane((int)rec.wx0); Gwx1[k] = __builtin_amdgcn_readfirstlane((int)rec.wx1);
            Gox[k] = __builtin_amdgcn_readfirstlane((int)rec.ox);
            Gwsh[k] = __builtin_amdgcn_readfirstlane((int)rec.denom) + 6; /* uni_w: shift = denom + 14 - 8;  bi_w: log2Wd = denom + 6 */
            Gs2[k] = __builtin_amdgcn_readfirstlane(rec.src2_offset);
        }
        Gel[k] = b0 + k < n && w == 16 && h == 16;
        tile = tile && Gel[k] && !((reinterpret_cast<uintptr_t>(dst) + (uintptr_t)(intptr_t)Gdoff[k]) & 3);
        const uint8_t *s0 = src + __builtin_amdgcn_readfirstlane(rec.src_offset) - 3 - 3 * srcstride;
        Gsh16[k] = (uint32_t)(reinterpret_cast<uintptr_t>(s0) & 15);
        const uint8_t *sa = s0 - Gsh16[k];
        /* the part of the footprint the position reads: all 23 rows / columns behind a filter, the block's own 16 otherwise */
        const int r_lo = Gmy[k] || full ? 0 : 3, r_hi = Gmy[k] || full ? 22 : 18;
        const int c_lo = (int)Gsh16[k] + (Gmx[k] || full ? 0 : 3), c_hi = (int)Gsh16[k] + (Gmx[k] || full ? 22 : 18);   /* bytes of the aligned row */
        const bool want0 = Gel[k] && fr0 >= r_lo && fr0 <= r_hi && 16 * fc0 + 15 >= c_lo && 16 * fc0 <= c_hi;
        const bool want1 = Gel[k] && lane < 5 && fr1 >= r_lo && fr1 <= r_hi && 16 * fc1 + 15 >= c_lo && 16 * fc1 <= c_hi;
        Gf0[k] = want0 ? *reinterpret_cast<const hq_u4 *>(sa + (ptrdiff_t)fr0 * srcstride + 16 * fc0) : (hq_u4){ 0, 0, 0, 0 };
        Gf1[k] = want1 ? *reinterpret_cast<const hq_u4 *>(sa + (ptrdiff_t)fr1 * srcstride + 16 * fc1) : (hq_u4){ 0, 0, 0, 0 };
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!Gel[k])
            continue; /* k_hevc_mc<.., SKIP16> takes it */
        /* the other list's int16 row piece of this lane (bi, bi_w): in flight under the products */
        int o2[4] = { 0, 0, 0, 0 };
        if constexpr (MODE >= 3) {
            const int16_t *q = src2 + Gs2[k] + ry * 64 + 4 * rxg;
            if (!(reinterpret_cast<uintptr_t>(q) & 7)) {
                const uint2 v = *reinterpret_cast<const uint2 *>(q);
                o2[0] = (int16_t)(v.x & 0xffff); o2[1] = (int)v.x >> 16; o2[2] = (int16_t)(v.y & 0xffff); o2[3] = (int)v.y >> 16;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    o2[j] = q[j];
            }
        }
        *reinterpret_cast<hq_u4 *>(raw + fr0 * 12 + 4 * fc0) = Gf0[k];
        if (lane < 5)
            *reinterpret_cast<hq_u4 *>(raw + fr1 * 12 + 4 * fc1) = Gf1[k];
        __builtin_amdgcn_wave_barrier();
        const int mx = Gmx[k], my = Gmy[k];
        const uint32_t sh = Gsh16[k] & 3;
        const uint32_t *r0p = raw + (Gsh16[k] >> 2);   /* the dword that holds footprint byte 0 of row 0 */
        uint32_t out = 0;                               /* MODE 1: the four rounded samples of the lane's row piece */
        int v[4] = { 0, 0, 0, 0 };                      /* other modes: their 14-bit intermediates */
        if (mx | my) {
            /* my = 0: the sixteen output rows are footprint rows 3 .. 18 — the lanes pick those, and stage 1 is the whole filter */
            const uint32_t *pa = r0p + (my ? m : m + 3) * 12 + 2 * g;
            const uint32_t a0 = pa[0], a1 = pa[1], a2 = pa[2];
            const long fa = hq_long(__builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh)) ^ K80;
            const long cTh = (long)hq_tab.th[mx][lane];
            const int bias1 = mx ? 8192 : 128;
            const hq_i4 h0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa, cTh, hq_splat(bias1), 0, 0, 0);
            hq_i4 s = h0;                               /* the lane's column strip: rows 4g .. 4g + 3 of column n */
            if (my) {
                const long cTv = (long)hq_tab.tv[my][lane];
                const uint32_t *pb = r0p + min(16 + m, 22) * 12 + 2 * g;
                const uint32_t b0_ = pb[0], b1_ = pb[1], b2_ = pb[2];
                const long fb = hq_long(__builtin_amdgcn_alignbyte(b1_, b0_, sh), __builtin_amdgcn_alignbyte(b2_, b1_, sh)) ^ K80;
                const hq_i4 h1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fb, cTh, hq_splat(bias1), 0, 0, 0);
                if (!mx) {
                    /* the raw samples of column n in the stage-2 layout */
                    const uint32_t x0 = __builtin_amdgcn_perm((uint32_t)h0.y, (uint32_t)h0.x, 0x0c0c0400u), x1 = __builtin_amdgcn_perm((uint32_t)h0.w, (uint32_t)h0.z, 0x0c0c0400u);
                    const uint32_t x2 = __builtin_amdgcn_perm((uint32_t)h1.y, (uint32_t)h1.x, 0x0c0c0400u), x3 = __builtin_amdgcn_perm((uint32_t)h1.w, (uint32_t)h1.z, 0x0c0c0400u);
                    const long bv = hq_long(__builtin_amdgcn_perm(x1, x0, 0x05040100u), __builtin_amdgcn_perm(x3, x2, 0x05040100u)) ^ K80;
                    s = __builtin_amdgcn_mfma_i32_16x16x32_i8(cTv, bv, hq_splat(8192), 0, 0, 0);
                } else {
                    /* 16-bit sums: a low byte (made signed by ^0x80, + 128 * 64 in the accumulator) and a signed high byte */
                    const uint32_t p0 = __builtin_amdgcn_perm((uint32_t)h0.y, (uint32_t)h0.x, 0x05010400u), p1 = __builtin_amdgcn_perm((uint32_t)h0.w, (uint32_t)h0.z, 0x05010400u);
                    const uint32_t p2 = __builtin_amdgcn_perm((uint32_t)h1.y, (uint32_t)h1.x, 0x05010400u), p3 = __builtin_amdgcn_perm((uint32_t)h1.w, (uint32_t)h1.z, 0x05010400u);
                    const long blo = hq_long(__builtin_amdgcn_perm(p1, p0, 0x05040100u), __builtin_amdgcn_perm(p3, p2, 0x05040100u)) ^ K80;
                    const long bhi = hq_long(__builtin_amdgcn_perm(p1, p0, 0x07060302u), __builtin_amdgcn_perm(p3, p2, 0x07060302u));
                    const hq_i4 chi = __builtin_amdgcn_mfma_i32_16x16x32_i8(cTv, bhi, hq_splat(0), 0, 0, 0);
                    const hq_i4 clo = __builtin_amdgcn_mfma_i32_16x16x32_i8(cTv, blo, hq_splat(8192), 0, 0, 0);
                    s = (hq_i4){ ((chi.x << 8) + clo.x) >> 6, ((chi.y << 8) + clo.y) >> 6, ((chi.z << 8) + clo.z) >> 6, ((chi.w << 8) + clo.w) >> 6 };
                }
            }
            if constexpr (MODE == 1) {
                out = hq_transpose(hq_round6(s.x, s.y, s.z, s.w), selT1, selT2);
            } else {
                /* every intermediate fits int16 (|.| <= 30855): two byte planes through the same transposition, then int16 again */
                const uint32_t pk01 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(s.x, s.y)), pk23 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(s.z, s.w));
                const uint32_t lo = hq_transpose(__builtin_amdgcn_perm(pk23, pk01, 0x06040200u), selT1, selT2);
                const uint32_t hi = hq_transpose(__builtin_amdgcn_perm(pk23, pk01, 0x07050301u), selT1, selT2);
                const uint32_t w01 = __builtin_amdgcn_perm(hi, lo, 0x05010400u), w23 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
                v[0] = (int)(w01 << 16) >> 16; v[1] = (int)w01 >> 16; v[2] = (int)(w23 << 16) >> 16; v[3] = (int)w23 >> 16;
            }
        } else {
            /* the block itself, in the row layout: put_hevc_pel_uni_pixels, or << 6 as the other stages' intermediate */
            const uint32_t o = sh + 3;
            const uint32_t *pf = r0p + (ry + 3) * 12 + rxg + (o >> 2);
            out = __builtin_amdgcn_alignbyte(pf[1], pf[0], o & 3);
            if constexpr (MODE != 1) {
                v[0] = (int)(out & 255) << 6; v[1] = (int)((out >> 8) & 255) << 6; v[2] = (int)((out >> 16) & 255) << 6; v[3] = (int)(out >> 24) << 6;
            }
        }
        if constexpr (MODE == 0) {
            int16_t *d = static_cast<int16_t *>(dst_) + Gdoff[k] + ry * 64 + 4 * rxg;
            if (!(reinterpret_cast<uintptr_t>(d) & 7)) {
                *reinterpret_cast<uint2 *>(d) = make_uint2(((uint32_t)v[0] & 0xffffu) | ((uint32_t)v[1] << 16), ((uint32_t)v[2] & 0xffffu) | ((uint32_t)v[3] << 16));
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    d[j] = (int16_t)v[j];
            }
        } else {
            if constexpr (MODE >= 2) {
                const int wsh = Gwsh[k], wx0 = Gwx0[k], wx1 = Gwx1[k], ox = Gox[k];
                const int wofs = MODE == 2 ? 1 << (wsh - 1) : (ox + 1) << wsh;
                int r[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if constexpr (MODE == 2)
                        r[j] = ((v[j] * wx0 + wofs) >> wsh) + ox;
                    else if constexpr (MODE == 3)
                        r[j] = (v[j] + o2[j] + 64) >> 7;
                    else
                        r[j] = (v[j] * wx1 + o2[j] * wx0 + wofs) >> (wsh + 1);
                }
                out = (uint32_t)clip_u8(r[0]) | (uint32_t)clip_u8(r[1]) << 8 | (uint32_t)clip_u8(r[2]) << 16 | (uint32_t)clip_u8(r[3]) << 24;
            }
            if (tile) {
                ob[64 * k + 4 * ry + rxg] = out;
            } else {
                uint8_t *d = dst + Gdoff[k] + (ptrdiff_t)ry * dststride + 4 * rxg;
                if (!((reinterpret_cast<uintptr_t>(d)) & 3)) {
                    *reinterpret_cast<uint32_t *>(d) = out;
                } else {
                    for (int i = 0; i < 4; i++)
                        d[i] = (uint8_t)(out >> (8 * i));
                }
            }
        }
        __builtin_amdgcn_wave_barrier(); /* the next block overwrites the plane */
    }
    if (MODE != 0 && tile) {
        const int y = lane >> 2, c = lane & 3;
        const hq_u4 o = *reinterpret_cast<const hq_u4 *>(ob + 64 * c + 4 * y);
        const int doff = c == 0 ? Gdoff[0] : c == 1 ? Gdoff[1] : c == 2 ? Gdoff[2] : Gdoff[3];
        *reinterpret_cast<hq_u4 *>(dst + doff + (ptrdiff_t)y * dststride) = o;
    }
}

/* true when the kernel above may take the batch's 16 x 16 blocks: the aligned 16-byte chunk loads need a source stride that keeps a
 * row's alignment, the dword stores a destination stride of whole dwords (put writes int16 rows of its own pitch) */
bool ffhip_hevc_qpel_m_ok(int mode, ptrdiff_t dststride, ptrdiff_t srcstride) { return !(srcstride & 15) && (mode == 0 || !(dststride & 3)); }

void ffhip_launch_hevc_qpel_m(int mode, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                              const void *blocks, int n, hipStream_t stream)
{
    const int per_xcd = cdiv(cdiv(n, 16), 8);
    const char *ef = FFHIP_KNOB("FFHIP_HEVC_QM_FULL"); /* measured variant: 1 = every block loads its whole 23 x 23 footprint */
    const int full = ef && ef[0] == '1' ? 1 : 0;
#define QM_CASE(M) case M: hipLaunchKernelGGL((k_hevc_qpel_m<M>), dim3(8 * per_xcd), dim3(256), 0, stream, dst, dststride, src, srcstride, src2, blocks, n, per_xcd, full); break;
    switch (mode) {
    QM_CASE(0) QM_CASE(1) QM_CASE(2) QM_CASE(3)
    default:
    QM_CASE(4)
    }
#undef QM_CASE
}
