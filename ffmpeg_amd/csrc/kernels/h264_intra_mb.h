/*
 * h264_intra_mb.h — reconstruction of ONE intra macroblock (4:2:0, frame macroblock, no transform bypass; templated on the sample
 * type: uint8_t with int16_t coefficients, uint16_t at 9..14 bits with int32_t coefficients — pixel / dctcoef of
 * libavcodec/bit_depth_template.c:39-50): what
 * hl_decode_mb() does between the two xchg_mb_border() calls and after them (libavcodec/h264_mb_template.c:151-262,
 * hl_decode_mb_predict_luma / hl_decode_mb_idct_luma libavcodec/h264_mb.c:612-760):
 *
 *     pred8x8[chroma_pred_mode] on Cb, Cr
 *     Intra4x4:            16 x ( pred4x4[dir]  -> idct_add / idct_dc_add )            in block order 0..15
 *     Intra4x4 + 8x8 DCT:   4 x ( pred8x8l[dir] -> idct8_add / idct8_dc_add )
 *     Intra16x16:          pred16x16[mode], luma_dc_dequant_idct, idct_add16intra
 *     chroma (cbp & 0x30): chroma_dc_dequant_idct per plane, idct_add8
 *
 * on a TILE: the macroblock's samples plus the neighbours the predictors read (the row above running on into the top-right
 * macroblock, the column to the left, the corner).  The code is a sequence of PHASES; inside a phase the 64 lanes are independent
 * (no lane reads what another lane writes in the same phase), between phases the tile is synchronised.  The phase bodies are
 * plain C++ shared by the two things that run them:
 *
 *   - k_h264_intra_frame (h264_intra.hip): one wave per macroblock row, tile in LDS, a phase = body(lane) + wave barrier;
 *   - the host emulation (oracle/emul_h264_intra.cpp, test infrastructure): a phase = for (lane = 0..63) body(lane) — it pins
 *     this logic against the oracle's restatement of hl_decode_mb() on the CPU, where there is no GPU to run the kernel.
 *
 * The prediction rules are the per-sample rules over the block's edge line that h264_pred.hip uses (h264pred_template.c).
 */
#ifndef FFHIP_H264_INTRA_MB_H
#define FFHIP_H264_INTRA_MB_H
#include <stdint.h>

#include "ffhip.h"

#if defined(__HIPCC__)
#define IMB_FN __host__ __device__ __forceinline__
#define IMB_MEM __host__ __device__ __forceinline__
#else
#define IMB_FN static inline
#define IMB_MEM inline
#endif

/* ---- prediction rules over the edge line e[]: left column bottom-up, corner, row above, top-right ---- */
IMB_FN int hp_a2(int a, int b) { return (a + b + 1) >> 1; }
IMB_FN int hp_a3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }

/* neighbours a pred4x4 / pred8x8l mode reads: bit0 left, bit1 top, bit2 corner, bit3 top-right (h264pred.h:35-48 order) */
IMB_FN unsigned hp_need(int mode)
{
    /* 12 nibbles, mode 0 in the low one: V=2 H=1 DC=3 DDL=a DDR=7 VR=7 HD=7 VL=a HU=1 LEFT_DC=1 TOP_DC=2 DC_128=0 */
    return (unsigned)(0x0211a777a312ull >> (4 * mode)) & 15u;
}

/* the edge line as a callable e(k) (an array in LDS, or the tile read in place) */
struct HpArr {
    const int *p;
    IMB_MEM int operator()(int k) const { return p[k]; }
};

template <int N, class E>
IMB_FN int hp_dir_sample_e(int mode, const E &e, int x, int y, int dc)
{
    auto T = [&](int k) { return e(N + 1 + k); };
    switch (mode) {
    case 0: return T(x);
    case 1: return e(N - 1 - y);
    case 3: {
        const int i = x + y;
        return i < 2 * N - 2 ? hp_a3(T(i), T(i + 1), T(i + 2)) : (T(2 * N - 2) + 3 * T(2 * N - 1) + 2) >> 2;
    }
    case 4: {
        const int i = N - 1 - y + x;
        return hp_a3(e(i), e(i + 1), e(i + 2));
    }
    case 5: {
        const int d = 2 * x - y, h = d >> 1;
        if (d < 0)
            return hp_a3(e(N + d), e(N + d + 1), e(N + d + 2));
        return (d & 1) ? hp_a3(e(N + h), e(N + h + 1), e(N + h + 2)) : hp_a2(e(N + h), e(N + h + 1));
    }
    case 6: {
        const int d = 2 * y - x, h = d >> 1;
        if (d < 0)
            return hp_a3(e(N - d - 2), e(N - d - 1), e(N - d));
        return (d & 1) ? hp_a3(e(N - h), e(N - h - 1), e(N - h - 2)) : hp_a2(e(N - h), e(N - h - 1));
    }
    case 7: {
        const int i = (y >> 1) + x;
        return (y & 1) ? hp_a3(T(i), T(i + 1), T(i + 2)) : hp_a2(T(i), T(i + 1));
    }
    case 8: {
        const int i = 2 * y + x, j = N - 1 - (i >> 1);
        if (i >= 2 * N - 2)
            return e(0);
        if (i == 2 * N - 3)
            return (e(1) + 3 * e(0) + 2) >> 2;
        return (i & 1) ? hp_a3(e(j), e(j - 1), e(j - 2)) : hp_a2(e(j), e(j - 1));
    }
    default: return dc;
    }
}

template <int N>
IMB_FN int hp_dir_sample(int mode, const int *e, int x, int y, int dc)
{
    return hp_dir_sample_e<N>(mode, HpArr{ e }, x, y, dc);
}

/* the DC of modes 2 (both sides), 9 (LEFT_DC), 10 (TOP_DC); 128 otherwise */
template <int N, class E>
IMB_FN int hp_dir_dc_e(int mode, const E &e, int mid = 128 /* 1 << (bit_depth - 1) */)
{
    if (mode != 2 && mode != 9 && mode != 10)
        return mid;
    int sl = 0, st = 0;
    for (int i = 0; i < N; i++) {
        sl += mode != 10 ? e(i) : 0;
        st += mode != 9 ? e(N + 1 + i) : 0;
    }
    return mode == 2 ? (sl + st + N) >> (N == 8 ? 4 : 3) : ((mode == 9 ? sl : st) + N / 2) >> (N == 8 ? 3 : 2);
}

template <int N>
IMB_FN int hp_dir_dc(int mode, const int *e)
{
    return hp_dir_dc_e<N>(mode, HpArr{ e });
}

/* PREDICT_8x8_LOAD_LEFT / _TOP / _TOPRIGHT / _TOPLEFT (h264pred_template.c:822-856): entry j of the low-pass filtered line from
 * the raw line w(0..24); entries the mode does not read are 0 */
template <class W>
IMB_FN int hp_filter8_e(const W &w, int j, unsigned need, bool tl, bool tr)
{
    if (j < 8) {
        if (!(need & 1))
            return 0;
        return j == 7 ? hp_a3(tl ? w(8) : w(7), w(7), w(6)) : j == 0 ? (w(1) + 3 * w(0) + 2) >> 2 : hp_a3(w(j + 1), w(j), w(j - 1));
    }
    if (j == 8)
        return (need & 4) ? hp_a3(w(7), w(8), w(9)) : 0;
    if (j < 17) {
        if (!(need & 2))
            return 0;
        return j == 9 ? hp_a3(tl ? w(8) : w(9), w(9), w(10)) : j == 16 ? hp_a3(tr ? w(17) : w(16), w(16), w(15)) : hp_a3(w(j - 1), w(j), w(j + 1));
    }
    if (!(need & 8))
        return 0;
    return !tr ? w(16) : j == 24 ? (w(23) + 3 * w(24) + 2) >> 2 : hp_a3(w(j - 1), w(j), w(j + 1));
}

IMB_FN int hp_filter8(const int *w, int j, unsigned need, bool tl, bool tr)
{
    return hp_filter8_e(HpArr{ w }, j, need, tl, tr);
}

/* ---- the tile ---- */
template <typename PIX> struct ImbCoef { typedef int16_t T; };
template <> struct ImbCoef<uint16_t> { typedef int32_t T; };

template <typename PIX>
struct ImbTileT {
    PIX y[17 * 32];        /* luma:   sample (r, c), r = -1..15, c = -4..27, at [(r + 1) * 32 + c + 4] */
    PIX c[2][9 * 16];      /* chroma: sample (r, c), r = -1..7,  c = -4..11, at [(r + 1) * 16 + c + 4] */
    int t8[4][64];         /* first pass of the four 8x8 inverse transforms */
    int dcq[16];           /* luma_dc_dequant_idct's results by block */
    int edge[32];          /* pred8x8l: the block's low-pass filtered edge line (25 entries) */
    typename ImbCoef<PIX>::T zero[16]; /* sixteen zero coefficients: the block of a lane whose block was not stored (set once per tile) */
};
typedef ImbTileT<uint8_t> ImbTile;

IMB_FN int imb_yi(int r, int c) { return (r + 1) * 32 + c + 4; }
IMB_FN int imb_ci(int r, int c) { return (r + 1) * 16 + c + 4; }
IMB_FN int imb_clip_u8(int v)
{
    int r = v < 0 ? 0 : v > 255 ? 255 : v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(r)); /* keeps hipcc from folding neighbouring clips into v_ashr_pk_u8_i32 (common.h: its upper half is not zero) */
#endif
    return r;
}
/* av_clip_pixel at the tile's depth: maxv = (1 << bit_depth) - 1 */
template <typename PIX>
IMB_FN int imb_clip(int v, int maxv);
/* prediction + residual as the sample keeps it: clipped by the inverse transforms' adds, modulo the sample type by add_pixels*_clear and
 * the pred*_add forms of the transform bypass (`dst[i] += src[i]` on pixel, h264addpx_template.c:35-40) */
template <typename PIX>
IMB_FN int imb_fin(int v, int maxv, bool bypass)
{
    return bypass ? (int)(PIX)v : imb_clip<PIX>(v, maxv);
}
template <typename PIX>
IMB_FN int imb_clip(int v, int maxv)
{
    if (sizeof(PIX) == 1)
        return imb_clip_u8(v);
    return v < 0 ? 0 : v > maxv ? maxv : v;
}
IMB_FN int imb_popc(uint32_t v) { return __builtin_popcount(v); }

/* position of 4x4 block i inside the macroblock (h->block_offset[i], h264_slice.c init_scan_tables / block_offset) */
IMB_FN int imb_bx(int i) { return 4 * ((i & 1) + ((i >> 2) & 1) * 2); }
IMB_FN int imb_by(int i) { return 4 * (((i >> 1) & 1) + ((i >> 3) & 1) * 2); }

/* coefficients of luma block i (0..15; 8x8 transform: i = 0, 4, 8, 12, 64 coefficients) / chroma block 16 + k (Cb), 20 + k (Cr)
 * in the macroblock's packed run (run = the picture's coefficient array + R.coef, or a copy of the run); nullptr when the block was
 * all zero and not stored.  Above 8 bits a run that carries FFHIP_H264_INTRA_LUMA_DC starts with the sixteen int32 luma DCs
 * (sl->mb_luma_dc as dctcoef; the record's int16 luma_dc[] cannot hold them). */
template <typename CF>
IMB_FN const CF *imb_block(const FFHipH264IntraMB &R, const CF *run, int bit)
{
    if (!((R.blocks >> bit) & 1u))
        return nullptr;
    const int lsz = R.type == FFHIP_H264_INTRA_8x8 ? 64 : 16;
    const uint32_t below = R.blocks & ((1u << bit) - 1u);
    const int lead = sizeof(CF) == 4 && (R.flags & FFHIP_H264_INTRA_LUMA_DC) ? 16 : 0;
    return run + lead + imb_popc(below & 0xFFFFu) * lsz + imb_popc(below >> 16) * 16;
}

/* the record's fields every lane reads alike: on the device the record lives in LDS and a field read in place is a vector value —
 * a branch on it is compiled as a masked region, its popcounts as vector code.  Taken once per macroblock into scalar registers. */
struct ImbHead {
    int type, cbp, flags;
    uint32_t blocks;
};
template <typename CF>
IMB_FN const CF *imb_block(const ImbHead &H, const CF *run, int bit)
{
    if (!((H.blocks >> bit) & 1u))
        return nullptr;
    const int lsz = H.type == FFHIP_H264_INTRA_8x8 ? 64 : 16;
    const uint32_t below = H.blocks & ((1u << bit) - 1u);
    const int lead = sizeof(CF) == 4 && (H.flags & FFHIP_H264_INTRA_LUMA_DC) ? 16 : 0;
    return run + lead + imb_popc(below & 0xFFFFu) * lsz + imb_popc(below >> 16) * 16;
}

/* int16 entries of a macroblock's run; psz = sizeof(sample) */
IMB_FN int imb_run_len(int type, uint32_t blocks, int psz = 1, int flags = 0)
{
    if (type == FFHIP_H264_INTRA_PCM)
        return 192 * psz;
    const int n = imb_popc(blocks & 0xFFFFu) * (type == FFHIP_H264_INTRA_8x8 ? 64 : 16) + imb_popc(blocks >> 16) * 16;
    return psz == 1 ? n : 2 * (n + ((flags & FFHIP_H264_INTRA_LUMA_DC) ? 16 : 0));
}

/* output k of the 4-point butterfly of ff_h264_idct_add (h264idct_template.c:39-64): modulo 2^32, arithmetic shifts */
IMB_FN int imb_bfly4(int k, int s0, int s1, int s2, int s3)
{
    const uint32_t e0 = (uint32_t)s0 + (uint32_t)s2, e1 = (uint32_t)s0 - (uint32_t)s2;
    const uint32_t o0 = (uint32_t)(s1 >> 1) - (uint32_t)s3, o1 = (uint32_t)s1 + (uint32_t)(s3 >> 1);
    return (int)(k == 0 ? e0 + o1 : k == 1 ? e1 + o0 : k == 2 ? e1 - o0 : e0 - o1);
}

/* what ff_h264_idct_add adds to sample (x, y) of its block; b[0] is passed separately (Intra16x16 puts the dequantised DC there) */
template <typename CF>
IMB_FN int imb_idct4_at(const CF *b, int b0, int x, int y)
{
    int r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int s0 = j == 0 ? (CF)(b0 + 32) : (b ? b[j] : 0);
        r[j] = (CF)imb_bfly4(x, s0, b ? b[j + 4] : 0, b ? b[j + 8] : 0, b ? b[j + 12] : 0); /* the first pass stores dctcoef */
    }
    return imb_bfly4(y, r[0], r[1], r[2], r[3]) >> 6;
}

/* the same for the four samples (x, 0..3) of a column: the four first-pass butterflies serve all of them */
template <typename CF>
IMB_FN void imb_idct4_col(const CF *b, int b0, int x, int out[4])
{
    int r[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        r[j] = (CF)imb_bfly4(x, j == 0 ? (CF)(b0 + 32) : (b ? b[j] : 0), b ? b[j + 4] : 0, b ? b[j + 8] : 0, b ? b[j + 12] : 0);
#pragma unroll
    for (int y = 0; y < 4; y++)
        out[y] = imb_bfly4(y, r[0], r[1], r[2], r[3]) >> 6;
}

/* The residual of column x of a 4x4 block for EVERY lane in one straight line: b = the block's coefficients, or the tile's zero block
 * when it has none / only its DC counts — ff_h264_idct_add on (dc, 0, 0, ...) adds (dc + 32) >> 6 everywhere, which is
 * ff_h264_idct_dc_add (h264idct_template.c:145-160), and on all zeros adds nothing.  The one difference between the two functions is
 * kept: idct_add stores block[0] + 32 back as dctcoef before it is read, idct_dc_add computes in int. */
/* bypass (FFHIP_H264_INTRA_BYPASS): the block holds residual samples, row-major — add_pixels4_clear (h264addpx_template.c:30-48) */
template <typename CF>
IMB_FN void imb_resid4_col(const CF *b, int dc, bool dconly, int x, int out[4], bool bypass = false)
{
    if (bypass) {
#pragma unroll
        for (int y = 0; y < 4; y++)
            out[y] = b[x + 4 * y];
        return;
    }
    int r[4];
    r[0] = imb_bfly4(x, dconly ? dc + 32 : (int)(CF)(dc + 32), b[4], b[8], b[12]);
    r[0] = dconly ? r[0] : (int)(CF)r[0];
#pragma unroll
    for (int j = 1; j < 4; j++)
        r[j] = (CF)imb_bfly4(x, b[j], b[j + 4], b[j + 8], b[j + 12]);
#pragma unroll
    for (int y = 0; y < 4; y++)
        out[y] = imb_bfly4(y, r[0], r[1], r[2], r[3]) >> 6;
}

/* one 8-point pass of ff_h264_idct8_add (h264idct_template.c:69-143) */
IMB_FN void imb_idct8_1d(const int in[8], uint32_t out[8])
{
    const uint32_t a0 = (uint32_t)in[0] + (uint32_t)in[4];
    const uint32_t a2 = (uint32_t)in[0] - (uint32_t)in[4];
    const uint32_t a4 = (uint32_t)(in[2] >> 1) - (uint32_t)in[6];
    const uint32_t a6 = (uint32_t)(in[6] >> 1) + (uint32_t)in[2];
    const uint32_t b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    const int a1 = (int)(-(uint32_t)in[3] + (uint32_t)in[5] - (uint32_t)in[7] - (uint32_t)(in[7] >> 1));
    const int a3 = (int)((uint32_t)in[1] + (uint32_t)in[7] - (uint32_t)in[3] - (uint32_t)(in[3] >> 1));
    const int a5 = (int)(-(uint32_t)in[1] + (uint32_t)in[7] + (uint32_t)in[5] + (uint32_t)(in[5] >> 1));
    const int a7 = (int)((uint32_t)in[3] + (uint32_t)in[5] + (uint32_t)in[1] + (uint32_t)(in[1] >> 1));
    const uint32_t b1 = (uint32_t)(a7 >> 2) + (uint32_t)a1;
    const uint32_t b3 = (uint32_t)a3 + (uint32_t)(a5 >> 2);
    const uint32_t b5 = (uint32_t)(a3 >> 2) - (uint32_t)a5;
    const uint32_t b7 = (uint32_t)a7 - (uint32_t)(a1 >> 2);
    out[0] = b0 + b7; out[7] = b0 - b7;
    out[1] = b2 + b5; out[6] = b2 - b5;
    out[2] = b4 + b3; out[5] = b4 - b3;
    out[3] = b6 + b1; out[4] = b6 - b1;
}

/* pred8x8 (N = 8, with the one-sided "mad cow" DC variants) / pred16x16 (N = 16) sample (x, y); t / l index the tile's row above
 * and left column through TOP(i) / LEFT(i), i = -1 the corner (h264pred_template.c:389-820; modes h264pred.h:67-82) */
template <int N, typename PIX = uint8_t, class Top, class Left>
IMB_FN int imb_pred_blk(int mode, int x, int y, Top TOP, Left LEFT, int maxv = 255)
{
    const int mid = (maxv + 1) >> 1;
    constexpr int H2 = N / 2;
    if (mode == 1)
        return LEFT(y);
    if (mode == 2)
        return TOP(x);
    if (mode == 3) {
        int H = 0, V = 0;
        for (int i = 1; i <= H2; i++) {
            H += i * (TOP(H2 - 1 + i) - TOP(H2 - 1 - i));
            V += i * (LEFT(H2 - 1 + i) - (i == H2 ? TOP(-1) : LEFT(H2 - 1 - i)));
        }
        H = N == 16 ? (5 * H + 32) >> 6 : (17 * H + 16) >> 5;
        V = N == 16 ? (5 * V + 32) >> 6 : (17 * V + 16) >> 5;
        const int a = 16 * (LEFT(N - 1) + TOP(N - 1) + 1) - (H2 - 1) * (V + H);
        return imb_clip<PIX>((a + y * V + x * H) >> 5, maxv);
    }
    if (N == 16) {
        int sl = 0, st = 0;
        for (int i = 0; i < 16; i++) {
            sl += (mode == 0 || mode == 4) ? LEFT(i) : 0;
            st += (mode == 0 || mode == 5) ? TOP(i) : 0;
        }
        return mode == 0 ? (sl + st + 16) >> 5 : mode == 4 ? (sl + 8) >> 4 : mode == 5 ? (st + 8) >> 4 : mid;
    }
    const bool ut = mode == 0 || mode == 5 || mode == 7 || mode == 8, ul = mode == 0 || mode == 4 || mode >= 7;
    int t0 = 0, t1 = 0, l0 = 0, l1 = 0;
    for (int i = 0; i < 4; i++) {
        t0 += ut ? TOP(i) : 0;
        t1 += ut ? TOP(4 + i) : 0;
        l0 += ul ? LEFT(i) : 0;
        l1 += (ul && mode != 7) ? LEFT(4 + i) : 0;
    }
    int q0 = mid, q1 = mid, q2 = mid, q3 = mid;
    switch (mode) {
    case 0: q0 = (t0 + l0 + 4) >> 3; q1 = (t1 + 2) >> 2; q2 = (l1 + 2) >> 2; q3 = (t1 + l1 + 4) >> 3; break;
    case 4: q0 = q1 = (l0 + 2) >> 2; q2 = q3 = (l1 + 2) >> 2; break;
    case 5: q0 = q2 = (t0 + 2) >> 2; q1 = q3 = (t1 + 2) >> 2; break;
    case 7: q0 = (t0 + l0 + 4) >> 3; q2 = (t0 + 2) >> 2; q1 = q3 = (t1 + 2) >> 2; break;
    case 8: q0 = (t0 + 2) >> 2; q1 = (t1 + 2) >> 2; q2 = (l1 + 2) >> 2; q3 = (t1 + l1 + 4) >> 3; break;
    case 9: q0 = q1 = (l0 + 2) >> 2; break;
    case 10: q2 = q3 = (l1 + 2) >> 2; break;
    default: break;
    }
    return (y >> 2) ? ((x >> 2) ? q3 : q2) : ((x >> 2) ? q1 : q0);
}

/* ---- pred4x4 as a table: every directional rule is (w0 e[i0] + w1 e[i1] + w2 e[i2] + 2) >> 2 with weights (1, 2, 1), (2, 2, 0) or
 * (4, 0, 0) — hp_a3, hp_a2, a copy — over the edge line.  imb_p4_entry() is hp_dir_sample_e<4> solved for (i0, i1, i2, kind): bits 0-3,
 * 4-7, 8-11 the three indices, bits 12-13 the kind (0 a3, 1 a2, 2 copy): the switch yields three indices, the edge reads and the
 * arithmetic are common to all modes (two blocks with two modes share a step).  (Keeping a lane's nine entries in five registers
 * instead was no faster and cost registers the prefetches need.) */
IMB_FN uint32_t imb_p4_a3(int a, int b, int c) { return (uint32_t)(a | b << 4 | c << 8); }
IMB_FN uint32_t imb_p4_a2(int a, int b) { return (uint32_t)(a | b << 4 | b << 8 | 1 << 12); }
IMB_FN uint32_t imb_p4_cp(int a) { return (uint32_t)(a | a << 4 | a << 8 | 2 << 12); }
IMB_FN uint32_t imb_p4_entry(int mode, int x, int y)
{
    switch (mode) {
    case 0: return imb_p4_cp(5 + x);
    case 1: return imb_p4_cp(3 - y);
    case 3: {
        const int i = x + y;
        return i < 6 ? imb_p4_a3(5 + i, 6 + i, 7 + i) : imb_p4_a3(11, 12, 12);
    }
    case 4: {
        const int i = 3 - y + x;
        return imb_p4_a3(i, i + 1, i + 2);
    }
    case 5: {
        const int d = 2 * x - y, h = d >> 1;
        return d < 0 ? imb_p4_a3(4 + d, 5 + d, 6 + d) : (d & 1) ? imb_p4_a3(4 + h, 5 + h, 6 + h) : imb_p4_a2(4 + h, 5 + h);
    }
    case 6: {
        const int d = 2 * y - x, h = d >> 1;
        return d < 0 ? imb_p4_a3(2 - d, 3 - d, 4 - d) : (d & 1) ? imb_p4_a3(4 - h, 3 - h, 2 - h) : imb_p4_a2(4 - h, 3 - h);
    }
    case 7: {
        const int i = (y >> 1) + x;
        return (y & 1) ? imb_p4_a3(5 + i, 6 + i, 7 + i) : imb_p4_a2(5 + i, 6 + i);
    }
    case 8: {
        const int i = 2 * y + x, j = 3 - (i >> 1);
        return i >= 6 ? imb_p4_cp(0) : i == 5 ? imb_p4_a3(1, 0, 0) : (i & 1) ? imb_p4_a3(j, j - 1, j - 2) : imb_p4_a2(j, j - 1);
    }
    default: return 0; /* the DC modes: not a table rule */
    }
}

/* ... and solved once more, for the TILE: entry [(tr * 12 + mode) * 16 + 4 y + x] holds where sample (x, y) of a block finds its three
 * edge samples — offsets from the block's first sample in the tile (pitch 32), + 33, a byte each — and the kind in bits 24-25; tr: the
 * block's top-right neighbour exists (else e[9..12] = e[8], h264_mb.c:672-689).  The DC family (kind 3) carries what its one formula
 * (left + top + round) >> shift takes: bit 26 the left column, 27 the row above, 28 shift 3 (else 2), 29 the mid value (DC_128).
 * The kernel keeps the table in LDS: a step of the Intra4x4 wavefront is then the same straight line of code in every lane — read
 * three samples + the two sums, evaluate the four forms, select — where a switch over the lane's mode was nine masked branches. */
#define IMB_P4_ON  0x80000000u   /* the lane has a sample in this step */
#define IMB_P4_RES 0x40000000u   /* its block has a residual */
#define IMB_P4_TAB (2 * 12 * 16)
IMB_FN uint32_t imb_p4_tab(int idx)
{
    const int pos = idx & 15, mode = (idx >> 4) % 12, tr = (idx >> 4) / 12;
    if (mode == 2 || mode >= 9) {
        const uint32_t p = mode == 2 ? 7u : mode == 9 ? 1u : mode == 10 ? 2u : 8u;
        return 3u << 24 | p << 26 | 33u | 33u << 8 | 33u << 16;
    }
    const uint32_t code = imb_p4_entry(mode, pos & 3, pos >> 2);
    uint32_t v = (code >> 12) << 24;
    for (int j = 0; j < 3; j++) {
        const int k = (int)((code >> (4 * j)) & 15u);
        const int rel = k < 4 ? (3 - k) * 32 - 1 : k == 4 ? -33 : -32 + ((k < 9 || tr) ? k - 5 : 3);
        v |= (uint32_t)(rel + 33) << (8 * j);
    }
    return v;
}

/* pred8x8l the same way, in two tables.  imb_p8_edge_tab: entry [(tl * 2 + tr) * 32 + j] = where entry j of the low-pass filtered line
 * (hp_filter8_e; every entry is (w[lo] + 2 w[j'] + w[hi] + 2) >> 2 over the raw line, the ends and the corners of an absent neighbour
 * repeat a sample) finds its three raw samples: offsets from the block's first sample in the tile, + 33, nine bits each.  Entries the
 * block's mode does not read are computed all the same (their samples lie inside the tile; nothing reads the result).
 * imb_p8_tab: entry [mode * 64 + 8 y + x], mode 0..8 without DC = hp_dir_sample_e<8> solved for three indices into the filtered line
 * (five bits each) and the kind in bits 15-16. */
#define IMB_P8_EDGE_TAB (4 * 32)
#define IMB_P8_TAB (9 * 64)
IMB_FN uint32_t imb_p8_edge_tab(int idx)
{
    const int j = idx & 31, tr = (idx >> 5) & 1, tl = (idx >> 6) & 1;
    if (j > 24)
        return 33u | 33u << 9 | 33u << 18;
    int k[3] = { j ? j - 1 : 0, j, j < 24 ? j + 1 : 24 };
    if (j == 7 && !tl)
        k[2] = 7;
    if (j == 9 && !tl)
        k[0] = 9;
    if (j == 16 && !tr)
        k[2] = 16;
    if (j >= 17 && !tr)
        k[0] = k[1] = k[2] = 16;
    uint32_t v = 0;
    for (int q = 0; q < 3; q++)
        v |= (uint32_t)((k[q] < 8 ? (7 - k[q]) * 32 - 1 : k[q] - 41) + 33) << (9 * q);
    return v;
}
IMB_FN uint32_t imb_p8_a3(int a, int b, int c) { return (uint32_t)(a | b << 5 | c << 10); }
IMB_FN uint32_t imb_p8_a2(int a, int b) { return (uint32_t)(a | b << 5 | b << 10 | 1 << 15); }
IMB_FN uint32_t imb_p8_cp(int a) { return (uint32_t)(a | a << 5 | a << 10 | 2 << 15); }
IMB_FN uint32_t imb_p8_tab(int idx)
{
    const int mode = idx >> 6, x = idx & 7, y = (idx >> 3) & 7;
    constexpr int N = 8;
    switch (mode) {
    case 0: return imb_p8_cp(N + 1 + x);
    case 1: return imb_p8_cp(N - 1 - y);
    case 3: {
        const int i = x + y;
        return i < 2 * N - 2 ? imb_p8_a3(N + 1 + i, N + 2 + i, N + 3 + i) : imb_p8_a3(3 * N - 1, 3 * N, 3 * N);
    }
    case 4: {
        const int i = N - 1 - y + x;
        return imb_p8_a3(i, i + 1, i + 2);
    }
    case 5: {
        const int d = 2 * x - y, h = d >> 1;
        return d < 0 ? imb_p8_a3(N + d, N + d + 1, N + d + 2) : (d & 1) ? imb_p8_a3(N + h, N + h + 1, N + h + 2) : imb_p8_a2(N + h, N + h + 1);
    }
    case 6: {
        const int d = 2 * y - x, h = d >> 1;
        return d < 0 ? imb_p8_a3(N - d - 2, N - d - 1, N - d) : (d & 1) ? imb_p8_a3(N - h, N - h - 1, N - h - 2) : imb_p8_a2(N - h, N - h - 1);
    }
    case 7: {
        const int i = (y >> 1) + x;
        return (y & 1) ? imb_p8_a3(N + 1 + i, N + 2 + i, N + 3 + i) : imb_p8_a2(N + 1 + i, N + 2 + i);
    }
    case 8: {
        const int i = 2 * y + x, j = N - 1 - (i >> 1);
        return i >= 2 * N - 2 ? imb_p8_cp(0) : i == 2 * N - 3 ? imb_p8_a3(1, 0, 0) : (i & 1) ? imb_p8_a3(j, j - 1, j - 2) : imb_p8_a2(j, j - 1);
    }
    default: return 0; /* DC (mode 2): not a table rule */
    }
}
/* the three tables, one after the other, as the kernel keeps them in LDS */
#define IMB_TABS (IMB_P4_TAB + IMB_P8_EDGE_TAB + IMB_P8_TAB)
IMB_FN uint32_t imb_tab(int idx)
{
    return idx < IMB_P4_TAB ? imb_p4_tab(idx) : idx < IMB_P4_TAB + IMB_P8_EDGE_TAB ? imb_p8_edge_tab(idx - IMB_P4_TAB) : imb_p8_tab(idx - IMB_P4_TAB - IMB_P8_EDGE_TAB);
}

/* The same rules, split for a lane that produces the four samples (x0 .. x0 + 3, y): everything the rule reads is read ONCE, before
 * the lane writes (the compiler cannot hoist tile reads over tile writes itself: imb_pred_blk per sample re-read the whole edge — 64
 * byte reads for a DC, 64 for a plane — four times). */
struct ImbPred {
    int s[4], dc, a, H, V; /* s[j]: sample j's value under the horizontal / vertical rules */
};

/* a phase body is inlined into its one call site whatever the size of the caller: a body left as a call takes its captures by
 * address, i.e. through scratch memory */
#define IMB_INL __attribute__((always_inline))
#if defined(__HIP_DEVICE_COMPILE__)
/* per-lane state carried from one phase to the next: a register on the device, a slot per lane in the emulation */
#define IMB_STATE(name) uint32_t name = 0
#define IMB_AT(name, lane) name
#define IMB_STATE_N(name, n) uint32_t name[n]
#define IMB_AT_N(name, k, lane) name[k]
#else
#define IMB_STATE(name) uint32_t name[64]
#define IMB_AT(name, lane) name[lane]
#define IMB_STATE_N(name, n) uint32_t name[n][64]
#define IMB_AT_N(name, k, lane) name[k][lane]
#endif
#if defined(__HIP_DEVICE_COMPILE__)
/* a value read ahead of the selects that use it stays read ahead: the compiler otherwise sinks a tile read into the one arm that uses
 * it, which turns the select back into a masked branch with a second round trip inside */
#define IMB_PIN(v) asm volatile("" : "+v"(v))
#else
#define IMB_PIN(v) (void)(v)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define IMB_UNIFORM(v) __builtin_amdgcn_readfirstlane(v) /* the same in every lane: keep it in a scalar register, branch without masks */
#else
#define IMB_UNIFORM(v) (v)
#endif

/* the four samples are (x0 + j, y) of a row, or with COL (x0, y + j) of a column (y a multiple of 4) */
template <int N, bool COL, class Top, class Left>
IMB_FN ImbPred imb_pred_quad(int mode, int x0, int y, Top TOP, Left LEFT, int mid)
{
    constexpr int H2 = N / 2;
    ImbPred P;
    P.dc = P.a = P.H = P.V = 0;
    P.s[0] = P.s[1] = P.s[2] = P.s[3] = 0;
    if (mode == 1) {
        if (COL) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                P.s[j] = LEFT(y + j);
        } else {
            P.s[0] = P.s[1] = P.s[2] = P.s[3] = LEFT(y);
        }
    } else if (mode == 2) {
        if (COL) {
            P.s[0] = P.s[1] = P.s[2] = P.s[3] = TOP(x0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                P.s[j] = TOP(x0 + j);
        }
    } else if (mode == 3) {
        int H = 0, V = 0;
        for (int i = 1; i <= H2; i++) {
            H += i * (TOP(H2 - 1 + i) - TOP(H2 - 1 - i));
            V += i * (LEFT(H2 - 1 + i) - (i == H2 ? TOP(-1) : LEFT(H2 - 1 - i)));
        }
        P.H = N == 16 ? (5 * H + 32) >> 6 : (17 * H + 16) >> 5;
        P.V = N == 16 ? (5 * V + 32) >> 6 : (17 * V + 16) >> 5;
        P.a = 16 * (LEFT(N - 1) + TOP(N - 1) + 1) - (H2 - 1) * (P.V + P.H);
    } else if (N == 16) {
        int sl = 0, st = 0;
        for (int i = 0; i < 16; i++) {
            sl += (mode == 0 || mode == 4) ? LEFT(i) : 0;
            st += (mode == 0 || mode == 5) ? TOP(i) : 0;
        }
        P.dc = mode == 0 ? (sl + st + 16) >> 5 : mode == 4 ? (sl + 8) >> 4 : mode == 5 ? (st + 8) >> 4 : mid;
    } else {
        const bool ut = mode == 0 || mode == 5 || mode == 7 || mode == 8, ul = mode == 0 || mode == 4 || mode >= 7;
        int t0 = 0, t1 = 0, l0 = 0, l1 = 0;
        for (int i = 0; i < 4; i++) {
            t0 += ut ? TOP(i) : 0;
            t1 += ut ? TOP(4 + i) : 0;
            l0 += ul ? LEFT(i) : 0;
            l1 += (ul && mode != 7) ? LEFT(4 + i) : 0;
        }
        int q0 = mid, q1 = mid, q2 = mid, q3 = mid;
        switch (mode) {
        case 0: q0 = (t0 + l0 + 4) >> 3; q1 = (t1 + 2) >> 2; q2 = (l1 + 2) >> 2; q3 = (t1 + l1 + 4) >> 3; break;
        case 4: q0 = q1 = (l0 + 2) >> 2; q2 = q3 = (l1 + 2) >> 2; break;
        case 5: q0 = q2 = (t0 + 2) >> 2; q1 = q3 = (t1 + 2) >> 2; break;
        case 7: q0 = (t0 + l0 + 4) >> 3; q2 = (t0 + 2) >> 2; q1 = q3 = (t1 + 2) >> 2; break;
        case 8: q0 = (t0 + 2) >> 2; q1 = (t1 + 2) >> 2; q2 = (l1 + 2) >> 2; q3 = (t1 + l1 + 4) >> 3; break;
        case 9: q0 = q1 = (l0 + 2) >> 2; break;
        case 10: q2 = q3 = (l1 + 2) >> 2; break;
        default: break;
        }
        P.dc = (y >> 2) ? ((x0 >> 2) ? q3 : q2) : ((x0 >> 2) ? q1 : q0);
    }
    return P;
}

template <typename PIX>
IMB_FN int imb_pred_px(int mode, const ImbPred &P, int j, int x, int y, int maxv)
{
    return (mode == 1 || mode == 2) ? P.s[j] : mode == 3 ? imb_clip<PIX>((P.a + y * P.V + x * P.H) >> 5, maxv) : P.dc;
}

/*
 * The macroblock, phase by phase.  X.run(body) runs body(lane) for the 64 lanes and synchronises the tile.
 * T holds the neighbours (unavailable ones as 0) on entry and the reconstructed macroblock on return.
 * Luma and chroma share nothing but the record: `parts` selects the planes a caller wants (the kernel runs a picture's chroma as a
 * wavefront of its own beside the luma one when it has the room: a macroblock step of the luma chain is a quarter shorter without it).
 */
template <typename PIX, class X>
IMB_FN void imb_reconstruct(X &x, ImbTileT<PIX> &T, const FFHipH264IntraMB &R, const typename ImbCoef<PIX>::T *coefs /* the macroblock's run */,
                            const uint32_t *p4tab /* imb_tab(0 .. IMB_TABS - 1) */, int maxv = 255 /* (1 << bit_depth) - 1 */,
                            int parts = 3 /* 1: the luma samples, 2: the two chroma planes; the same in every lane */)
{
    typedef typename ImbCoef<PIX>::T CF;
    const int mid = (maxv + 1) >> 1;
    const ImbHead H = { IMB_UNIFORM((int)R.type), IMB_UNIFORM((int)R.cbp), IMB_UNIFORM((int)R.flags), (uint32_t)IMB_UNIFORM((int)R.blocks) };
    const bool byp = (H.flags & FFHIP_H264_INTRA_BYPASS) != 0; /* the transform bypass: the run holds residual samples (the packer made them so) */
    if (H.type == FFHIP_H264_INTRA_PCM) {
        /* the samples as they stand in the bitstream: 256 luma, 64 Cb, 64 Cr (h264_mb_template.c:98-150; above 8 bits the host side
         * has unpacked the bit_depth-bit fields into uint16_t) */
        const PIX *pcm = reinterpret_cast<const PIX *>(coefs);
        x.run([&](int lane) IMB_INL {
            if (parts & 1)
                for (int j = 0; j < 4; j++)
                    T.y[imb_yi(lane >> 2, 4 * (lane & 3) + j)] = pcm[16 * (lane >> 2) + 4 * (lane & 3) + j];
            if (lane < 32 && (parts & 2))
                for (int j = 0; j < 4; j++)
                    T.c[lane >> 4][imb_ci((lane >> 1) & 7, 4 * (lane & 1) + j)] = pcm[256 + 64 * (lane >> 4) + 4 * (lane & 15) + j];
        });
        return;
    }
    /* ---- chroma: pred8x8 on both planes, then chroma_dc_dequant_idct + idct_add8 when cbp & 0x30; lane = 4 samples of a row.
     *      Intra16x16 also dequantises its 16 luma DCs here (lanes 32..47), the 8x8 transform runs its first pass (lanes 32..63) ---- */
    x.run([&](int lane) IMB_INL {
        if (lane < 32) {
            if (!(parts & 2))
                return;
            /* lane = column (lane & 3) of chroma block k = (lane >> 2) & 3 of plane p: everything is read before the column is written */
            const int p = lane >> 4, k = (lane >> 2) & 3, xc = 4 * (k & 1) + (lane & 3), y0 = 4 * (k >> 1);
            const PIX *tc = T.c[p];
            auto TOP = [&](int i) { return (int)tc[imb_ci(-1, i)]; };
            auto LEFT = [&](int i) { return (int)tc[imb_ci(i, -1)]; };
            /* the residual in one straight line for every lane (imb_resid4_col): a lane whose block is not in the run works on the
             * tile's zero block — masked branches per case cost more than the arithmetic they skip */
            int dc = 0;
            bool dconly = false;
            const CF *b = T.zero;
            if (H.cbp & 0x30) {
                /* the plane's blocks follow the luma blocks in the run (imb_block, with the luma part counted once in scalar registers) */
                const uint32_t cb = H.blocks >> 16;
                const CF *c0 = coefs + (sizeof(CF) == 4 && (H.flags & FFHIP_H264_INTRA_LUMA_DC) ? 16 : 0) +
                               imb_popc(H.blocks & 0xFFFFu) * (H.type == FFHIP_H264_INTRA_8x8 ? 64 : 16);
                /* chroma_dc_dequant_idct (h264idct_template.c:323-345) on the four DCs of the plane; taken when the plane has a DC block */
                int d4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int bit = 4 * p + q;
                    const CF *bq = ((cb >> bit) & 1u) ? c0 + 16 * imb_popc(cb & ((1u << bit) - 1u)) : T.zero;
                    d4[q] = bq[0];
                }
                const uint32_t qm = (uint32_t)R.qmul[1 + p];
                uint32_t a = (uint32_t)d4[0], bb = (uint32_t)d4[1], c = (uint32_t)d4[2], d = (uint32_t)d4[3];
                const uint32_t e = a - bb;
                a = a + bb; bb = c - d; c = c + d;
                const uint32_t v = k == 0 ? a + c : k == 1 ? e + bb : k == 2 ? a - c : e - bb;
                const int own = k == 0 ? d4[0] : k == 1 ? d4[1] : k == 2 ? d4[2] : d4[3];
                dc = ((H.flags >> p) & FFHIP_H264_INTRA_CB_DC) ? (int)(CF)((int)(v * qm) >> 7) : own;
                const int bit = 4 * p + k;
                const bool full = R.nnz[16 + bit] != 0 && ((cb >> bit) & 1u);
                b = full ? c0 + 16 * imb_popc(cb & ((1u << bit) - 1u)) : T.zero;
                dconly = !full && dc != 0;
            }
            const int cmode = IMB_UNIFORM((int)R.chroma_pred);
            const ImbPred P = imb_pred_quad<8, true>(cmode, xc, y0, TOP, LEFT, mid);
            int res[4];
            imb_resid4_col(b, dc, dconly, lane & 3, res, byp);
#pragma unroll
            for (int j = 0; j < 4; j++)
                T.c[p][imb_ci(y0 + j, xc)] = (PIX)imb_fin<PIX>(imb_pred_px<PIX>(cmode, P, j, xc, y0 + j, maxv) + res[j], maxv, byp);
        } else if (!(parts & 1)) {
            return;
        } else if (H.type == FFHIP_H264_INTRA_8x8) {
            /* first pass of the four 8x8 inverse transforms (they depend on the coefficients alone): lane 32 + 8 q + j = transform j
             * of block 4 q, working on block[j + 8 k], results stored as int16 */
            const int q = (lane - 32) >> 3, j = lane & 7, nnz = R.nnz[4 * q];
            const CF *b = nnz ? imb_block(H, coefs, 4 * q) : nullptr;
            if (b && byp) {
                /* add_pixels8_clear: the samples of column j where the steps read their transform's output */
#pragma unroll
                for (int k = 0; k < 8; k++)
                    T.t8[q][8 * j + k] = b[j + 8 * k];
            } else if (b && !(nnz == 1 && b[0])) {
                int in[8];
                uint32_t out[8];
#pragma unroll
                for (int k = 0; k < 8; k++)
                    in[k] = b[j + 8 * k];
                if (j == 0)
                    in[0] = (CF)(in[0] + 32);
                imb_idct8_1d(in, out);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    T.t8[q][j + 8 * k] = (CF)out[k];
            }
        } else if (lane < 48 && H.type == FFHIP_H264_INTRA_16x16 && (H.flags & FFHIP_H264_INTRA_LUMA_DC)) {
            /* luma_dc_dequant_idct (h264idct_template.c:259-293): lane 32 + o computes output o of the 4x4 Hadamard */
            const int o = lane - 32, i = o & 3, w = o >> 2; /* second-pass column i, output w of { z0+z3, z1+z2, z1-z2, z0-z3 } */
            int t[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                /* above 8 bits the sixteen DCs lead the run (imb_block) */
                const int a = sizeof(CF) == 4 ? (int)coefs[4 * r] : R.luma_dc[4 * r], b = sizeof(CF) == 4 ? (int)coefs[4 * r + 1] : R.luma_dc[4 * r + 1];
                const int c = sizeof(CF) == 4 ? (int)coefs[4 * r + 2] : R.luma_dc[4 * r + 2], d = sizeof(CF) == 4 ? (int)coefs[4 * r + 3] : R.luma_dc[4 * r + 3];
                const int z0 = a + b, z1 = a - b, z2 = c - d, z3 = c + d;
                t[r] = i == 0 ? z0 + z3 : i == 1 ? z0 - z3 : i == 2 ? z1 - z2 : z1 + z2;
            }
            const uint32_t z0 = (uint32_t)t[0] + (uint32_t)t[2], z1 = (uint32_t)t[0] - (uint32_t)t[2];
            const uint32_t z2 = (uint32_t)t[1] - (uint32_t)t[3], z3 = (uint32_t)t[1] + (uint32_t)t[3];
            const uint32_t v = w == 0 ? z0 + z3 : w == 1 ? z1 + z2 : w == 2 ? z1 - z2 : z0 - z3;
            /* output[16 * {0, 1, 4, 5}[w] + x_offset[i]], x_offset = {0, 32, 128, 160}: as a block number */
            const int blk = (w == 0 ? 0 : w == 1 ? 1 : w == 2 ? 4 : 5) + (i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 8 : 10);
            T.dcq[blk] = (CF)((int)(v * (uint32_t)IMB_UNIFORM(R.qmul[0]) + 128) >> 8);
        }
    });

    if (!(parts & 1))
        return;
    if (H.type == FFHIP_H264_INTRA_16x16) {
        /* pred16x16 + idct_add16intra (h264idct_template.c:191-200): lane = column (lane & 3) of block (lane >> 2) */
        x.run([&](int lane) IMB_INL {
            const int i = lane >> 2, y0 = imb_by(i), xc = imb_bx(i) + (lane & 3);
            auto TOP = [&](int k) { return (int)T.y[imb_yi(-1, k)]; };
            auto LEFT = [&](int k) { return (int)T.y[imb_yi(k, -1)]; };
            const CF *bp = imb_block(H, coefs, i);
            const CF *bs = bp ? bp : T.zero;
            const int dc = (H.flags & FFHIP_H264_INTRA_LUMA_DC) ? T.dcq[i] : (int)bs[0];
            const bool full = R.nnz[i] != 0, dconly = !full && dc != 0;
            const int lmode = IMB_UNIFORM((int)R.pred16);
            const ImbPred P = imb_pred_quad<16, true>(lmode, xc, y0, TOP, LEFT, mid);
            int res[4];
            imb_resid4_col(full ? bs : T.zero, dc, dconly, lane & 3, res, byp);
#pragma unroll
            for (int j = 0; j < 4; j++)
                T.y[imb_yi(y0 + j, xc)] = (PIX)imb_fin<PIX>(imb_pred_px<PIX>(lmode, P, j, xc, y0 + j, maxv) + res[j], maxv, byp);
        });
        return;
    }

    if (H.type == FFHIP_H264_INTRA_4x4) {
        /* A 4x4 block reads its left, upper-left, upper and — where the decoding order has it — upper-right neighbours, so the blocks
         * on an anti-diagonal x + 2 y = t of the 4 x 4 grid are independent: ten steps of one or two blocks instead of sixteen
         * (results cannot differ: each block sees exactly the samples it sees in block order).  A step is ONE phase: a lane reads
         * the edge samples its rule needs straight from the tile (they lie outside the blocks written in this step) and adds its
         * own sample of the residual. */
        /* The residuals do not depend on the prediction: all sixteen blocks' in ONE phase ahead of the ten steps, lane = column x of
         * block i (four first-pass butterflies serve the column's four samples: 8 butterflies per lane instead of 5 per sample inside a
         * step), parked in t8 (unused by this macroblock type) as res[16 i + 4 y + x]. */
        /* ... and a lane of the steps collects what its ten steps need: the table entry of its sample under the block's mode
         * (imb_p4_tab) with "has a residual" — a register per step: the record lives in LDS and a fence ends every step, so read in
         * place it was a dependent round trip at the head of each step, and the mode a masked branch per rule (measured, round 4:
         * 1,630 cycles per step, of which the tile's round trip is a tenth).  (The host emulation runs the lanes one after another:
         * a slot per lane.) */
        IMB_STATE_N(inf, 10);
        x.run([&](int lane) IMB_INL {
#pragma unroll
            for (int t = 0; t < 10; t++) {
                const int y0 = t <= 3 ? 0 : (t - 2) >> 1, y4 = y0 + ((lane >> 4) & 1), x4 = t - 2 * y4;
                const bool on = lane < 32 && y4 <= 3 && x4 >= 0 && x4 <= 3;
                const int bi = on ? (x4 & 1) | (y4 & 1) << 1 | (x4 >> 1) << 2 | (y4 >> 1) << 3 : 0;
                const int mode = R.pred4[bi] < 11 ? R.pred4[bi] : 11, tr = ((R.topright_avail << bi) & 0x8000) ? 12 : 0;
                const uint32_t e = p4tab[(tr + mode) * 16 + (lane & 15)] | (R.nnz[bi] ? IMB_P4_RES : 0u) | IMB_P4_ON;
                IMB_AT_N(inf, t, lane) = on ? e : 0u;
            }
            const int i = lane >> 2, xx = lane & 3, nnz = R.nnz[i];
            int *res = &T.t8[0][0] + 16 * i;
            const CF *bp = imb_block(H, coefs, i); /* travels whenever nnz != 0 */
            const CF *bs = nnz && bp ? bp : T.zero;
            const int dc = bs[0];
            const bool dconly = nnz == 1 && dc;
            int out[4];
            imb_resid4_col(dconly ? T.zero : bs, dc, dconly, xx, out, byp);
#pragma unroll
            for (int y = 0; y < 4; y++)
                res[4 * y + xx] = out[y];
        });
#pragma unroll
        for (int t = 0; t < 10; t++) {
            x.run([&](int lane) IMB_INL {
                const uint32_t ent = IMB_AT_N(inf, t, lane);
                if (!(ent & IMB_P4_ON))
                    return;
                const int y0 = t <= 3 ? 0 : (t - 2) >> 1, y4 = y0 + ((lane >> 4) & 1), x4 = t - 2 * y4;
                const int i = (x4 & 1) | (y4 & 1) << 1 | (x4 >> 1) << 2 | (y4 >> 1) << 3;
                const int xx = lane & 3, yy = (lane >> 2) & 3;
                PIX *o = &T.y[imb_yi(4 * y4, 4 * x4)];
                /* everything any form reads, unconditionally: one round trip to the tile */
                int a = o[(int)(ent & 255u) - 33], b = o[(int)((ent >> 8) & 255u) - 33], c = o[(int)((ent >> 16) & 255u) - 33];
                const int sl = (int)o[-1] + (int)o[31] + (int)o[63] + (int)o[95], st = (int)o[-32] + (int)o[-31] + (int)o[-30] + (int)o[-29];
                int res = (&T.t8[0][0])[16 * i + 4 * yy + xx];
                IMB_PIN(a);
                IMB_PIN(b);
                IMB_PIN(c);
                IMB_PIN(res);
                const uint32_t kind = (ent >> 24) & 3u;
                const int dir = kind == 0 ? (a + 2 * b + c + 2) >> 2 : kind == 1 ? (a + b + 1) >> 1 : a;
                const int sh = (ent >> 28) & 1u ? 3 : 2;
                const int dcv = (((ent >> 26) & 1u ? sl : 0) + ((ent >> 27) & 1u ? st : 0) + ((ent >> 29) & 1u ? 4 * mid : 1 << (sh - 1))) >> sh;
                int v = kind == 3 ? dcv : dir;
                const int vr = imb_fin<PIX>(v + res, maxv, byp);
                v = (ent & IMB_P4_RES) ? vr : v;
                o[32 * yy + xx] = (PIX)v;
            });
        }
        return;
    }

    /* Intra4x4 with the 8x8 transform: pred8x8l over the low-pass filtered edge line (evaluated where it is read, from the raw
     * samples in the tile), idct8_add's second pass / idct8_dc_add: one phase per block */
    /* the second pass of the four transforms, in place, once (a block's step then adds t8[8 x + y]; inside the step every sample's lane
     * ran the whole 8-point pass of its column to keep one output) */
    x.run([&](int lane) IMB_INL {
        if (lane >= 32)
            return;
        const int q = lane >> 3, xx = lane & 7, nnz = R.nnz[4 * q];
        const CF *b = nnz ? imb_block(H, coefs, 4 * q) : nullptr;
        if (!b || (nnz == 1 && b[0]) || byp)
            return;
        int in[8];
        uint32_t out[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            in[k] = T.t8[q][8 * xx + k];
        imb_idct8_1d(in, out);
#pragma unroll
        for (int k = 0; k < 8; k++)
            T.t8[q][8 * xx + k] = (int)out[k] >> 6;
    });
    for (int i = 0; i < 16; i += 4) {
        const int bx = imb_bx(i), by = imb_by(i), mode = IMB_UNIFORM((int)R.pred4[i]), nnz = IMB_UNIFORM((int)R.nnz[i]);
        const bool tl = (IMB_UNIFORM((int)R.topleft_avail) << i) & 0x8000, tr = (IMB_UNIFORM((int)R.topright_avail) << i) & 0x4000;
        const CF *b = nnz ? imb_block(H, coefs, i) : nullptr;
        const int dc = b ? IMB_UNIFORM((int)b[0]) : 0;
        const bool dconly = nnz == 1 && dc, full = nnz && !dconly;
        /* the filtered line once per block (25 lanes), then the rule reads it: evaluated where it was read, a DC cost every lane 16
         * filtered entries of three raw reads each */
        x.run([&](int lane) IMB_INL {
            if (lane >= 25)
                return;
            const uint32_t ent = p4tab[IMB_P4_TAB + (tl ? 64 : 0) + (tr ? 32 : 0) + lane];
            const PIX *o = &T.y[imb_yi(by, bx)];
            const int a = o[(int)(ent & 511u) - 33], bq = o[(int)((ent >> 9) & 511u) - 33], c = o[(int)((ent >> 18) & 511u) - 33];
            T.edge[lane] = (a + 2 * bq + c + 2) >> 2;
        });
        x.run([&](int lane) IMB_INL {
            auto ef = [&](int k) { return T.edge[k]; };
            const int xx = lane & 7, yy = lane >> 3;
            int v;
            if (mode == 2 || mode >= 9) {
                v = hp_dir_dc_e<8>(mode, ef, mid);
            } else {
                const uint32_t ent = p4tab[IMB_P4_TAB + IMB_P8_EDGE_TAB + 64 * mode + lane];
                const int a = T.edge[ent & 31u], bq = T.edge[(ent >> 5) & 31u], c = T.edge[(ent >> 10) & 31u];
                const uint32_t kind = ent >> 15;
                v = kind == 0 ? (a + 2 * bq + c + 2) >> 2 : kind == 1 ? (a + bq + 1) >> 1 : a;
            }
            if (full) {
                /* transform xx worked on block[8 xx + k], output yy goes to dst[xx + yy * stride] */
                v = imb_fin<PIX>(v + T.t8[i >> 2][8 * xx + yy], maxv, byp);
            } else if (dconly) {
                v = imb_clip<PIX>(v + ((dc + 32) >> 6), maxv);
            }
            T.y[imb_yi(by + yy, bx + xx)] = (PIX)v;
        });
    }
}
/* ================================================================================================================================
 * 4:2:2 (round 4): the two 8 x 16 chroma planes of an intra macroblock — hl_decode_mb() at chroma_format_idc 2
 * (libavcodec/h264_mb_template.c:151-262): pred8x8[chroma_pred_mode] is a pred8x16 function there (h264pred.c:480-512,
 * h264pred_template.c:567-817), the residual is chroma422_dc_dequant_idct per plane (h264idct_template.c:295-321, qmul =
 * dequant4_coeff[1 + p][chroma_qp[p] + 3][0]: h264_mb_template.c:230-245) + idct_add8_422 (:230-252: eight blocks per plane, the cache
 * rows running on downwards).  The luma plane of such a macroblock is a luma-only record of the wavefront above; the chroma planes
 * are a record and a wavefront of their own (k_h264_intra_c422: chroma prediction reads the left, upper-left and upper neighbours only).
 * ONE phase: lane = column (lane & 3) of block (lane >> 2) & 7 of plane lane >> 5; prediction reads samples outside the macroblock only.
 * ================================================================================================================================ */
/* (the record: FFHipH264IntraC422 in include/ffhip.h) */

template <typename PIX>
struct ImbTileC422 {
    PIX c[2][17 * 16];    /* sample (r, c), r = -1..15, c = -4..11, at [(r + 1) * 16 + c + 4] (imb_ci) */
    typename ImbCoef<PIX>::T zero[16];
};

/* pred8x16 sample column (xc, y0 .. y0 + 3) of block k: modes as H264PredContext.pred8x8[] at chroma_format_idc 2 */
template <typename PIX, class Top, class Left>
IMB_FN void imb_pred8x16_col(int mode, int k, int xc, int y0, Top TOP, Left LEFT, int maxv, int out[4])
{
    const int mid = (maxv + 1) >> 1;
    if (mode == 1) {
        for (int j = 0; j < 4; j++)
            out[j] = LEFT(y0 + j);
        return;
    }
    if (mode == 2) {
        out[0] = out[1] = out[2] = out[3] = TOP(xc);
        return;
    }
    if (mode == 3) { /* pred8x16_plane (h264pred_template.c:781-817) */
        int H = 0, V = 0;
        for (int i = 1; i <= 4; i++)
            H += i * (TOP(3 + i) - TOP(3 - i));
        for (int i = 1; i <= 8; i++)
            V += i * (LEFT(7 + i) - (i == 8 ? TOP(-1) : LEFT(7 - i)));
        H = (17 * H + 16) >> 5;
        V = (5 * V + 32) >> 6;
        const int a = 16 * (LEFT(15) + TOP(7) + 1) - 7 * V - 3 * H;
        for (int j = 0; j < 4; j++)
            out[j] = imb_clip<PIX>((a + (y0 + j) * V + xc * H) >> 5, maxv);
        return;
    }
    /* the DC family: one value per 4 x 4 cell (r = cell row 0..3, cx = cell column) */
    const int r = k >> 1, cx = k & 1;
    const bool use_t = mode == 0 || mode == 5 || mode == 7 || mode == 8, use_l = mode == 0 || mode == 4 || mode >= 7;
    int t0 = 0, t1 = 0, l0 = 0, lr = 0;
    for (int i = 0; i < 4; i++) {
        t0 += use_t ? TOP(i) : 0;
        t1 += use_t ? TOP(4 + i) : 0;
        l0 += use_l ? LEFT(i) : 0;
        lr += (use_l && mode != 7) ? LEFT(4 * r + i) : 0;
    }
    int v = mid;
    switch (mode) {
    case 0: case 8: /* pred8x16_dc (:650-695); 8 = mad_cow_dc_0lt: cell 0 from the top alone */
        v = cx == 0 ? (r == 0 ? (mode == 0 ? (t0 + l0 + 4) >> 3 : (t0 + 2) >> 2) : (lr + 2) >> 2)
                    : (r == 0 ? (t1 + 2) >> 2 : (t1 + lr + 4) >> 3);
        break;
    case 4: case 9: case 10: /* left_dc (:567-571); 9 = _l00: the cells of rows 4..7 stay mid; 10 = _0l0: those of rows 0..3 (:725-749) */
        v = ((mode == 9 && r == 1) || (mode == 10 && r == 0)) ? mid : (lr + 2) >> 2;
        break;
    case 5: case 7: /* top_dc (:599-603); 7 = _l0t: cell 0 is pred4x4_dc */
        v = cx == 0 ? (t0 + 2) >> 2 : (t1 + 2) >> 2;
        if (mode == 7 && k == 0)
            v = (t0 + l0 + 4) >> 3;
        break;
    default: break; /* 6: DC_128 */
    }
    out[0] = out[1] = out[2] = out[3] = v;
}

template <typename PIX, class X>
IMB_FN void imb_c422_reconstruct(X &x, ImbTileC422<PIX> &T, const FFHipH264IntraC422 &R, const typename ImbCoef<PIX>::T *coefs, int maxv)
{
    typedef typename ImbCoef<PIX>::T CF;
    if (R.type == FFHIP_H264_INTRA_PCM) {
        /* 8 x 16 samples of Cb, then of Cr, as they stand in the bitstream (h264_mb_template.c:98-150 with block_h = 16) */
        const PIX *pcm = reinterpret_cast<const PIX *>(coefs);
        x.run([&](int lane) IMB_INL {
            const int p = lane >> 5, r = (lane >> 1) & 15, c0 = 4 * (lane & 1);
            for (int j = 0; j < 4; j++)
                T.c[p][imb_ci(r, c0 + j)] = pcm[128 * p + 8 * r + c0 + j];
        });
        return;
    }
    x.run([&](int lane) IMB_INL {
        const int p = lane >> 5, k = (lane >> 2) & 7, xx = lane & 3, xc = 4 * (k & 1) + xx, y0 = 4 * (k >> 1);
        const PIX *tc = T.c[p];
        auto TOP = [&](int i) { return (int)tc[imb_ci(-1, i)]; };
        auto LEFT = [&](int i) { return (int)tc[imb_ci(i, -1)]; };
        int pred[4];
        imb_pred8x16_col<PIX>((int)R.chroma_pred, k, xc, y0, TOP, LEFT, maxv, pred);
        int dc = 0;
        bool dconly = false;
        const CF *b = T.zero;
        if (R.cbp & 0x30) {
            const uint32_t trav = (uint32_t)R.blocks;
            auto blk = [&](int bit) -> const CF * { return ((trav >> bit) & 1u) ? coefs + 16 * imb_popc(trav & ((1u << bit) - 1u)) : T.zero; };
            int d[8];
            for (int q = 0; q < 8; q++)
                d[q] = blk(8 * p + q)[0];
            const int own = d[k];
            if ((R.flags >> p) & 1) {
                /* chroma422_dc_dequant_idct: block q = row q >> 1, column q & 1 of the 4 x 2 DC array */
                uint32_t t[8];
                for (int i = 0; i < 4; i++) {
                    t[2 * i] = (uint32_t)d[2 * i] + (uint32_t)d[2 * i + 1];
                    t[2 * i + 1] = (uint32_t)d[2 * i] - (uint32_t)d[2 * i + 1];
                }
                const int i = k & 1, w = k >> 1;
                const uint32_t z0 = t[i] + t[4 + i], z1 = t[i] - t[4 + i], z2 = t[2 + i] - t[6 + i], z3 = t[2 + i] + t[6 + i];
                const uint32_t v = w == 0 ? z0 + z3 : w == 1 ? z1 + z2 : w == 2 ? z1 - z2 : z0 - z3;
                dc = (int)(CF)((int)(v * (uint32_t)R.qmul[p] + 128u) >> 8);
            } else {
                dc = own;
            }
            const int bit = 8 * p + k;
            const bool full = ((R.full >> bit) & 1) && ((trav >> bit) & 1u);
            b = full ? blk(bit) : T.zero;
            dconly = !full && dc != 0;
        }
        /* (the transform bypass, FFHIP_H264_INTRA_BYPASS in the record's flags: the run holds residual samples, added modulo the sample type) */
        const bool byp = (R.flags & FFHIP_H264_INTRA_BYPASS) != 0;
        int res[4];
        imb_resid4_col(b, dc, dconly, xx, res, byp);
        for (int j = 0; j < 4; j++)
            T.c[p][imb_ci(y0 + j, xc)] = (PIX)imb_fin<PIX>(pred[j] + res[j], maxv, byp);
    });
}
#endif
