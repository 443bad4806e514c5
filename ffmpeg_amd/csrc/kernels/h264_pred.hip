/*
 * h264_pred.hip — H.264 intra prediction, 4:2:0, 8 bits and (templates on the sample type) 9 / 10 / 12 / 14, batched and in place: H264PredContext
 * (libavcodec/h264pred.h:92-116; bodies libavcodec/h264pred_template.c, table libavcodec/h264pred.c:448-538).
 *
 * A block's neighbours are staged once into LDS as its "edge line" e[] = the left column bottom-up, the corner, the row above
 * running on into the top-right block — the samples met walking up the left side, round the corner and along the top.  The nine
 * directional modes of pred4x4 and pred8x8l are then one set of per-sample rules over that line (pred8x8l over the low-pass
 * filtered line, h264pred_template.c:822-856); pred8x8 / pred16x16 reduce the line to four quadrant DCs or a plane (a, H, V).
 * One thread writes 4 samples of a row; 256 / (N*N/4) blocks share a workgroup.  Only the neighbours the mode's C function reads
 * are loaded (a picture-edge block has no row above to read).  Blocks of a launch are independent: intra prediction chains
 * through the reconstruction, so the decoder's wavefront orders the launches.
 *
 * The lossless _add members (h264pred_template.c:1104-1330) integrate the residual along the prediction direction in wrapping
 * 8-bit arithmetic: one thread per column (VERT) or row (HOR), which also clears the coefficients it consumed.
 */
#include <type_traits>

#include "common.h"
#include "h264_intra_mb.h"
#include "h264_kernels.h"

static_assert(sizeof(FFHipH264Pred) == 12, "FFHipH264Pred is a 12-byte record");

/* hp_a2 / hp_a3 / hp_need / hp_dir_sample: the per-sample rules over the edge line, shared with the picture pipeline's intra
 * reconstruction (h264_intra_mb.h) */

__device__ __forceinline__ void hp_store4(uint16_t *d, const int *v) /* 16-bit samples: two dwords when aligned */
{
    if (!(reinterpret_cast<uintptr_t>(d) & 7)) {
        *reinterpret_cast<uint2 *>(d) = make_uint2((uint32_t)v[0] | (uint32_t)v[1] << 16, (uint32_t)v[2] | (uint32_t)v[3] << 16);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            d[j] = (uint16_t)v[j];
    }
}
__device__ __forceinline__ void hp_store4(uint8_t *d, const int *v)
{
    if (!(reinterpret_cast<uintptr_t>(d) & 3)) {
        *reinterpret_cast<uint32_t *>(d) = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            d[j] = (uint8_t)v[j];
    }
}

/* pred4x4 (L8 = false, N = 4) and pred8x8l (L8 = true, N = 8) */
template <bool L8, typename P>
__global__ __launch_bounds__(256) void k_h264_pred_dir(uint8_t *plane, ptrdiff_t stride_b, const FFHipH264Pred *blocks, int n, int bd)
{
    const ptrdiff_t stride = stride_b / (ptrdiff_t)sizeof(P); /* offsets and strides arrive in bytes, samples are P */
    constexpr int N = L8 ? 8 : 4, ITEMS = N * N / 4, RPB = 256 / ITEMS, LINE = 3 * N + 1, QW = N / 4;
    __shared__ int raw[RPB][LINE + 1];
    __shared__ int flt[L8 ? RPB : 1][LINE + 1];
    const int r = threadIdx.x / ITEMS, it = threadIdx.x % ITEMS, b = blockIdx.x * RPB + r;
    const bool valid = b < n;
    FFHipH264Pred k = {};
    if (valid)
        k = blocks[b];
    const int mode = k.mode;
    const unsigned need = valid ? hp_need(mode) : 0u;
    const bool tl = k.flags & FFHIP_H264_PRED_TOPLEFT, tr = k.flags & FFHIP_H264_PRED_TOPRIGHT;
    P *src = reinterpret_cast<P *>(plane + k.offset);
    for (int j = it; j < LINE; j += ITEMS) {
        int v = 0;
        if (j < N) {
            if (need & 1)
                v = src[(ptrdiff_t)(N - 1 - j) * stride - 1];
        } else if (j == N) {
            if ((need & 4) || (L8 && tl && (need & 3)))
                v = src[-stride - 1];
        } else if (j < 2 * N + 1) {
            if (need & 2)
                v = src[j - N - 1 - stride];
        } else if (L8) {
            if (tr && ((need & 8) || (j == 2 * N + 1 && (need & 2))))
                v = src[j - N - 1 - stride];
        } else if (need & 8) {
            v = (k.flags & FFHIP_H264_PRED_TR_SPLAT) ? src[3 - stride] : reinterpret_cast<const P *>(plane + k.aux)[j - 2 * N - 1];
        }
        raw[r][j] = v;
    }
    __syncthreads();
    const int *e = raw[r];
    if (L8) {
        /* PREDICT_8x8_LOAD_LEFT / _TOP / _TOPRIGHT / _TOPLEFT (h264pred_template.c:822-856) */
        const int *w = raw[r];
        for (int j = it; j < LINE; j += ITEMS) {
            int v = 0;
            if (j < 8) {
                if (need & 1)
                    v = j == 7 ? hp_a3(tl ? w[8] : w[7], w[7], w[6]) : j == 0 ? (w[1] + 3 * w[0] + 2) >> 2 : hp_a3(w[j + 1], w[j], w[j - 1]);
            } else if (j == 8) {
                if (need & 4)
                    v = hp_a3(w[7], w[8], w[9]);
            } else if (j < 17) {
                if (need & 2)
                    v = j == 9 ? hp_a3(tl ? w[8] : w[9], w[9], w[10]) : j == 16 ? hp_a3(tr ? w[17] : w[16], w[16], w[15]) : hp_a3(w[j - 1], w[j], w[j + 1]);
            } else if (need & 8) {
                v = !tr ? w[16] : j == 24 ? (w[23] + 3 * w[24] + 2) >> 2 : hp_a3(w[j - 1], w[j], w[j + 1]);
            }
            flt[r][j] = v;
        }
        __syncthreads();
        e = flt[r];
    }
    if (!valid)
        return;
    int dc = 1 << (bd - 1); /* DC_128_PRED at the depth */
    if (mode == 2 || mode == 9 || mode == 10) {
        int sl = 0, st = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            sl += e[i];
            st += e[N + 1 + i];
        }
        dc = mode == 2 ? (sl + st + N) >> (L8 ? 4 : 3) : ((mode == 9 ? sl : st) + N / 2) >> (L8 ? 3 : 2);
    }
    const int y = it / QW, x0 = 4 * (it % QW);
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        v[j] = hp_dir_sample<N>(mode, e, x0 + j, y, dc);
    hp_store4(src + (ptrdiff_t)y * stride + x0, v);
}

/* pred8x8 (chroma, N = 8: DC per 4x4 quadrant, the "mad cow" edge variants) and pred16x16 (N = 16) */
template <int N, typename P>
__global__ __launch_bounds__(256) void k_h264_pred_blk(uint8_t *plane, ptrdiff_t stride_b, const FFHipH264Pred *blocks, int n, int bd)
{
    const ptrdiff_t stride = stride_b / (ptrdiff_t)sizeof(P);
    const int mid = 1 << (bd - 1), maxv = (1 << bd) - 1;
    constexpr int ITEMS = N * N / 4, RPB = 256 / ITEMS, QW = N / 4, H2 = N / 2;
    __shared__ int L[RPB][N], T[RPB][N + 1]; /* T[0] is the corner */
    __shared__ int PP[RPB][4];
    const int r = threadIdx.x / ITEMS, it = threadIdx.x % ITEMS, b = blockIdx.x * RPB + r;
    const bool valid = b < n;
    FFHipH264Pred k = {};
    if (valid)
        k = blocks[b];
    const int mode = k.mode;
    /* modes: 0 DC 1 HOR 2 VERT 3 PLANE 4 LEFT_DC 5 TOP_DC 6 DC_128; pred8x8 also 7 L0T 8 0LT 9 L00 10 0L0 (h264pred.h:67-82) */
    const bool use_t = valid && (mode == 0 || mode == 2 || mode == 3 || mode == 5 || mode == 7 || mode == 8);
    const bool use_l = valid && (mode == 0 || mode == 1 || mode == 3 || mode == 4 || mode >= 7);
    const int lrows = mode == 7 ? 4 : N; /* L0T: pred4x4_dc on the first quadrant reads four rows of the left column */
    P *src = reinterpret_cast<P *>(plane + k.offset);
    if (it < N) {
        L[r][it] = use_l && it < lrows ? src[(ptrdiff_t)it * stride - 1] : 0;
        T[r][it + 1] = use_t ? src[it - stride] : 0;
    } else if (it == N) {
        T[r][0] = valid && mode == 3 ? src[-stride - 1] : 0;
    }
    __syncthreads();
    if (it == 0 && valid) {
        const int *l = L[r], *t = T[r] + 1;
        if (mode == 3) {
            int H = 0, V = 0;
#pragma unroll
            for (int i = 1; i <= H2; i++) {
                H += i * (t[H2 - 1 + i] - t[H2 - 1 - i]);
                V += i * (l[H2 - 1 + i] - (i == H2 ? t[-1] : l[H2 - 1 - i]));
            }
            H = N == 16 ? (5 * H + 32) >> 6 : (17 * H + 16) >> 5;
            V = N == 16 ? (5 * V + 32) >> 6 : (17 * V + 16) >> 5;
            PP[r][0] = 16 * (l[N - 1] + t[N - 1] + 1) - (H2 - 1) * (V + H);
            PP[r][1] = H;
            PP[r][2] = V;
        } else if (N == 16) {
            int sl = 0, st = 0;
#pragma unroll
            for (int i = 0; i < N; i++) {
                sl += l[i];
                st += t[i];
            }
            PP[r][0] = mode == 0 ? (sl + st + 16) >> 5 : mode == 4 ? (sl + 8) >> 4 : mode == 5 ? (st + 8) >> 4 : mid;
        } else {
            const int t0 = t[0] + t[1] + t[2] + t[3], t1 = t[4] + t[5] + t[6] + t[7];
            const int l0 = l[0] + l[1] + l[2] + l[3], l1 = l[4] + l[5] + l[6] + l[7];
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
            PP[r][0] = q0; PP[r][1] = q1; PP[r][2] = q2; PP[r][3] = q3;
        }
    }
    __syncthreads();
    if (!valid)
        return;
    const int y = it / QW, x0 = 4 * (it % QW);
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        if (mode == 1)
            v[j] = L[r][y];
        else if (mode == 2)
            v[j] = T[r][x + 1];
        else if (mode == 3)
            v[j] = min(max((PP[r][0] + y * PP[r][2] + x * PP[r][1]) >> 5, 0), maxv);
        else
            v[j] = N == 16 ? PP[r][0] : PP[r][2 * (y >> 2) + (x >> 2)];
    }
    hp_store4(src + (ptrdiff_t)y * stride + x0, v);
}

/*
 * pred8x8[] at chroma_format_idc 2 (4:2:2): the 8 wide x 16 tall forms (h264pred_template.c:477-530 vertical / horizontal / 128,
 * :567-571 left_dc = two 8x8 left_dc, :599-620 top_dc, :650-698 dc = eight 4x4 DCs, :700-745 the "mad cow" edge variants, :781-817
 * plane with its own V weighting; installed by ff_h264_pred_init() h264pred.c:478-512).  32 items (16 rows x two groups of 4) per
 * block, 8 blocks per workgroup.  Same modes as pred8x8: 0 DC 1 HOR 2 VERT 3 PLANE 4 LEFT_DC 5 TOP_DC 6 DC_128 7 L0T 8 0LT 9 L00
 * 10 0L0.
 */
template <typename P>
__global__ __launch_bounds__(256) void k_h264_pred_8x16(uint8_t *plane, ptrdiff_t stride_b, const FFHipH264Pred *blocks, int n, int bd)
{
    const ptrdiff_t stride = stride_b / (ptrdiff_t)sizeof(P);
    const int mid = 1 << (bd - 1), maxv = (1 << bd) - 1;
    __shared__ int L[8][16], T[8][9]; /* T[0] is the corner */
    __shared__ int PP[8][8];
    const int r = threadIdx.x >> 5, it = threadIdx.x & 31, b = blockIdx.x * 8 + r;
    const bool valid = b < n;
    FFHipH264Pred k = {};
    if (valid)
        k = blocks[b];
    const int mode = k.mode;
    const bool use_t = valid && (mode == 0 || mode == 2 || mode == 3 || mode == 5 || mode == 7 || mode == 8);
    const bool use_l = valid && (mode == 0 || mode == 1 || mode == 3 || mode == 4 || mode >= 7);
    const int lrows = mode == 7 ? 4 : 16; /* L0T: pred4x4_dc on the first 4x4 reads four rows of the left column */
    P *src = reinterpret_cast<P *>(plane + k.offset);
    if (it < 16)
        L[r][it] = use_l && it < lrows ? src[(ptrdiff_t)it * stride - 1] : 0;
    else if (it < 24)
        T[r][it - 15] = use_t ? src[(it - 16) - stride] : 0;
    else if (it == 24)
        T[r][0] = valid && mode == 3 ? src[-stride - 1] : 0;
    __syncthreads();
    if (it == 0 && valid) {
        const int *l = L[r], *t = T[r] + 1;
        if (mode == 3) {
            /* H over the top row as for 8x8, V over the 16 left rows: k (l[7 + k] - l[7 - k]), k = 1..8, l[-1] = the corner */
            int H = 0, V = 0;
#pragma unroll
            for (int i = 1; i <= 4; i++)
                H += i * (t[3 + i] - t[3 - i]);
#pragma unroll
            for (int i = 1; i <= 8; i++)
                V += i * (l[7 + i] - (i == 8 ? t[-1] : l[7 - i]));
            H = (17 * H + 16) >> 5;
            V = (5 * V + 32) >> 6;
            PP[r][0] = 16 * (l[15] + t[7] + 1) - 7 * V - 3 * H;
            PP[r][1] = H;
            PP[r][2] = V;
        } else {
            const int t0 = t[0] + t[1] + t[2] + t[3], t1 = t[4] + t[5] + t[6] + t[7];
            int lq[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                lq[g] = l[4 * g] + l[4 * g + 1] + l[4 * g + 2] + l[4 * g + 3];
            int q[8]; /* [2 * row group + column half] */
#pragma unroll
            for (int i = 0; i < 8; i++)
                q[i] = mid;
            auto left_dc = [&]() { /* pred8x16_left_dc: each 4-row group's own left sum, both halves */
#pragma unroll
                for (int g = 0; g < 4; g++)
                    q[2 * g] = q[2 * g + 1] = (lq[g] + 2) >> 2;
            };
            auto top_dc = [&]() {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    q[2 * g] = (t0 + 2) >> 2;
                    q[2 * g + 1] = (t1 + 2) >> 2;
                }
            };
            auto full_dc = [&]() { /* pred8x16_dc */
                q[0] = (lq[0] + t0 + 4) >> 3;
                q[1] = (t1 + 2) >> 2;
#pragma unroll
                for (int g = 1; g < 4; g++) {
                    q[2 * g] = (lq[g] + 2) >> 2;
                    q[2 * g + 1] = (t1 + lq[g] + 4) >> 3;
                }
            };
            switch (mode) {
            case 0: full_dc(); break;
            case 4: left_dc(); break;
            case 5: top_dc(); break;
            case 7: top_dc(); q[0] = (t0 + lq[0] + 4) >> 3; break;   /* top_dc, then pred4x4_dc on the first 4x4 */
            case 8: full_dc(); q[0] = (t0 + 2) >> 2; break;          /* dc, then pred4x4_top_dc on the first 4x4 */
            case 9: left_dc(); q[2] = q[3] = mid; break;             /* left_dc, then 128 on rows 4..7 */
            case 10: left_dc(); q[0] = q[1] = mid; break;            /* left_dc, then 128 on rows 0..3 */
            default: break;
            }
#pragma unroll
            for (int i = 0; i < 8; i++)
                PP[r][i] = q[i];
        }
    }
    __syncthreads();
    if (!valid)
        return;
    const int y = it >> 1, x0 = 4 * (it & 1);
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        if (mode == 1)
            v[j] = L[r][y];
        else if (mode == 2)
            v[j] = T[r][x + 1];
        else if (mode == 3)
            v[j] = min(max((PP[r][0] + y * PP[r][2] + x * PP[r][1]) >> 5, 0), maxv);
        else
            v[j] = PP[r][2 * (y >> 2) + (x >> 2)];
    }
    hp_store4(src + (ptrdiff_t)y * stride + x0, v);
}

/* pred4x4_add / pred8x8l_add / pred8x8l_filter_add: thread i owns column i (mode 0, VERT_PRED) or row i (mode 1, HOR_PRED) */
template <int N, bool FILTER, typename P>
__global__ __launch_bounds__(256) void k_h264_pred_add(uint8_t *plane, ptrdiff_t stride_b, int16_t *coeffs, const FFHipH264Pred *blocks, int n)
{
    typedef typename std::conditional<sizeof(P) == 2, int32_t, int16_t>::type CF; /* dctcoef: int32_t above 8 bits */
    const ptrdiff_t stride = stride_b / (ptrdiff_t)sizeof(P);
    const int gid = blockIdx.x * 256 + threadIdx.x, b = gid / N, i = gid % N;
    if (b >= n)
        return;
    const FFHipH264Pred k = blocks[b];
    P *pix = reinterpret_cast<P *>(plane + k.offset);
    CF *blk = reinterpret_cast<CF *>(coeffs) + k.aux; /* aux counts coefficients */
    const bool vert = k.mode == 0;
    const ptrdiff_t along = vert ? stride : 1, across = vert ? 1 : stride; /* steps along / across the prediction direction */
    const P *edge = pix - along;                                              /* the border sample in front of line 0 */
    unsigned v;
    if (FILTER) {
        const bool tl = k.flags & FFHIP_H264_PRED_TOPLEFT, tr = k.flags & FFHIP_H264_PRED_TOPRIGHT;
        const int c = edge[i * across];
        if (i == 0)
            v = hp_a3(tl ? edge[-across] : c, c, edge[across]);
        else if (i == 7)
            v = vert ? hp_a3(tr ? edge[8 * across] : c, c, edge[6 * across]) : (edge[6 * across] + 3 * c + 2) >> 2;
        else
            v = hp_a3(edge[(i - 1) * across], c, edge[(i + 1) * across]);
    } else {
        v = edge[i * across];
    }
    const int cal = vert ? N : 1, cac = vert ? 1 : N; /* the same two steps in the coefficient block */
#pragma unroll
    for (int j = 0; j < N; j++) {
        v = (v + (unsigned)blk[j * cal + i * cac]) & (sizeof(P) == 2 ? 0xFFFFu : 255u); /* the sample type's wrap-around */
        pix[j * along + i * across] = (P)v;
        blk[j * cal + i * cac] = 0;
    }
}


/*
 * The forms ff_h264_pred_init() installs for the OTHER codecs that share H264PredContext (libavcodec/h264pred.c:540-578, bodies
 * :57-432; 8 bits): SVQ3, RV40, VP7 / VP8.  `mode` is a FFHIP_H264_PREDV_* code (include/ffhip.h) that also says the block size.
 * One thread per sample; a thread reads the handful of neighbours its sample's rule names straight from the plane (tails of the
 * table: no staging).  t(i) = the row above (i >= 4 of a 4x4 block: topright[i - 4]), l(i) = the left column (i >= 4: the rows
 * below the block, RV40's "down-left" edge; the _NODOWN forms repeat l3), lt = the corner.
 */
__global__ __launch_bounds__(256) void k_h264_pred_codec(uint8_t *plane, ptrdiff_t stride, const FFHipH264Pred *blocks, int n)
{
    const int b = blockIdx.x;
    if (b >= n)
        return;
    const FFHipH264Pred k = blocks[b];
    const int mode = k.mode;
    const int N = mode < 16 ? 4 : mode < 32 ? 8 : 16;
    const int tid = threadIdx.x;
    if (tid >= N * N)
        return;
    const int x = tid % N, y = tid / N;
    uint8_t *src = plane + k.offset;
    const uint8_t *tr = plane + k.aux;
    auto t = [&](int i) -> int { return N == 4 && i >= 4 ? tr[i - 4] : src[i - stride]; };
    auto lraw = [&](int i) -> int { return src[(ptrdiff_t)i * stride - 1]; };
    const bool nodown = mode == FFHIP_H264_PREDV_DL_RV40_NODOWN || mode == FFHIP_H264_PREDV_VL_RV40_NODOWN || mode == FFHIP_H264_PREDV_HU_RV40_NODOWN;
    auto l = [&](int i) -> int { return lraw(nodown && i > 3 ? 3 : i); };
    auto clip = [](int v) -> int { return v < 0 ? 0 : v > 255 ? 255 : v; };
    int v = 0;
    switch (mode) {
    case FFHIP_H264_PREDV_127_DC: case FFHIP_H264_PREDV8_127_DC: case FFHIP_H264_PREDV16_127_DC: v = 127; break;
    case FFHIP_H264_PREDV_129_DC: case FFHIP_H264_PREDV8_129_DC: case FFHIP_H264_PREDV16_129_DC: v = 129; break;
    case FFHIP_H264_PREDV_VERT_VP8: /* pred4x4_vertical_vp8_c: the row above, smoothed */
        v = ((x ? t(x - 1) : (int)src[-1 - stride]) + 2 * t(x) + t(x + 1) + 2) >> 2;
        break;
    case FFHIP_H264_PREDV_HOR_VP8:  /* pred4x4_horizontal_vp8_c */
        v = ((y ? lraw(y - 1) : (int)src[-1 - stride]) + 2 * lraw(y) + lraw(y < 3 ? y + 1 : 3) + 2) >> 2;
        break;
    case FFHIP_H264_PREDV_DL_SVQ3: { /* pred4x4_down_left_svq3_c */
        const int i = min(x + y + 1, 3);
        v = (lraw(i) + t(i)) >> 1;
        break;
    }
    case FFHIP_H264_PREDV_DL_RV40: case FFHIP_H264_PREDV_DL_RV40_NODOWN: { /* pred4x4_down_left_rv40{,_nodown}_c */
        const int d = x + y;
        v = d < 6 ? (t(d) + t(d + 2) + 2 * t(d + 1) + 2 + l(d) + l(d + 2) + 2 * l(d + 1) + 2) >> 3 : (t(6) + t(7) + 1 + l(6) + l(7) + 1) >> 2;
        break;
    }
    case FFHIP_H264_PREDV_VL_RV40: case FFHIP_H264_PREDV_VL_RV40_NODOWN: /* pred4x4_vertical_left_rv40 (l4 = l3 without the down-left edge) */
        if (!(y & 1)) {
            const int q = x + (y >> 1);
            v = x == 0 && y == 0 ? (2 * t(0) + 2 * t(1) + l(1) + 2 * l(2) + l(3) + 4) >> 3 : (t(q) + t(q + 1) + 1) >> 1;
        } else {
            const int q = x + (y >> 1);
            v = x == 0 && y == 1 ? (t(0) + 2 * t(1) + t(2) + l(2) + 2 * l(3) + l(4) + 4) >> 3 : (t(q) + 2 * t(q + 1) + t(q + 2) + 2) >> 2;
        }
        break;
    case FFHIP_H264_PREDV_VL_VP8: { /* pred4x4_vertical_left_vp8_c: H.264's but for the last column's lower half */
        const int q = x + (y >> 1);
        if (x == 3 && y >= 2)
            v = (t(y + 2) + 2 * t(y + 3) + t(y + 4) + 2) >> 2;
        else
            v = (y & 1) ? (t(q) + 2 * t(q + 1) + t(q + 2) + 2) >> 2 : (t(q) + t(q + 1) + 1) >> 1;
        break;
    }
    case FFHIP_H264_PREDV_HU_RV40: case FFHIP_H264_PREDV_HU_RV40_NODOWN: { /* pred4x4_horizontal_up_rv40{,_nodown}_c */
        const int z = x + 2 * y;
        switch (z) {
        case 0: v = (t(1) + 2 * t(2) + t(3) + 2 * l(0) + 2 * l(1) + 4) >> 3; break;
        case 1: v = (t(2) + 2 * t(3) + t(4) + l(0) + 2 * l(1) + l(2) + 4) >> 3; break;
        case 2: v = (t(3) + 2 * t(4) + t(5) + 2 * l(1) + 2 * l(2) + 4) >> 3; break;
        case 3: v = (t(4) + 2 * t(5) + t(6) + l(1) + 2 * l(2) + l(3) + 4) >> 3; break;
        case 4: v = (t(5) + 2 * t(6) + t(7) + 2 * l(2) + 2 * l(3) + 4) >> 3; break;
        case 5: v = (t(6) + 3 * t(7) + l(2) + 3 * l(3) + 4) >> 3; break;
        case 6: v = (t(6) + t(7) + l(3) + l(4) + 2) >> 2; break;
        case 7: v = (l(3) + 2 * l(4) + l(5) + 2) >> 2; break;
        case 8: v = (l(4) + l(5) + 1) >> 1; break;
        default: v = (l(4) + 2 * l(5) + l(6) + 2) >> 2; break;
        }
        break;
    }
    case FFHIP_H264_PREDV_TM_VP8: case FFHIP_H264_PREDV8_TM_VP8: case FFHIP_H264_PREDV16_TM_VP8: /* pred{4x4,8x8,16x16}_tm_vp8_c */
        v = clip(lraw(y) + t(x) - (int)src[-1 - stride]);
        break;
    case FFHIP_H264_PREDV8_DC_RV40: case FFHIP_H264_PREDV8_LEFT_DC_RV40: case FFHIP_H264_PREDV8_TOP_DC_RV40: { /* pred8x8_*dc_rv40_c */
        int sl = 0, st = 0;
        for (int i = 0; i < 8; i++) {
            if (mode != FFHIP_H264_PREDV8_TOP_DC_RV40) sl += lraw(i);
            if (mode != FFHIP_H264_PREDV8_LEFT_DC_RV40) st += t(i);
        }
        v = mode == FFHIP_H264_PREDV8_DC_RV40 ? (sl + st + 8) >> 4 : (sl + st + 4) >> 3;
        break;
    }
    default: { /* FFHIP_H264_PREDV16_PLANE_SVQ3 / _RV40: pred16x16_plane_compat_8_c (h264pred_template.c:410-456) */
        int H = 0, V = 0;
        for (int q = 1; q <= 8; q++) {
            H += q * (t(7 + q) - (q == 8 ? (int)src[-1 - stride] : t(7 - q)));
            V += q * (lraw(7 + q) - (q == 8 ? (int)src[-1 - stride] : lraw(7 - q)));
        }
        if (mode == FFHIP_H264_PREDV16_PLANE_SVQ3) {
            const int h2 = (5 * (H / 4)) / 16, v2 = (5 * (V / 4)) / 16;
            H = v2; V = h2; /* "required for 100% accuracy": the two are swapped */
        } else {
            H = (H + (H >> 2)) >> 4;
            V = (V + (V >> 2)) >> 4;
        }
        const int a = 16 * (lraw(15) + t(15) + 1) - 7 * (V + H);
        v = clip((a + y * V + x * H) >> 5);
        break;
    }
    }
    /* every sample is computed before any is stored: the block's own samples are nobody's neighbours, but a 4x4 block's topright
     * pointer may lead anywhere */
    __syncthreads();
    src[(ptrdiff_t)y * stride + x] = (uint8_t)v;
}

int ffhip_launch_h264_pred_bd(int bd, int kind, uint8_t *plane, ptrdiff_t stride, int16_t *coeffs, const FFHipH264Pred *blocks, int n,
                              hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (bd != 8 && bd != 9 && bd != 10 && bd != 12 && bd != 14) {
        ffhip_set_error("ffhip_h264_pred: bit depth %d (8, 9, 10, 12 and 14 are the depths H.264 defines)", bd);
        return FFHIP_EINVAL;
    }
    if (bd > 8 && (stride & 1))
        return FFHIP_EINVAL;
    const dim3 block(256);
#define PRED_GO(K, G, ...)                                                                                                   \
    do {                                                                                                                    \
        if (bd > 8) hipLaunchKernelGGL((K<__VA_ARGS__, uint16_t>), dim3(G), block, 0, stream, plane, stride, blocks, n, bd); \
        else        hipLaunchKernelGGL((K<__VA_ARGS__, uint8_t>), dim3(G), block, 0, stream, plane, stride, blocks, n, bd);  \
    } while (0)
#define PRED_ADD(G, ...)                                                                                                            \
    do {                                                                                                                            \
        if (bd > 8) hipLaunchKernelGGL((k_h264_pred_add<__VA_ARGS__, uint16_t>), dim3(G), block, 0, stream, plane, stride, coeffs, blocks, n); \
        else        hipLaunchKernelGGL((k_h264_pred_add<__VA_ARGS__, uint8_t>), dim3(G), block, 0, stream, plane, stride, coeffs, blocks, n);  \
    } while (0)
    switch (kind) {
    case FFHIP_H264_PRED4x4:   PRED_GO(k_h264_pred_dir, cdiv(n, 64), false); break;
    case FFHIP_H264_PRED8x8L:  PRED_GO(k_h264_pred_dir, cdiv(n, 16), true); break;
    case FFHIP_H264_PRED8x8:   PRED_GO(k_h264_pred_blk, cdiv(n, 16), 8); break;
    case FFHIP_H264_PRED16x16: PRED_GO(k_h264_pred_blk, cdiv(n, 4), 16); break;
    case FFHIP_H264_PRED8x16:
        if (bd > 8) hipLaunchKernelGGL((k_h264_pred_8x16<uint16_t>), dim3(cdiv(n, 8)), block, 0, stream, plane, stride, blocks, n, bd);
        else        hipLaunchKernelGGL((k_h264_pred_8x16<uint8_t>), dim3(cdiv(n, 8)), block, 0, stream, plane, stride, blocks, n, bd);
        break;
    case FFHIP_H264_PRED4x4_ADD:         PRED_ADD(cdiv(n, 64), 4, false); break;
    case FFHIP_H264_PRED8x8L_ADD:        PRED_ADD(cdiv(n, 32), 8, false); break;
    case FFHIP_H264_PRED8x8L_FILTER_ADD: PRED_ADD(cdiv(n, 32), 8, true); break;
    case FFHIP_H264_PRED_CODEC:
        if (bd != 8) {
            ffhip_set_error("ffhip_h264_pred: the other codecs' forms exist at 8 bits only (libavcodec/h264pred.c:540)");
            return FFHIP_EINVAL;
        }
        hipLaunchKernelGGL(k_h264_pred_codec, dim3(n), block, 0, stream, plane, stride, blocks, n);
        break;
    default:
        ffhip_set_error("ffhip_h264_pred: kind %d outside 0..8", kind);
        return FFHIP_EINVAL;
    }
#undef PRED_GO
#undef PRED_ADD
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_h264_pred(int kind, uint8_t *plane, ptrdiff_t stride, int16_t *coeffs, const FFHipH264Pred *blocks, int n, hipStream_t stream)
{
    return ffhip_launch_h264_pred_bd(8, kind, plane, stride, coeffs, blocks, n, stream);
}
