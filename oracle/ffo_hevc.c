/*
 * ffo_hevc.c — CPU restatement of hevcdsp (8, 10 and 12 bits).  TEST INFRASTRUCTURE ONLY.
 *
 * Every function exists as name_bd(bit_depth, ...) — the reference instantiates its templates per BIT_DEPTH (hevc/dsp.c:133-196,
 * bit_depth_template.c): pixels are uint16_t above 8 bits, strides stay in BYTES as in the reference's signatures — and as the
 * 8-bit name(...) the first round's tests call.
 *
 * Follows the BEHAVIOUR of libavcodec/hevc/dsp_template.c:
 *   idct_{4,8,16,32}   :192-284   two 1-D passes (columns, then rows) of the HEVC core transform, each result
 *                                 av_clip_int16((sum + add) >> shift), shift 7 then 20 - bit depth; the partial
 *                                 butterflies sum exact integers, so a pass is the matrix product restricted to the
 *                                 coefficients the `end` limits keep (see hevc_keeps below)
 *   idct_*_dc          :286-300   every residual = (((c0 + 1) >> 1) + add) >> (14 - depth)
 *   transform_4x4_luma :155-188   the 4x4 DST-VII of intra luma
 *   add_residual       :46-59     dst = clip_pixel(dst + res)
 * The 32x32 coefficient matrix is generated from the 31 constants of the H.265 core transform
 * (T[k][i] = sign * g[(2i+1)k mod 128 folded into a quadrant]); tests/test_oracle_vs_ref.py pins all of it against the
 * reference's own table and code compiled in place.
 */
#include <string.h>

#include "ffo.h"

static int clip16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
/* pixel access by bit depth (bit_depth_template.c: pixel = uint8_t / uint16_t, av_clip_pixel); i counts SAMPLES from a byte pointer */
static inline int pget(const uint8_t *p, ptrdiff_t i, int bd) { return bd > 8 ? ((const uint16_t *)p)[i] : p[i]; }
static inline void pput(uint8_t *p, ptrdiff_t i, int v, int bd)
{
    if (bd > 8)
        ((uint16_t *)p)[i] = (uint16_t)v;
    else
        p[i] = (uint8_t)v;
}
static int clipp(int v, int bd) { const int m = (1 << bd) - 1; return v < 0 ? 0 : v > m ? m : v; }
static ptrdiff_t spx(ptrdiff_t stride_bytes, int bd) { return bd > 8 ? stride_bytes / 2 : stride_bytes; }

/* |64 * sqrt(2) * cos(m * pi / 64)| as the standard rounds it, m = 0..31 (g[0] is the DC row's 64) */
static const int g_mag[32] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                               64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4 };

int ffo_hevc_coef(int k, int i) /* T32[k][i]: basis k, sample i of the 32-point transform */
{
    const int m = ((2 * i + 1) * k) & 127; /* angle m * pi / 64 */
    if (k == 0)
        return 64;
    if (m < 32)  return g_mag[m];
    if (m == 32) return 0;
    if (m < 64)  return -g_mag[64 - m];
    if (m < 96)  return -g_mag[m - 64];
    if (m == 96) return 0;
    return g_mag[128 - m];
}

/*
 * Which input coefficients a 1-D pass of size n with limit `end` keeps (TR_32/16/8/4 nest with end, end/2, 8, 4:
 * dsp_template.c:205-251): odd k need k < end; k = 2 mod 4 of the 32-point transform need k/2 < end/2; everything the
 * inner 8- and 4-point stages see is always kept.
 */
static int hevc_keeps(int n, int k, int end)
{
    if (n == 4)
        return 1;
    if (k & 1)
        return k < end;
    if (n == 32 && (k & 3) == 2)
        return (k >> 1) < (end >> 1);
    return 1;
}

static void pass(int16_t *dst, const int16_t *src, int n, int dstep, int sstep, int end, int shift)
{
    const int add = 1 << (shift - 1), scale = 32 / n;
    int out[32];
    for (int i = 0; i < n; i++) {
        int s = 0;
        for (int k = 0; k < n; k++)
            if (hevc_keeps(n, k, end))
                s += ffo_hevc_coef(k * scale, i) * src[k * sstep];
        out[i] = clip16((s + add) >> shift);
    }
    for (int i = 0; i < n; i++)
        dst[i * dstep] = (int16_t)out[i];
}

void ffo_hevc_idct_bd(int bd, int log2_size, int16_t *coeffs, int col_limit)
{
    const int n = 1 << log2_size;
    int limit = col_limit < n ? col_limit : n;
    int limit2 = col_limit + 4 < n ? col_limit + 4 : n;
    for (int i = 0; i < n; i++) { /* columns; the limit shrinks by 4 after columns 4, 8, ... (:271-275) */
        pass(coeffs + i, coeffs + i, n, n, n, limit2, 7);
        if (limit2 < n && i % 4 == 0 && i)
            limit2 -= 4;
    }
    for (int i = 0; i < n; i++)
        pass(coeffs + i * n, coeffs + i * n, n, 1, 1, limit, 20 - bd);
}
void ffo_hevc_idct(int log2_size, int16_t *coeffs, int col_limit) { ffo_hevc_idct_bd(8, log2_size, coeffs, col_limit); }

void ffo_hevc_idct_dc_bd(int bd, int log2_size, int16_t *coeffs)
{
    const int n = 1 << log2_size, shift = 14 - bd, add = 1 << (shift - 1);
    const int v = (((coeffs[0] + 1) >> 1) + add) >> shift;
    for (int i = 0; i < n * n; i++)
        coeffs[i] = (int16_t)v;
}
void ffo_hevc_idct_dc(int log2_size, int16_t *coeffs) { ffo_hevc_idct_dc_bd(8, log2_size, coeffs); }

static void dst4(int16_t *dst, const int16_t *src, int step, int shift)
{
    const int add = 1 << (shift - 1);
    const int s0 = src[0], s1 = src[step], s2 = src[2 * step], s3 = src[3 * step];
    const int c0 = s0 + s2, c1 = s2 + s3, c2 = s0 - s3, c3 = 74 * s1;
    const int o0 = 29 * c0 + 55 * c1 + c3, o1 = 55 * c2 - 29 * c1 + c3, o2 = 74 * (s0 - s2 + s3), o3 = 55 * c0 + 29 * c2 - c3;
    dst[0] = (int16_t)clip16((o0 + add) >> shift);
    dst[step] = (int16_t)clip16((o1 + add) >> shift);
    dst[2 * step] = (int16_t)clip16((o2 + add) >> shift);
    dst[3 * step] = (int16_t)clip16((o3 + add) >> shift);
}

void ffo_hevc_transform_4x4_luma_bd(int bd, int16_t *coeffs)
{
    for (int i = 0; i < 4; i++)
        dst4(coeffs + i, coeffs + i, 4, 7);
    for (int i = 0; i < 4; i++)
        dst4(coeffs + 4 * i, coeffs + 4 * i, 1, 20 - bd);
}
void ffo_hevc_transform_4x4_luma(int16_t *coeffs) { ffo_hevc_transform_4x4_luma_bd(8, coeffs); }

void ffo_hevc_add_residual_bd(int bd, int log2_size, uint8_t *dst, const int16_t *res, ptrdiff_t stride)
{
    const int n = 1 << log2_size;
    stride = spx(stride, bd);
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            pput(dst, y * stride + x, clipp(pget(dst, y * stride + x, bd) + res[y * n + x], bd), bd);
}
void ffo_hevc_add_residual(int log2_size, uint8_t *dst, const int16_t *res, ptrdiff_t stride) { ffo_hevc_add_residual_bd(8, log2_size, dst, res, stride); }

/*
 * HEVC deblocking, 8-bit: hevc_{h,v}_loop_filter_{luma,chroma} (libavcodec/hevc/dsp_template.c:834-929) with the
 * strong / weak / chroma filters of libavcodec/h26x/h2656_deblock_template.c:25-104.  One call covers 8 sample lines
 * along the edge in two groups of 4; a group's decisions read its lines 0 and 3.  vertical != 0: the edge is vertical
 * (hevc_v_*: samples of a line are 1 byte apart, lines are stride apart).
 */
static int iabs(int v) { return v < 0 ? -v : v; }
static int clip3i(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

void ffo_hevc_loop_filter_bd(int bd, int chroma, int vertical, uint8_t *pix_, ptrdiff_t stride, int beta, const int32_t *tc_in,
                             const uint8_t *no_p_in, const uint8_t *no_q_in)
{
    /* beta and tc arrive in 8-bit units: beta <<= BIT_DEPTH - 8, tc = _tc[j] << (BIT_DEPTH - 8) (hevc/dsp_template.c:845,862,907) */
    const ptrdiff_t st = spx(stride, bd), xs = vertical ? 1 : st, ys = vertical ? st : 1;
    ptrdiff_t pix = 0; /* sample offset from pix_ */
    beta <<= bd - 8;
#define PX(line, k) pget(pix_, pix + (line) * ys + (k) * xs, bd) /* k = -4..3: p3 p2 p1 p0 | q0 q1 q2 q3 */
#define PW(line, k, v) pput(pix_, pix + (line) * ys + (k) * xs, (v), bd)
    for (int j = 0; j < 2; j++) {
        const ptrdiff_t save = pix;
        pix += j * 4 * ys;
        const int tc = tc_in[j] << (bd - 8), no_p = no_p_in[j], no_q = no_q_in[j];
        if (chroma) {
            if (tc > 0)
                for (int d = 0; d < 4; d++) {
                    const int p1 = PX(d, -2), p0 = PX(d, -1), q0 = PX(d, 0), q1 = PX(d, 1);
                    const int delta = clip3i((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
                    if (!no_p) PW(d, -1, clipp(p0 + delta, bd));
                    if (!no_q) PW(d, 0, clipp(q0 - delta, bd));
                }
            pix = save;
            continue;
        }
        const int dp0 = iabs(PX(0, -3) - 2 * PX(0, -2) + PX(0, -1)), dq0 = iabs(PX(0, 2) - 2 * PX(0, 1) + PX(0, 0));
        const int dp3 = iabs(PX(3, -3) - 2 * PX(3, -2) + PX(3, -1)), dq3 = iabs(PX(3, 2) - 2 * PX(3, 1) + PX(3, 0));
        const int d0 = dp0 + dq0, d3 = dp3 + dq3;
        if (d0 + d3 < beta) {
            const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
            if (iabs(PX(0, -4) - PX(0, -1)) + iabs(PX(0, 3) - PX(0, 0)) < beta_3 && iabs(PX(0, -1) - PX(0, 0)) < tc25 &&
                iabs(PX(3, -4) - PX(3, -1)) + iabs(PX(3, 3) - PX(3, 0)) < beta_3 && iabs(PX(3, -1) - PX(3, 0)) < tc25 &&
                (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
                const int t = tc << 1; /* the reference passes the same bound three times */
                for (int d = 0; d < 4; d++) {
                    const int p3 = PX(d, -4), p2 = PX(d, -3), p1 = PX(d, -2), p0 = PX(d, -1);
                    const int q0 = PX(d, 0), q1 = PX(d, 1), q2 = PX(d, 2), q3 = PX(d, 3);
                    if (!no_p) {
                        PW(d, -1, (p0 + clip3i(((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0, -t, t)));
                        PW(d, -2, (p1 + clip3i(((p2 + p1 + p0 + q0 + 2) >> 2) - p1, -t, t)));
                        PW(d, -3, (p2 + clip3i(((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2, -t, t)));
                    }
                    if (!no_q) {
                        PW(d, 0, (q0 + clip3i(((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0, -t, t)));
                        PW(d, 1, (q1 + clip3i(((p0 + q0 + q1 + q2 + 2) >> 2) - q1, -t, t)));
                        PW(d, 2, (q2 + clip3i(((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2, -t, t)));
                    }
                }
            } else {
                const int side = (beta + (beta >> 1)) >> 3;
                const int nd_p = dp0 + dp3 < side ? 2 : 1, nd_q = dq0 + dq3 < side ? 2 : 1, tc_2 = tc >> 1;
                for (int d = 0; d < 4; d++) {
                    const int p2 = PX(d, -3), p1 = PX(d, -2), p0 = PX(d, -1), q0 = PX(d, 0), q1 = PX(d, 1), q2 = PX(d, 2);
                    int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
                    if (iabs(delta) < 10 * tc) {
                        delta = clip3i(delta, -tc, tc);
                        if (!no_p) PW(d, -1, clipp(p0 + delta, bd));
                        if (!no_q) PW(d, 0, clipp(q0 - delta, bd));
                        if (!no_p && nd_p > 1)
                            PW(d, -2, clipp(p1 + clip3i((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc_2, tc_2), bd));
                        if (!no_q && nd_q > 1)
                            PW(d, 1, clipp(q1 + clip3i((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc_2, tc_2), bd));
                    }
                }
            }
        }
        pix = save;
    }
#undef PX
#undef PW
}
void ffo_hevc_loop_filter(int chroma, int vertical, uint8_t *pix, ptrdiff_t stride, int beta, const int32_t *tc_in, const uint8_t *no_p_in,
                          const uint8_t *no_q_in)
{
    ffo_hevc_loop_filter_bd(8, chroma, vertical, pix, stride, beta, tc_in, no_p_in, no_q_in);
}

/*
 * HEVC sample adaptive offset, 8-bit: sao_band_filter and sao_edge_filter (libavcodec/h26x/h2656_sao_template.c:24-84).
 * offset_val[0] is unused by the band filter (classes 1..4) and is the "no edge" offset (always 0 from the decoder) of
 * the edge filter.  The reference's edge filter reads a padded copy of the CTB with a FIXED stride of
 * 2*MAX_PB_SIZE + AV_INPUT_BUFFER_PADDING_SIZE = 192 bytes; here the stride is an argument (192 reproduces it).
 */
void ffo_hevc_sao_band_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *offset_val,
                          int left_class, int width, int height)
{
    int table[32] = { 0 };
    const int shift = bd - 5;
    stride_dst = spx(stride_dst, bd);
    stride_src = spx(stride_src, bd);
    for (int k = 0; k < 4; k++)
        table[(k + left_class) & 31] = offset_val[k + 1];
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int s = pget(src, y * stride_src + x, bd);
            pput(dst, y * stride_dst + x, clipp(s + table[(s >> shift) & 31], bd), bd);
        }
}
void ffo_hevc_sao_band(uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *offset_val,
                       int left_class, int width, int height)
{
    ffo_hevc_sao_band_bd(8, dst, src, stride_dst, stride_src, offset_val, left_class, width, height);
}

void ffo_hevc_sao_edge_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *offset_val,
                          int eo, int width, int height)
{
    static const int idx[5] = { 1, 2, 0, 3, 4 };
    static const int dx[4][2] = { { -1, 1 }, { 0, 0 }, { -1, 1 }, { 1, -1 } }, dy[4][2] = { { 0, 0 }, { -1, 1 }, { -1, 1 }, { -1, 1 } };
    stride_dst = spx(stride_dst, bd);
    stride_src = spx(stride_src, bd);
    const ptrdiff_t a = dx[eo][0] + dy[eo][0] * stride_src, b = dx[eo][1] + dy[eo][1] * stride_src;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const ptrdiff_t at = y * stride_src + x;
            const int c = pget(src, at, bd), pa = pget(src, at + a, bd), pb = pget(src, at + b, bd);
            const int d0 = (c > pa) - (c < pa), d1 = (c > pb) - (c < pb);
            pput(dst, y * stride_dst + x, clipp(c + offset_val[idx[2 + d0 + d1]], bd), bd);
        }
}
void ffo_hevc_sao_edge(uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *offset_val, int eo,
                       int width, int height)
{
    ffo_hevc_sao_edge_bd(8, dst, src, stride_dst, stride_src, offset_val, eo, width, height);
}

/*
 * HEVC motion compensation, 8-bit, uni-directional: put_hevc_{qpel,epel}[idx][!!my][!!mx] (14-bit int16 intermediates,
 * row stride MAX_PB_SIZE = 64) and put_hevc_{qpel,epel}_uni (pixels) — libavcodec/h26x/h2656_inter_template.c:29-58,
 * 97-245 (luma), 342-485 (chroma), wired in libavcodec/hevc/dsp.c:133-190.  The [!!my][!!mx] table index picks
 * pixels / h / v / hv; luma filters are the standard's 8-tap quarter-sample set, chroma the 4-tap eighth-sample set.
 * hv: rows -3..height+3 (chroma: -1..height+1) are filtered horizontally first (no shift at 8 bits), then vertically
 * with >> 6.
 */
static const int8_t hevc_luma_filter[4][8] = { { 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 },
                                               { 0, 1, -5, 17, 58, -10, 4, -1 } };
static const int8_t hevc_chroma_filter[8][4] = { { 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 },
                                                 { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

void ffo_hevc_mc_bd(int bd, int chroma, int uni, void *dst_, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, int mx,
                    int my, int width)
{
    /* above 8 bits the one-dimensional sums lose BIT_DEPTH - 8 bits (>> (BIT_DEPTH - 8), h2656_inter_template.c:113,131,160), the
     * unfiltered copy gains 14 - BIT_DEPTH; the uni stage rounds by 14 - BIT_DEPTH (:195-245) */
    const int taps = chroma ? 4 : 8, before = chroma ? 1 : 3, sh1 = bd - 8, shu = 14 - bd;
    const int8_t *hf = chroma ? hevc_chroma_filter[mx] : hevc_luma_filter[mx];
    const int8_t *vf = chroma ? hevc_chroma_filter[my] : hevc_luma_filter[my];
    int16_t *d16 = dst_;
    uint8_t *d8 = dst_;
    srcstride = spx(srcstride, bd);
    dststride = spx(dststride, bd);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            int val;
            if (!mx && !my) {
                val = pget(src, y * srcstride + x, bd) << shu;
                if (uni) { /* put_uni_pixels is a copy */
                    pput(d8, y * dststride + x, pget(src, y * srcstride + x, bd), bd);
                    continue;
                }
            } else if (mx && !my) {
                val = 0;
                for (int t = 0; t < taps; t++)
                    val += hf[t] * pget(src, y * srcstride + x + t - before, bd);
                val >>= sh1;
            } else if (!mx) {
                val = 0;
                for (int t = 0; t < taps; t++)
                    val += vf[t] * pget(src, (y + t - before) * srcstride + x, bd);
                val >>= sh1;
            } else {
                int acc = 0;
                for (int s = 0; s < taps; s++) {
                    int h = 0;
                    for (int t = 0; t < taps; t++)
                        h += hf[t] * pget(src, (y + s - before) * srcstride + x + t - before, bd);
                    acc += vf[s] * (int16_t)(h >> sh1);
                }
                val = acc >> 6;
            }
            if (uni)
                pput(d8, y * dststride + x, clipp((val + (1 << (shu - 1))) >> shu, bd), bd);
            else
                d16[y * 64 + x] = (int16_t)val;
        }
}
void ffo_hevc_mc(int chroma, int uni, void *dst_, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, int mx,
                 int my, int width)
{
    ffo_hevc_mc_bd(8, chroma, uni, dst_, dststride, src, srcstride, height, mx, my, width);
}

/*
 * Weighted and bi-directional prediction: put_hevc_{qpel,epel}_uni_w (h26x/h2656_inter_template.c:60-88,247-340,487-578),
 * put_hevc_{qpel,epel}_bi and _bi_w (hevc/dsp_template.c:368-420,432-625,630-815).  The interpolation is ffo_hevc_mc's 14-bit
 * intermediate; only the output stage differs.  mode: 2 uni_w (wx0 = wx), 3 bi, 4 bi_w; src2 rows are 64 elements apart.
 * Offsets scale with the depth (ox * (1 << (BIT_DEPTH - 8))), shifts are 14 - BIT_DEPTH based.
 */
void ffo_hevc_mc_w_bd(int bd, int chroma, int mode, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                      const int16_t *src2, int height, int denom, int wx0, int wx1, int ox, int mx, int my, int width)
{
    int16_t tmp[64 * 64];
    ffo_hevc_mc_bd(bd, chroma, 0, tmp, 0, src, srcstride, height, mx, my, width);
    dststride = spx(dststride, bd);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int val = tmp[y * 64 + x];
            int out;
            if (mode == 2) {
                const int shift = denom + 14 - bd;
                out = ((val * wx0 + (1 << (shift - 1))) >> shift) + ox * (1 << (bd - 8));
            } else if (mode == 3) {
                const int shift = 14 + 1 - bd;
                out = (val + src2[y * 64 + x] + (1 << (shift - 1))) >> shift;
            } else {
                const int log2wd = denom + 14 - bd;
                out = (val * wx1 + src2[y * 64 + x] * wx0 + (ox * (1 << (bd - 8)) + 1) * (1 << log2wd)) >> (log2wd + 1);
            }
            pput(dst, y * dststride + x, clipp(out, bd), bd);
        }
}
void ffo_hevc_mc_w(int chroma, int mode, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                   int height, int denom, int wx0, int wx1, int ox, int mx, int my, int width)
{
    ffo_hevc_mc_w_bd(8, chroma, mode, dst, dststride, src, srcstride, src2, height, denom, wx0, wx1, ox, mx, my, width);
}

/*
 * The remaining small members of HEVCDSPContext, 8-bit: dequant (transform-skip scaling) and transform_rdpcm
 * (libavcodec/hevc/dsp_template.c:85-143) and sao_edge_restore[2] (libavcodec/h26x/h2656_sao_template.c:81-214).
 */
void ffo_hevc_dequant_bd(int bd, int16_t *coeffs, int log2_size)
{
    const int shift = 15 - bd - log2_size, n = 1 << (2 * log2_size);
    if (bd <= 9 || shift > 0) {
        for (int i = 0; i < n; i++)
            coeffs[i] = (int16_t)((coeffs[i] + (1 << (shift - 1))) >> shift);
    } else if (bd > 10 && shift < 0) {
        for (int i = 0; i < n; i++)
            coeffs[i] = (int16_t)((uint16_t)coeffs[i] << -shift);
    } /* shift == 0: identity */
}
void ffo_hevc_dequant(int16_t *coeffs, int log2_size) { ffo_hevc_dequant_bd(8, coeffs, log2_size); }

void ffo_hevc_transform_rdpcm(int16_t *coeffs, int log2_size, int mode)
{
    const int n = 1 << log2_size;
    if (mode) {
        for (int y = 1; y < n; y++)
            for (int x = 0; x < n; x++)
                coeffs[y * n + x] = (int16_t)(coeffs[y * n + x] + coeffs[(y - 1) * n + x]);
    } else {
        for (int y = 0; y < n; y++)
            for (int x = 1; x < n; x++)
                coeffs[y * n + x] = (int16_t)(coeffs[y * n + x] + coeffs[y * n + x - 1]);
    }
}

/* variant 0 / 1 = sao_edge_restore[0] / [1]; offset0 = sao_offset_val[0]; eo = SAO_EO_*: 0 horizontal, 1 vertical, 2 135 degrees, 3 45 degrees (hevc/hevcdec.h:169-174) */
void ffo_hevc_sao_edge_restore_bd(int bd, int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t sd, ptrdiff_t ss, int eo, int offset0,
                                  const int *borders, int width, int height, const uint8_t *vert_edge, const uint8_t *horiz_edge,
                                  const uint8_t *diag_edge)
{
    sd = spx(sd, bd);
    ss = spx(ss, bd);
    enum { D135 = 2, D45 = 3 };
    int init_x = 0, init_y = 0;
    if (eo != 1) {
        if (borders[0]) {
            for (int y = 0; y < height; y++)
                pput(dst, y * sd, clipp(pget(src, y * ss, bd) + offset0, bd), bd);
            init_x = 1;
        }
        if (borders[2]) {
            for (int y = 0; y < height; y++)
                pput(dst, y * sd + width - 1, clipp(pget(src, y * ss + width - 1, bd) + offset0, bd), bd);
            width--;
        }
    }
    if (eo != 0) {
        if (borders[1]) {
            for (int x = init_x; x < width; x++)
                pput(dst, x, clipp(pget(src, x, bd) + offset0, bd), bd);
            init_y = 1;
        }
        if (borders[3]) {
            for (int x = init_x; x < width; x++)
                pput(dst, x + sd * (height - 1), clipp(pget(src, x + ss * (height - 1), bd) + offset0, bd), bd);
            height--;
        }
    }
    if (!variant)
        return;
    const int save_ul = !diag_edge[0] && eo == D135 && !borders[0] && !borders[1];
    const int save_ur = !diag_edge[1] && eo == D45 && !borders[1] && !borders[2];
    const int save_lr = !diag_edge[2] && eo == D135 && !borders[2] && !borders[3];
    const int save_ll = !diag_edge[3] && eo == D45 && !borders[0] && !borders[3];
    if (vert_edge[0] && eo != 1)
        for (int y = init_y + save_ul; y < height - save_ll; y++)
            pput(dst, y * sd, pget(src, y * ss, bd), bd);
    if (vert_edge[1] && eo != 1)
        for (int y = init_y + save_ur; y < height - save_lr; y++)
            pput(dst, y * sd + width - 1, pget(src, y * ss + width - 1, bd), bd);
    if (horiz_edge[0] && eo != 0)
        for (int x = init_x + save_ul; x < width - save_ur; x++)
            pput(dst, x, pget(src, x, bd), bd);
    if (horiz_edge[1] && eo != 0)
        for (int x = init_x + save_ll; x < width - save_lr; x++)
            pput(dst, (height - 1) * sd + x, pget(src, (height - 1) * ss + x, bd), bd);
    if (diag_edge[0] && eo == D135)
        pput(dst, 0, pget(src, 0, bd), bd);
    if (diag_edge[1] && eo == D45)
        pput(dst, width - 1, pget(src, width - 1, bd), bd);
    if (diag_edge[2] && eo == D135)
        pput(dst, sd * (height - 1) + width - 1, pget(src, ss * (height - 1) + width - 1, bd), bd);
    if (diag_edge[3] && eo == D45)
        pput(dst, sd * (height - 1), pget(src, ss * (height - 1), bd), bd);
}
void ffo_hevc_sao_edge_restore(int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t sd, ptrdiff_t ss, int eo, int offset0,
                               const int *borders, int width, int height, const uint8_t *vert_edge, const uint8_t *horiz_edge,
                               const uint8_t *diag_edge)
{
    ffo_hevc_sao_edge_restore_bd(8, variant, dst, src, sd, ss, eo, offset0, borders, width, height, vert_edge, horiz_edge, diag_edge);
}
