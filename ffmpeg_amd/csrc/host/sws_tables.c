/*
 * sws_tables.c — host-side (cold path) table generation for the hip swscale arch.
 *
 * When libffhip is installed under FFmpeg, the tables come from FFmpeg itself
 * (ffhip_sws_from_tables).  Stand-alone, this file derives the same tables:
 *
 *   - separable filter banks  == initFilter()             libswscale/utils.c:197-612
 *     with the call-site arguments of ff_sws_init_single_context, utils.c:1250-1251,1393-1396,
 *     1428-1429,1675-1730 (cpu_flags == 0 => filterAlign 1; default chroma siting => pos 128)
 *   - yuv2rgb LUT coefficients == ff_yuv2rgb_c_init_tables() libswscale/yuv2rgb.c:717-800
 *   - converter selection      == ff_get_unscaled_swscale()  libswscale/swscale_unscaled.c:2425-2431
 *
 * Integer arithmetic follows the reference step for step because the GPU kernels must consume
 * bit-identical tables; tests/test_sws_tables.py pins every bank against the reference's.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ffhip.h"
#include "ffhip_internal.h"

#define REDUCE_CUTOFF 0.002 /* SWS_MAX_REDUCE_CUTOFF, libswscale/swscale.h:447 */

static int ilog2(unsigned v)
{
    int n = 0;
    while (v >>= 1)
        n++;
    return n;
}

static int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

/* round-to-nearest signed division, ties away from zero (ROUNDED_DIV, libavutil/common.h) */
static int64_t rdiv(int64_t a, int64_t b)
{
    return a > 0 ? (a + (b >> 1)) / b : (a - (b >> 1)) / b;
}

static int scaler_of(int flags, int chroma)
{
    static const int order[] = { FFHIP_SWS_FAST_BILINEAR, FFHIP_SWS_BILINEAR, FFHIP_SWS_BICUBIC, 0x8 /* X */,
                                 FFHIP_SWS_POINT, FFHIP_SWS_AREA, FFHIP_SWS_BICUBLIN, FFHIP_SWS_GAUSS,
                                 FFHIP_SWS_SINC, FFHIP_SWS_LANCZOS, 0x400 /* SPLINE */ };
    for (unsigned i = 0; i < sizeof(order) / sizeof(order[0]); i++)
        if (flags & order[i]) {
            if (order[i] == FFHIP_SWS_BICUBLIN)
                return chroma ? FFHIP_SWS_BILINEAR : FFHIP_SWS_BICUBIC;
            return order[i];
        }
    return FFHIP_SWS_BICUBIC; /* SWS_SCALE_AUTO default, utils.c:1226-1234 */
}

/* raw (un-normalised) weight of one tap at fixed-point distance d (1<<30 == one source sample) */
static int64_t tap_weight(int scaler, int64_t d, int64_t xInc, int64_t fone)
{
    double fd = (double)d * (1.0 / (1 << 30));
    int64_t w;
    switch (scaler) {
    case FFHIP_SWS_BICUBIC: {
        /* B = 0, C = 0.6 in 24-bit fixed point */
        const int64_t B = 0;
        const int64_t C = (int64_t)(0.6 * (1 << 24));
        if (d >= 1LL << 31) {
            w = 0;
        } else {
            int64_t d2 = (d * d) >> 30;
            int64_t d3 = (d2 * d) >> 30;
            if (d < 1LL << 30)
                w = (12 * (1 << 24) - 9 * B - 6 * C) * d3 + (-18 * (1 << 24) + 12 * B + 6 * C) * d2 +
                    (6 * (1 << 24) - 2 * B) * (int64_t)(1 << 30);
            else
                w = (-B - 6 * C) * d3 + (6 * B + 30 * C) * d2 + (-12 * B - 48 * C) * d +
                    (8 * B + 24 * C) * (int64_t)(1 << 30);
        }
        return w / ((1LL << 54) / fone);
    }
    case FFHIP_SWS_BILINEAR:
        w = (1 << 30) - d;
        if (w < 0)
            w = 0;
        return w * (fone >> 30);
    case FFHIP_SWS_AREA: {
        int64_t e = d - (1 << 29);
        if (e * xInc < -(1LL << (29 + 16)))
            w = (int64_t)(1.0 * (1LL << (30 + 16)));
        else if (e * xInc < (1LL << (29 + 16)))
            w = -e * xInc + (1LL << (29 + 16));
        else
            w = 0;
        return w * (fone >> (30 + 16));
    }
    case FFHIP_SWS_GAUSS:
        return (int64_t)(exp2(-3.0 * fd * fd) * fone);
    case FFHIP_SWS_SINC:
        return (int64_t)((d ? sin(fd * M_PI) / (fd * M_PI) : 1.0) * fone);
    case FFHIP_SWS_LANCZOS: {
        const double p = 3.0;
        w = (int64_t)((d ? sin(fd * M_PI) * sin(fd * M_PI / p) / (fd * fd * M_PI * M_PI / p) : 1.0) * fone);
        if (fd > p)
            w = 0;
        return w;
    }
    }
    return 0;
}

static int support_factor(int scaler)
{
    switch (scaler) {
    case FFHIP_SWS_AREA:     return 1;
    case FFHIP_SWS_BILINEAR: return 2;
    case FFHIP_SWS_BICUBIC:  return 4;
    case FFHIP_SWS_GAUSS:    return 8;
    case FFHIP_SWS_SINC:     return 20;
    case FFHIP_SWS_LANCZOS:  return 6;
    }
    return -1;
}

/*
 * One filter bank.  `one` is 1<<14 (horizontal) or 1<<12 (vertical); srcPos == dstPos == 128.
 * Returns the tap count, or <0.  *out_filter has (dstW+3)*size entries, *out_pos dstW+3 —
 * the same over-read padding the reference allocates (utils.c:211,564,590-601).
 */
int ffhip_host_init_filter(int16_t **out_filter, int32_t **out_pos, int xInc, int srcW, int dstW, int one,
                           int scaler, int flags)
{
    const int srcPos = 128, dstPos = 128;
    int ratio_log = srcW / dstW ? ilog2(srcW / dstW) : 0;
    const int64_t fone = 1LL << (54 - (ratio_log < 8 ? ratio_log : 8));
    int64_t *w = NULL, *w2 = NULL;
    int32_t *pos = calloc((size_t)dstW + 3, sizeof(*pos));
    int taps, i, j;

    if (!pos)
        return FFHIP_ENOMEM;

    if (abs(xInc - 0x10000) < 10 && srcPos == dstPos) {
        /* identity */
        taps = 1;
        w = calloc(dstW, sizeof(*w));
        if (!w)
            goto nomem;
        for (i = 0; i < dstW; i++) {
            w[i] = fone;
            pos[i] = i;
        }
    } else if (scaler == FFHIP_SWS_POINT) {
        int64_t x = (((int64_t)dstPos * xInc) >> 8) - ((srcPos * 0x8000LL) >> 7);
        taps = 1;
        w = calloc(dstW, sizeof(*w));
        if (!w)
            goto nomem;
        for (i = 0; i < dstW; i++, x += xInc) {
            pos[i] = (int32_t)((x + (1 << 15)) >> 16);
            w[i] = fone;
        }
    } else if ((xInc <= (1 << 16) && scaler == FFHIP_SWS_AREA) || scaler == FFHIP_SWS_FAST_BILINEAR) {
        int64_t x = (((int64_t)dstPos * xInc) >> 8) - ((srcPos * 0x8000LL) >> 7);
        taps = 2;
        w = calloc((size_t)dstW * 2, sizeof(*w));
        if (!w)
            goto nomem;
        for (i = 0; i < dstW; i++, x += xInc) {
            int xx = (int)((x - (1 << 15) + (1 << 15)) >> 16);
            pos[i] = xx;
            for (j = 0; j < 2; j++, xx++) {
                int64_t c = fone - iabs64((int64_t)xx * (1 << 16) - x) * (fone >> 16);
                w[i * 2 + j] = c < 0 ? 0 : c;
            }
        }
    } else {
        int sf = support_factor(scaler);
        int64_t x;
        if (sf < 0) {
            free(pos);
            return FFHIP_EINVAL;
        }
        taps = xInc <= (1 << 16) ? 1 + sf : 1 + (sf * srcW + dstW - 1) / dstW;
        if (taps > srcW - 2)
            taps = srcW - 2;
        if (taps < 1)
            taps = 1;
        w = calloc((size_t)dstW * taps, sizeof(*w));
        if (!w)
            goto nomem;
        x = (((int64_t)dstPos * xInc) >> 7) - ((srcPos * 0x10000LL) >> 7);
        for (i = 0; i < dstW; i++, x += 2LL * xInc) {
            int xx = (int)((x - (taps - 2) * (1LL << 16)) / (1 << 17));
            pos[i] = xx;
            for (j = 0; j < taps; j++, xx++) {
                int64_t d = iabs64(((int64_t)xx * (1 << 17)) - x) << 13;
                if (xInc > 1 << 16)
                    d = d * dstW / srcW;
                w[i * taps + j] = tap_weight(scaler, d, xInc, fone);
            }
        }
    }

    /* no src/dst blur vectors: filter2 == filter */
    w2 = w;
    w = NULL;
    {
        const int n2 = taps;
        int need = 0;
        const double cut = REDUCE_CUTOFF * (double)fone;
        /* trim near-zero taps: shift rows left while the discarded mass stays under the cut-off and
         * positions stay non-decreasing; count how many taps the widest row still needs */
        for (i = dstW - 1; i >= 0; i--) {
            int64_t *row = w2 + (size_t)i * n2;
            int keep = n2;
            int64_t mass = 0;
            for (j = 0; j < n2; j++) {
                mass += iabs64(row[0]);
                if ((double)mass > cut)
                    break;
                if (i < dstW - 1 && pos[i] >= pos[i + 1])
                    break;
                memmove(row, row + 1, (n2 - 1) * sizeof(*row));
                row[n2 - 1] = 0;
                pos[i]++;
            }
            mass = 0;
            for (j = n2 - 1; j > 0; j--) {
                mass += iabs64(row[j]);
                if ((double)mass > cut)
                    break;
                keep--;
            }
            if (keep > need)
                need = keep;
        }
        /* filterAlign == 1 with cpu_flags == 0 */
        (void)flags;
        if (need <= 0 || need >= 256) { /* MAX_FILTER_SIZE, swscale_internal.h:55 */
            free(w2);
            free(pos);
            return FFHIP_EINVAL; /* the reference would cascade contexts here (RETCODE_USE_CASCADE) */
        }
        w = calloc((size_t)dstW * need, sizeof(*w));
        if (!w)
            goto nomem;
        for (i = 0; i < dstW; i++)
            for (j = 0; j < need; j++)
                w[(size_t)i * need + j] = j < n2 ? w2[(size_t)i * n2 + j] : 0;
        free(w2);
        w2 = NULL;
        taps = need;
    }

    /* fold taps that fall outside [0, srcW) onto the border sample */
    for (i = 0; i < dstW; i++) {
        int64_t *row = w + (size_t)i * taps;
        if (pos[i] < 0) {
            for (j = 1; j < taps; j++) {
                int left = j + pos[i] > 0 ? j + pos[i] : 0;
                row[left] += row[j];
                row[j] = 0;
            }
            pos[i] = 0;
        }
        if (pos[i] + taps > srcW) {
            int shift = pos[i] + (taps - srcW < 0 ? taps - srcW : 0);
            int64_t acc = 0;
            for (j = taps - 1; j >= 0; j--)
                if (pos[i] + j >= srcW) {
                    acc += row[j];
                    row[j] = 0;
                }
            for (j = taps - 1; j >= 0; j--)
                row[j] = j < shift ? 0 : row[j - shift];
            pos[i] -= shift;
            row[srcW - 1 - pos[i]] += acc;
        }
    }

    /* normalise each row to `one` with error feedback between taps */
    {
        int16_t *f = calloc(((size_t)dstW + 3) * taps, sizeof(*f));
        if (!f)
            goto nomem;
        for (i = 0; i < dstW; i++) {
            const int64_t *row = w + (size_t)i * taps;
            int64_t sum = 0, err = 0;
            for (j = 0; j < taps; j++)
                sum += row[j];
            sum = (sum + one / 2) / one;
            if (!sum)
                sum = 1;
            for (j = 0; j < taps; j++) {
                int64_t v = row[j] + err;
                int q = (int)rdiv(v, sum);
                f[(size_t)i * taps + j] = (int16_t)q;
                err = v - q * sum;
            }
        }
        for (i = 0; i < 3; i++) {
            pos[dstW + i] = pos[dstW - 1];
            memcpy(f + (size_t)(dstW + i) * taps, f + (size_t)(dstW - 1) * taps, taps * sizeof(*f));
        }
        *out_filter = f;
    }
    *out_pos = pos;
    free(w);
    return taps;

nomem:
    free(w);
    free(w2);
    free(pos);
    return FFHIP_ENOMEM;
}

/* ITU-R BT.601 row of ff_yuv2rgb_coeffs[] == SWS_CS_DEFAULT (libswscale/yuv2rgb.c:47-59) */
static const int32_t cs_default[4] = { 104597, 132201, 25675, 53279 };

/* roundToInt16() (yuv2rgb.c:705-716) followed by the (int16_t) cast of its callers */
static int round_to_int16(int64_t f)
{
    const int r = (int)((f + (1 << 15)) >> 16);
    return r < -0x7FFF ? (int16_t)0x8000 : r > 0x7FFF ? 0x7FFF : r;
}

/* ff_yuv2rgb_c_init_tables() (libswscale/yuv2rgb.c:750-797) for any matrix row, range and brightness / contrast / saturation: the
 * coefficient arithmetic of that function, line by line, without the LUT fill (the kernels evaluate the ramp in closed form).
 * inv_table = c->srcColorspaceTable (sws_getCoefficients()), the other arguments as sws_setColorspaceDetails() stores them
 * (utils.c:848-905: c->brightness, c->contrast, c->saturation, sws->src_range). */
int ffhip_sws_yuv2rgb_coeffs(FFHipSwsTables *t, const int inv_table[4], int fullRange, int brightness, int contrast, int saturation)
{
    int64_t crv, cbu, cgu, cgv, cy = 1 << 16, oy = 0, d;
    if (!t || !inv_table)
        return FFHIP_EINVAL;
    crv = inv_table[0]; cbu = inv_table[1]; cgu = -(int64_t)inv_table[2]; cgv = -(int64_t)inv_table[3];
    if (!fullRange) {
        cy = (cy * 255) / 219;
        oy = 16 << 16;
    } else {
        crv = (crv * 224) / 255;
        cbu = (cbu * 224) / 255;
        cgu = (cgu * 224) / 255;
        cgv = (cgv * 224) / 255;
    }
    cy  = (cy  * contrast)              >> 16;
    crv = (crv * contrast * saturation) >> 32;
    cbu = (cbu * contrast * saturation) >> 32;
    cgu = (cgu * contrast * saturation) >> 32;
    cgv = (cgv * contrast * saturation) >> 32;
    oy -= 256LL * brightness;
    /* the full-chroma writers' int16 coefficients (yuv2rgb.c:786-791), from the values BEFORE the division by cy */
    t->yuv2rgb_full[0] = round_to_int16(cy * (1 << 13));
    t->yuv2rgb_full[1] = round_to_int16(oy * (1 << 9));
    t->yuv2rgb_full[2] = round_to_int16(crv * (1 << 13));
    t->yuv2rgb_full[3] = round_to_int16(cgv * (1 << 13));
    t->yuv2rgb_full[4] = round_to_int16(cgu * (1 << 13));
    t->yuv2rgb_full[5] = round_to_int16(cbu * (1 << 13));
    d = cy > 1 ? cy : 1; /* FFMAX(cy, 1), yuv2rgb.c:794-797 */
    t->yuv2rgb_cy  = cy;
    t->yuv2rgb_oy  = oy;
    t->yuv2rgb_crv = ((crv * (1 << 16)) + 0x8000) / d;
    t->yuv2rgb_cbu = ((cbu * (1 << 16)) + 0x8000) / d;
    t->yuv2rgb_cgu = ((cgu * (1 << 16)) + 0x8000) / d;
    t->yuv2rgb_cgv = ((cgv * (1 << 16)) + 0x8000) / d;
    t->yuv2rgb_yoffs = (fullRange ? 384 : 326) + 512; /* + YUVRGB_TABLE_LUMA_HEADROOM */
    return 0;
}

/* ... for the default matrix (SWS_CS_DEFAULT) and neutral brightness / contrast / saturation: what sws_getContext() starts from */
void ffhip_host_yuv2rgb_coeffs(FFHipSwsTables *t, int fullRange)
{
    ffhip_sws_yuv2rgb_coeffs(t, cs_default, fullRange, 0, 1 << 16, 1 << 16);
}

struct FFHipSwsHostTables {
    FFHipSwsTables t;
    int16_t *f[4];
    int32_t *p[4];
    int unscaled_yuv2rgb;
};

/* av_pix_fmt_desc_get() of the deeper YUV formats on this path: comp[0].depth, planar / semi-planar, log2_chroma_w / _h */
int ffhip_pixfmt_hbd(int fmt, int *depth, int *layout, int *hsub, int *vsub)
{
    int d, l = 0, hs = 1, vs = 1;
    switch (fmt) {
    case FFHIP_PIX_FMT_YUV420P9LE:  d = 9; break;
    case FFHIP_PIX_FMT_YUV420P10LE: d = 10; break;
    case FFHIP_PIX_FMT_YUV420P12LE: d = 12; break;
    case FFHIP_PIX_FMT_YUV420P14LE: d = 14; break;
    case FFHIP_PIX_FMT_YUV420P16LE: d = 16; break;
    case FFHIP_PIX_FMT_YUV422P9LE:  d = 9; vs = 0; break;
    case FFHIP_PIX_FMT_YUV422P10LE: d = 10; vs = 0; break;
    case FFHIP_PIX_FMT_YUV422P12LE: d = 12; vs = 0; break;
    case FFHIP_PIX_FMT_YUV422P14LE: d = 14; vs = 0; break;
    case FFHIP_PIX_FMT_YUV422P16LE: d = 16; vs = 0; break;
    case FFHIP_PIX_FMT_YUV444P9LE:  d = 9; hs = vs = 0; break;
    case FFHIP_PIX_FMT_YUV444P10LE: d = 10; hs = vs = 0; break;
    case FFHIP_PIX_FMT_YUV444P12LE: d = 12; hs = vs = 0; break;
    case FFHIP_PIX_FMT_YUV444P14LE: d = 14; hs = vs = 0; break;
    case FFHIP_PIX_FMT_YUV444P16LE: d = 16; hs = vs = 0; break;
    case FFHIP_PIX_FMT_P010LE: d = 10; l = 1; break;
    case FFHIP_PIX_FMT_P012LE: d = 12; l = 1; break;
    case FFHIP_PIX_FMT_P016LE: d = 16; l = 1; break;
    default: return 0;
    }
    if (depth) *depth = d;
    if (layout) *layout = l;
    if (hsub) *hsub = hs;
    if (vsub) *vsub = vs;
    return 1;
}

static int is_yuv(int fmt)
{
    return fmt == FFHIP_PIX_FMT_YUV420P || fmt == FFHIP_PIX_FMT_NV12 || fmt == FFHIP_PIX_FMT_NV21 || fmt == FFHIP_PIX_FMT_YUV422P ||
           fmt == FFHIP_PIX_FMT_YUV444P || ffhip_pixfmt_hbd(fmt, NULL, NULL, NULL, NULL);
}
/* av_pix_fmt_get_chroma_sub_sample() of the YUV formats on this path (libswscale/utils.c:1265-1266) */
static int chroma_hsub(int fmt)
{
    int hs;
    if (ffhip_pixfmt_hbd(fmt, NULL, NULL, &hs, NULL))
        return hs;
    return fmt == FFHIP_PIX_FMT_YUV444P ? 0 : 1;
}
static int chroma_vsub(int fmt)
{
    int vs;
    if (ffhip_pixfmt_hbd(fmt, NULL, NULL, NULL, &vs))
        return vs;
    return fmt == FFHIP_PIX_FMT_YUV444P || fmt == FFHIP_PIX_FMT_YUV422P ? 0 : 1;
}
static int is_rgb(int fmt)
{
    return fmt == FFHIP_PIX_FMT_RGB24 || fmt == FFHIP_PIX_FMT_BGR24 || (fmt >= FFHIP_PIX_FMT_ARGB && fmt <= FFHIP_PIX_FMT_BGRA);
}

static int ceil_rshift(int a, int b) { return -((-a) >> b); }

/* solve dst = ((src << src_shift) * coeff + offset) >> (mult_shift + src_shift) for the end points of the two ranges
 * (solve_range_convert / init_range_convert_constants, libswscale/swscale.c:568-624).  dst_depth = c->dstBpc (8 for 8-bit targets). */
static void range_solve(unsigned src_min, unsigned src_max, unsigned dst_min, unsigned dst_max, int src_shift, int mult_shift,
                        uint32_t *coeff, int64_t *offset)
{
    const unsigned sr = (uint16_t)(src_max - src_min), dr = (uint16_t)(dst_max - dst_min);
    const int total = mult_shift + src_shift;
    const uint64_t q = ((uint64_t)dr << total) / sr;
    *coeff = (uint32_t)((q + ((uint64_t)1 << src_shift) - 1) >> src_shift); /* AV_CEIL_RSHIFT */
    *offset = ((int64_t)dst_max << total) - ((int64_t)src_max << src_shift) * *coeff + (1U << (mult_shift - 1));
}
void ffhip_sws_range_constants(int src_range, int dst_depth, uint32_t *lum_coeff, int64_t *lum_offset, uint32_t *chr_coeff, int64_t *chr_offset)
{
    const int bd = dst_depth > 16 ? 16 : dst_depth, src_bits = bd <= 14 ? 15 : 19, src_shift = src_bits - bd, mult_shift = bd <= 14 ? 14 : 18;
    const unsigned mpeg_min = 16U << (bd - 8), mpeg_max_lum = 235U << (bd - 8), mpeg_max_chr = 240U << (bd - 8), jpeg_max = (1U << bd) - 1;
    if (src_range) {
        range_solve(0, jpeg_max, mpeg_min, mpeg_max_lum, src_shift, mult_shift, lum_coeff, lum_offset);
        range_solve(0, jpeg_max, mpeg_min, mpeg_max_chr, src_shift, mult_shift, chr_coeff, chr_offset);
    } else {
        range_solve(mpeg_min, mpeg_max_lum, 0, jpeg_max, src_shift, mult_shift, lum_coeff, lum_offset);
        range_solve(mpeg_min, mpeg_max_chr, 0, jpeg_max, src_shift, mult_shift, chr_coeff, chr_offset);
    }
}

/*
 * A packed 8-bit RGB source in front of a YUV target (kernels/sws_rgbin.hip): the 14-bit planar format whose context serves it, or 0 when
 * the conversion is off the hip path.  half: chrSrcHSubSample = 1 (utils.c:1340-1352: "drop every other pixel for chroma calculation
 * unless user wants full chroma").  table: input_rgb2yuv_table for SWS_CS_DEFAULT and a limited-range target — fill_rgb2yuv_table()'s
 * closing branch, the same double expressions (utils.c:693-703).  ofs: the R, G, B bytes of a pixel.
 */
int ffhip_sws_rgb_source_plan(int srcW, int srcH, int srcFormat, int dstW, int dstH, int dstFormat, int flags, int *half, int32_t table[9],
                              int *bpp, int ofs[3])
{
    const int S = 15; /* RGB2YUV_SHIFT */
    int has_alpha = 0;
    if (!is_rgb(srcFormat) || is_rgb(dstFormat) || dstFormat == FFHIP_PIX_FMT_GBRP)
        return 0;
    if (dstFormat >= FFHIP_PIX_FMT_YUVJ420P && dstFormat <= FFHIP_PIX_FMT_YUVJ444P)
        return 0; /* a full-range target: the range stage behind the converters (swscale.c:568-660) is not wired to this entry */
    switch (srcFormat) {
    case FFHIP_PIX_FMT_RGB24: *bpp = 3; ofs[0] = 0; ofs[1] = 1; ofs[2] = 2; break;
    case FFHIP_PIX_FMT_BGR24: *bpp = 3; ofs[0] = 2; ofs[1] = 1; ofs[2] = 0; break;
    case FFHIP_PIX_FMT_RGBA:  *bpp = 4; ofs[0] = 0; ofs[1] = 1; ofs[2] = 2; has_alpha = 1; break;
    case FFHIP_PIX_FMT_BGRA:  *bpp = 4; ofs[0] = 2; ofs[1] = 1; ofs[2] = 0; has_alpha = 1; break;
    case FFHIP_PIX_FMT_ARGB:  *bpp = 4; ofs[0] = 1; ofs[1] = 2; ofs[2] = 3; has_alpha = 1; break;
    case FFHIP_PIX_FMT_ABGR:  *bpp = 4; ofs[0] = 3; ofs[1] = 2; ofs[2] = 1; has_alpha = 1; break;
    default: return 0;
    }
    /* alpha on both sides would run alpToYV12 into the target's alpha plane (needAlpha, utils.c:1398): not built */
    if (has_alpha && (dstFormat == FFHIP_PIX_FMT_YUVA420P || dstFormat == FFHIP_PIX_FMT_YUVA422P || dstFormat == FFHIP_PIX_FMT_YUVA444P))
        return 0;
    /* the one unscaled special converter with an RGB source and a YUV target: bgr24ToYv12Wrapper -> ff_rgb24toyv12 (its own 2 x 2
     * chroma arithmetic, swscale_unscaled.c:2483-2491) */
    if (srcFormat == FFHIP_PIX_FMT_BGR24 && (dstFormat == FFHIP_PIX_FMT_YUV420P || dstFormat == FFHIP_PIX_FMT_YUVA420P) && srcW == dstW &&
        srcH == dstH && !(flags & FFHIP_SWS_ACCURATE_RND) && !(dstW & 1))
        return 0;
    {
        const int df = dstFormat == FFHIP_PIX_FMT_YUVA420P ? FFHIP_PIX_FMT_YUV420P : dstFormat == FFHIP_PIX_FMT_YUVA422P ? FFHIP_PIX_FMT_YUV422P :
                       dstFormat == FFHIP_PIX_FMT_YUVA444P ? FFHIP_PIX_FMT_YUV444P : dstFormat;
        *half = !(srcW & 1) && !(flags & FFHIP_SWS_FULL_CHR_H_INP) && (dstW >> chroma_hsub(df)) <= (srcW >> 1);
    }
    table[0] =  ((int)(0.299 * 219 / 255 * (1 << S) + 0.5));   /* RY */
    table[1] =  ((int)(0.587 * 219 / 255 * (1 << S) + 0.5));   /* GY */
    table[2] =  ((int)(0.114 * 219 / 255 * (1 << S) + 0.5));   /* BY */
    table[3] = (-(int)(0.169 * 224 / 255 * (1 << S) + 0.5));   /* RU */
    table[4] = (-(int)(0.331 * 224 / 255 * (1 << S) + 0.5));   /* GU */
    table[5] =  ((int)(0.500 * 224 / 255 * (1 << S) + 0.5));   /* BU */
    table[6] =  ((int)(0.500 * 224 / 255 * (1 << S) + 0.5));   /* RV */
    table[7] = (-(int)(0.419 * 224 / 255 * (1 << S) + 0.5));   /* GV */
    table[8] = (-(int)(0.081 * 224 / 255 * (1 << S) + 0.5));   /* BV */
    return *half ? FFHIP_PIX_FMT_YUV422P14LE : FFHIP_PIX_FMT_YUV444P14LE;
}

FFHipSwsHostTables *ffhip_sws_tables_create(int srcW, int srcH, int srcFormat, int dstW, int dstH,
                                            int dstFormat, int flags)
{
    FFHipSwsHostTables *h;
    int chrSrcW, chrSrcH, chrDstW, chrDstH, chrDstHSub, chrDstVSub, full_chr;
    int64_t lumXInc, lumYInc, chrXInc, chrYInc;
    int lum_scaler = scaler_of(flags, 0), chr_scaler = scaler_of(flags, 1);
    int r, src_range = 0, dst_range = 0, alpha_fill = 0, rgb_alpha = 0, rgbt;

    /* full-range twins on both sides: no range conversion, the base formats' scaler (handle_jpeg(), utils.c:1019-1050) */
    {
        const int sj = srcFormat >= FFHIP_PIX_FMT_YUVJ420P && srcFormat <= FFHIP_PIX_FMT_YUVJ444P;
        const int dj = dstFormat >= FFHIP_PIX_FMT_YUVJ420P && dstFormat <= FFHIP_PIX_FMT_YUVJ444P;
        static const int base[3] = { FFHIP_PIX_FMT_YUV420P, FFHIP_PIX_FMT_YUV422P, FFHIP_PIX_FMT_YUV444P };
        /* a J format is its base format with the range flag set (handle_jpeg(), utils.c:773-800, :1903-1904) */
        src_range = sj;
        dst_range = dj;
        if (sj)
            srcFormat = base[srcFormat - FFHIP_PIX_FMT_YUVJ420P];
        if (dj)
            dstFormat = base[dstFormat - FFHIP_PIX_FMT_YUVJ420P];
        /* packed RGB targets: the source's range goes into the yuv2rgb tables (ff_yuv2rgb_c_init_tables' fullRange branch) */
    }
    /* an alpha plane on one side only: ignored as a source (needAlpha = isALPHA(src) && isALPHA(dst), utils.c:1398), filled with 255 as
     * a target (ff_swscale's fillPlane, swscale.c:536-553) */
    {
        static const int base[3] = { FFHIP_PIX_FMT_YUV420P, FFHIP_PIX_FMT_YUV422P, FFHIP_PIX_FMT_YUV444P };
        const int sa = srcFormat == FFHIP_PIX_FMT_YUVA420P ? 1 : srcFormat == FFHIP_PIX_FMT_YUVA422P ? 2 : srcFormat == FFHIP_PIX_FMT_YUVA444P ? 3 : 0;
        const int da = dstFormat == FFHIP_PIX_FMT_YUVA420P ? 1 : dstFormat == FFHIP_PIX_FMT_YUVA422P ? 2 : dstFormat == FFHIP_PIX_FMT_YUVA444P ? 3 : 0;
        if (sa && (dstFormat == FFHIP_PIX_FMT_ARGB || dstFormat == FFHIP_PIX_FMT_RGBA || dstFormat == FFHIP_PIX_FMT_ABGR || dstFormat == FFHIP_PIX_FMT_BGRA)) {
            /* the source's alpha plane drives the alpha byte (needAlpha, utils.c:1398): through the table converter's yuva2rgba_c /
             * yuva2argb_c at equal sizes (yuva420p only: yuv2rgb.c:524-529, 640-648), through the scaler's yuv2rgba32_{1,2,X} / _full
             * writers otherwise (output.c:1789-1939, 2160-2310; round 5) */
            const int eq = srcW == dstW && srcH == dstH && !(flags & FFHIP_SWS_ACCURATE_RND) && !(dstH & 1);
            if (eq && (sa != 1 || (dstW & 1))) {
                /* equal sizes without ACCURATE_RND are the reference's special converters: yuva422p / yuva444p and odd widths have
                 * none on the hip path */
                ffhip_set_error("ffhip_sws: equal-size yuva422p / yuva444p (or an odd width) to 32-bit RGB is not on the hip path");
                return NULL;
            }
            rgb_alpha = 1;
        }
        if (sa)
            srcFormat = base[sa - 1];
        if (da)
            dstFormat = base[da - 1];
        /* alpha on both sides of a planar conversion: the alpha plane goes through the LUMA scaler — lum_h_scale and lum_planar_vscale run
         * hyScale / yuv2planeX on plane 3 with the luma banks and the luma dither (hscale.c:63-79, vscale.c:57-70), without the range
         * stage (hscale.c:57-59 converts plane 0 only) */
        alpha_fill = (sa && da) || rgb_alpha ? 2 : da != 0;
    }
    if (flags & FFHIP_SWS_FAST_BILINEAR) {
        /* the C path of SWS_FAST_BILINEAR runs ff_hyscale_fast_c, a different horizontal scaler
         * (libswscale/hscale_fast_bilinear.c) that is not part of this hot path */
        ffhip_set_error("ffhip_sws: SWS_FAST_BILINEAR is not on the hip path");
        return NULL;
    }
    /* the planar RGB target of the equal-size table converter (yuv420p_gbrp_c / yuv422p_gbrp_c, yuv2rgb.c:533, 553) */
    if (dstFormat == FFHIP_PIX_FMT_GBRP && !(srcW == dstW && srcH == dstH && (srcFormat == FFHIP_PIX_FMT_YUV420P || srcFormat == FFHIP_PIX_FMT_YUV422P) &&
                                             !(flags & FFHIP_SWS_ACCURATE_RND) && !(dstH & 1) && !(dstW & 1))) {
        ffhip_set_error("ffhip_sws: gbrp is a target of the equal-size yuv420p / yuv422p converter only");
        return NULL;
    }
    rgbt = is_rgb(dstFormat) || dstFormat == FFHIP_PIX_FMT_GBRP;
    if (!is_yuv(srcFormat) || (!is_yuv(dstFormat) && !rgbt) || srcW < 2 || srcH < 2 || dstW < 2 ||
        dstH < 2) {
        ffhip_set_error("ffhip_sws: unsupported conversion %d -> %d (%dx%d -> %dx%d)", srcFormat, dstFormat,
                        srcW, srcH, dstW, dstH);
        return NULL;
    }
    if (ffhip_pixfmt_hbd(srcFormat, NULL, NULL, NULL, NULL) || ffhip_pixfmt_hbd(dstFormat, NULL, NULL, NULL, NULL)) {
        /* (round 6: a source above 8 bits into packed 8-bit RGB runs in two stages — the 16-bit walker into a 4:2:2-shaped intermediate with
         * an unclipped luma, then the RGB writer of sws_y16rgb.hip; ffhip_sws_from_tables() decides whether the banks fit) */
        if (dstFormat == FFHIP_PIX_FMT_GBRP) {
            ffhip_set_error("ffhip_sws: sources above 8 bits to planar RGB are not on the hip path");
            return NULL;
        }
        /* (equal sizes without a range change take the reference's special converters — planarCopyWrapper, planarToP01xWrapper, ...:
         * swscale_unscaled.c — not the scaler's arithmetic: ffhip_sws_from_tables() refuses those, once the ranges are known) */
    }
    h = calloc(1, sizeof(*h));
    if (!h)
        return NULL;
    h->t.srcW = srcW; h->t.srcH = srcH; h->t.srcFormat = srcFormat;
    h->t.dstW = dstW; h->t.dstH = dstH; h->t.dstFormat = dstFormat;
    h->t.flags = flags;
    h->t.dst_alpha_fill = alpha_fill;
    h->t.src_range = src_range;
    h->t.dst_range = dst_range;
    if (src_range != dst_range && !rgbt) {
        int ddepth = 8;
        ffhip_pixfmt_hbd(dstFormat, &ddepth, NULL, NULL, NULL);
        ffhip_sws_range_constants(src_range, ddepth, &h->t.lumConvertRange_coeff, &h->t.lumConvertRange_offset,
                                  &h->t.chrConvertRange_coeff, &h->t.chrConvertRange_offset);
    }

    /* RGB output without SWS_FULL_CHR_H_INT keeps chroma at half horizontal resolution
     * (utils.c:1359-1360) and full vertical resolution */
    /* SWS_FULL_CHR_H_INT (utils.c:1270-1290): asked for, or forced on a packed RGB target by an odd width or by a source without chroma
     * sub-sampling (unless SWS_FAST_BILINEAR) — chroma then keeps full horizontal resolution and the yuv2rgb_full_* writers run */
    full_chr = rgbt && ((flags & FFHIP_SWS_FULL_CHR_H_INT) || (dstW & 1) ||
                                     (chroma_hsub(srcFormat) == 0 && chroma_vsub(srcFormat) == 0 && !(flags & FFHIP_SWS_FAST_BILINEAR)));
    h->t.full_chr_h_int = full_chr;
    if (full_chr)
        h->t.flags |= FFHIP_SWS_FULL_CHR_H_INT;
    /* RGB output without SWS_FULL_CHR_H_INT keeps chroma at half horizontal resolution (utils.c:1359-1360); full vertical resolution
     * either way.  4:2:2 sources: the chroma banks run from the source's own chroma plane size (chrSrcHSubSample stays the format's for
     * YUV sources: the "drop every other pixel" of utils.c:1368-1392 is for RGB sources) */
    chrDstHSub = rgbt ? (full_chr ? 0 : 1) : chroma_hsub(dstFormat);
    chrDstVSub = rgbt ? 0 : chroma_vsub(dstFormat);
    chrSrcW = ceil_rshift(srcW, chroma_hsub(srcFormat));
    chrSrcH = ceil_rshift(srcH, chroma_vsub(srcFormat));
    chrDstW = ceil_rshift(dstW, chrDstHSub);
    chrDstH = ceil_rshift(dstH, chrDstVSub);

    h->unscaled_yuv2rgb = srcW == dstW && srcH == dstH && (srcFormat == FFHIP_PIX_FMT_YUV420P || srcFormat == FFHIP_PIX_FMT_YUV422P) && rgbt &&
                          !(flags & FFHIP_SWS_ACCURATE_RND) && !(dstH & 1);

    lumXInc = (((int64_t)srcW << 16) + (dstW >> 1)) / dstW;
    lumYInc = (((int64_t)srcH << 16) + (dstH >> 1)) / dstH;
    chrXInc = (((int64_t)chrSrcW << 16) + (chrDstW >> 1)) / chrDstW;
    chrYInc = (((int64_t)chrSrcH << 16) + (chrDstH >> 1)) / chrDstH;

    struct { int xInc, s, d, one, scaler; } bank[4] = {
        { (int)lumXInc, srcW,    dstW,    1 << 14, lum_scaler },
        { (int)chrXInc, chrSrcW, chrDstW, 1 << 14, chr_scaler },
        { (int)lumYInc, srcH,    dstH,    1 << 12, lum_scaler },
        { (int)chrYInc, chrSrcH, chrDstH, 1 << 12, chr_scaler },
    };
    FFHipSwsFilter *out[4] = { &h->t.hLum, &h->t.hChr, &h->t.vLum, &h->t.vChr };
    for (int k = 0; k < 4; k++) {
        r = ffhip_host_init_filter(&h->f[k], &h->p[k], bank[k].xInc, bank[k].s, bank[k].d, bank[k].one,
                                   bank[k].scaler, flags);
        if (r < 0) {
            ffhip_set_error("ffhip_sws: filter bank %d failed (%d)", k, r);
            ffhip_sws_tables_free(h);
            return NULL;
        }
        out[k]->filter = h->f[k];
        out[k]->pos = h->p[k];
        out[k]->size = r;
        out[k]->n = bank[k].d;
    }
    ffhip_host_yuv2rgb_coeffs(&h->t, rgbt ? src_range : 0);
    return h;
}

int ffhip_sws_tables_get(const FFHipSwsHostTables *t, FFHipSwsTables *out)
{
    if (!t || !out)
        return FFHIP_EINVAL;
    *out = t->t;
    return 0;
}

int ffhip_sws_tables_is_unscaled_yuv2rgb(const FFHipSwsHostTables *t) { return t ? t->unscaled_yuv2rgb : 0; }

/* the srcRange / dstRange arguments of sws_setColorspaceDetails() (libswscale/utils.c:848-1000) for a YUV target: 0 limited, 1 full */
int ffhip_sws_tables_set_ranges(FFHipSwsHostTables *t, int src_range, int dst_range)
{
    int ddepth = 8;
    if (!t || (src_range | dst_range) & ~1)
        return FFHIP_EINVAL;
    if (t->t.dst_alpha_fill == 2 && src_range != dst_range) {
        /* the alpha plane runs as the luma of a second pass of the same context: that pass would convert its range, the reference does not */
        ffhip_set_error("ffhip_sws_tables_set_ranges: a range conversion together with a scaled alpha plane is not on the hip path");
        return FFHIP_ENOSYS;
    }
    if (is_rgb(t->t.dstFormat) || t->t.dstFormat == FFHIP_PIX_FMT_GBRP) /* the coefficients carry the source's range (sws_setColorspaceDetails -> ff_yuv2rgb_c_init_tables) */
        ffhip_host_yuv2rgb_coeffs(&t->t, src_range);
    t->t.src_range = src_range;
    t->t.dst_range = dst_range;
    t->t.lumConvertRange_coeff = t->t.chrConvertRange_coeff = 0;
    t->t.lumConvertRange_offset = t->t.chrConvertRange_offset = 0;
    if (src_range != dst_range && !is_rgb(t->t.dstFormat)) {
        ffhip_pixfmt_hbd(t->t.dstFormat, &ddepth, NULL, NULL, NULL);
        ffhip_sws_range_constants(src_range, ddepth, &t->t.lumConvertRange_coeff, &t->t.lumConvertRange_offset,
                                  &t->t.chrConvertRange_coeff, &t->t.chrConvertRange_offset);
    }
    return 0;
}

void ffhip_sws_tables_free(FFHipSwsHostTables *t)
{
    if (!t)
        return;
    for (int k = 0; k < 4; k++) {
        free(t->f[k]);
        free(t->p[k]);
    }
    free(t);
}
