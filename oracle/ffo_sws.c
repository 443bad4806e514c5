/*
 * ffo_sws.c — CPU restatement of the reference's swscale hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so; the
 * product (libffhip.so, ffmpeg_amd/) never does.  Pinned bit-exact against the real reference
 * (oracle/_ref/libffref.so, built from /root/reference by oracle/refbuild/Makefile) and against the
 * fixtures under tests/golden/ that were generated from it (tools/make_golden.py).
 *
 * Each function names the reference code it restates (paths relative to the FFmpeg tree).
 * Filter banks and yuv2rgb coefficients are INPUTS: they come from the reference's initFilter()
 * in the drop-in case, or from ffmpeg_amd/csrc/host/sws_tables.c stand-alone.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ffo.h"

static inline uint8_t clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : (uint8_t)v; }

/* ---------------------------------------------------------------------------------------------
 * yuv2rgb LUTs: ff_yuv2rgb_c_init_tables() case 24 + fill_table()/fill_gv_table()
 * libswscale/yuv2rgb.c:680-700,901-912.  One shared clipped luma ramp of 1024+2*512 entries and
 * four 256+2*512-entry tables of offsets into it, indexed by the chroma sample + 512.
 * ------------------------------------------------------------------------------------------- */
#define HEADROOM      512  /* YUVRGB_TABLE_HEADROOM, swscale_internal.h:52 */
#define LUMA_HEADROOM 512  /* YUVRGB_TABLE_LUMA_HEADROOM, swscale_internal.h:53 */
#define RAMP_SIZE     (1024 + 2 * LUMA_HEADROOM)
#define CTAB_SIZE     (256 + 2 * HEADROOM)

void ffo_yuv2rgb_luts_init(FfoYuv2RgbLuts *l, const FfoYuv2RgbCoeffs *k)
{
    int64_t yb = -(384 << 16) - LUMA_HEADROOM * k->cy - k->oy;
    for (int i = 0; i < RAMP_SIZE; i++, yb += k->cy)
        l->ramp[i] = clip_u8((int)((yb + 0x8000) >> 16));
    for (int i = 0; i < CTAB_SIZE; i++) {
        int c = clip_u8(i - HEADROOM);
        /* fill_table: base = yoffs - (inc >> 9); entry = base + ((c*inc) >> 16) */
        l->rV[i] = k->yoffs - (int)(k->crv >> 9) + (int)((c * k->crv) >> 16);
        l->gU[i] = k->yoffs - (int)(k->cgu >> 9) + (int)((c * k->cgu) >> 16);
        l->bU[i] = k->yoffs - (int)(k->cbu >> 9) + (int)((c * k->cbu) >> 16);
        /* fill_gv_table: plain offset, no ramp base */
        l->gV[i] = -(int)(k->cgv >> 9) + (int)((c * k->cgv) >> 16);
    }
}

/*
 * Packed layouts ("bgr" argument of the functions below): 0 rgb24, 1 bgr24, and the 32-bit ones of yuv2rgb_c_32 /
 * yuv2rgbx32_X (libswscale/yuv2rgb.c:522,943-966; output.c:1697-1714): 2 argb, 3 rgba, 4 abgr, 5 bgra.  Their tables
 * hold the same clipped ramp shifted to the component's byte, with 255 in the alpha byte for a source without alpha.
 */
static inline int px_bytes(int layout) { return layout < 2 ? 3 : 4; }

static inline void put_rgb(uint8_t *d, const FfoYuv2RgbLuts *l, int Y, int U, int V, int bgr)
{
    int r = l->ramp[l->rV[V + HEADROOM] + Y];
    int g = l->ramp[l->gU[U + HEADROOM] + l->gV[V + HEADROOM] + Y];
    int b = l->ramp[l->bU[U + HEADROOM] + Y];
    switch (bgr) {
    case 0: d[0] = r; d[1] = g; d[2] = b; break;
    case 1: d[0] = b; d[1] = g; d[2] = r; break;
    case 2: d[0] = 255; d[1] = r; d[2] = g; d[3] = b; break;
    case 3: d[0] = r; d[1] = g; d[2] = b; d[3] = 255; break;
    case 4: d[0] = 255; d[1] = b; d[2] = g; d[3] = r; break;
    default: d[0] = b; d[1] = g; d[2] = r; d[3] = 255; break;
    }
}

/*
 * yuv2rgb_c_24_rgb / yuv2rgb_c_24_bgr: libswscale/yuv2rgb.c:137-228,530-531.
 * Two luma rows share one chroma row; 8 pixels per iteration, then a 4- and a 2-pixel tail, so an
 * odd trailing column is never written.  Returns srcSliceH like the SwsFunc.
 */
int ffo_yuv420p_to_rgb24(const FfoYuv2RgbLuts *l, int width, const uint8_t *const src[3],
                         const int srcStride[3], int srcSliceY, int srcSliceH, uint8_t *dst, int dstStride,
                         int bgr)
{
    int npairs = (width >> 3) * 4 + ((width & 4) ? 2 : 0) + ((width & 2) ? 1 : 0);
    const int bp = px_bytes(bgr);
    for (int y = 0; y < srcSliceH; y += 2) {
        const uint8_t *py0 = src[0] + (ptrdiff_t)y * srcStride[0];
        const uint8_t *py1 = py0 + srcStride[0];
        const uint8_t *pu = src[1] + (ptrdiff_t)(y >> 1) * srcStride[1];
        const uint8_t *pv = src[2] + (ptrdiff_t)(y >> 1) * srcStride[2];
        uint8_t *d0 = dst + (ptrdiff_t)(y + srcSliceY) * dstStride;
        uint8_t *d1 = d0 + dstStride;
        for (int m = 0; m < npairs; m++) {
            int U = pu[m], V = pv[m];
            put_rgb(d0 + 2 * bp * m,      l, py0[2 * m],     U, V, bgr);
            put_rgb(d0 + 2 * bp * m + bp, l, py0[2 * m + 1], U, V, bgr);
            put_rgb(d1 + 2 * bp * m,      l, py1[2 * m],     U, V, bgr);
            put_rgb(d1 + 2 * bp * m + bp, l, py1[2 * m + 1], U, V, bgr);
        }
    }
    return srcSliceH;
}

/*
 * The table converter's other forms (ff_yuv2rgb_get_func_ptr(), libswscale/yuv2rgb.c:562-676):
 *   c422   YUV422FUNC (yuv2rgb.c:238-320): pu_2 = pu_1 + srcStride[1] — luma row y + 1 takes chroma row y + 1, rows are indexed by y
 *          (YUV2RGBFUNC :154-155: src[1] + (y >> !yuv422) * srcStride[1])
 *   alpha  yuva2rgba_c / yuva2argb_c (:524-529; PUTRGBA :88-93): the 32-bit pixel is r[Y] + g[Y] + b[Y] + (pa[i] << abase) with tables
 *          built WITHOUT the 255 (needAlpha, :943-966) — i.e. the alpha byte is the source's sample; src[3] / srcStride[3]
 *   layout 6: yuv420p_gbrp_c / yuv422p_gbrp_c (:533, 553; PUTGBRP :127-135): dst[0] = G, dst[1] = B, dst[2] = R planes
 * Same width rule as above (8 / 4 / 2-pixel groups: an odd trailing column is never written).
 */
int ffo_yuv2rgb_unscaled(const FfoYuv2RgbLuts *l, int width, const uint8_t *const src[4], const int srcStride[4], int srcSliceY,
                         int srcSliceH, uint8_t *const dst[3], const int dstStride[3], int layout, int c422, int alpha)
{
    const int npairs = (width >> 3) * 4 + ((width & 4) ? 2 : 0) + ((width & 2) ? 1 : 0);
    const int bp = layout == 6 ? 1 : px_bytes(layout);
    for (int y = 0; y < srcSliceH; y++) {
        const int crow = c422 ? y : y >> 1;
        const uint8_t *py = src[0] + (ptrdiff_t)y * srcStride[0];
        const uint8_t *pu = src[1] + (ptrdiff_t)crow * srcStride[1];
        const uint8_t *pv = src[2] + (ptrdiff_t)crow * srcStride[2];
        const uint8_t *pa = alpha ? src[3] + (ptrdiff_t)y * srcStride[3] : NULL;
        uint8_t *d = dst[0] + (ptrdiff_t)(y + srcSliceY) * dstStride[0];
        for (int x = 0; x < 2 * npairs; x++) {
            const int U = pu[x >> 1], V = pv[x >> 1], Y = py[x];
            if (layout == 6) {
                d[x] = l->ramp[l->gU[U + HEADROOM] + l->gV[V + HEADROOM] + Y];
                dst[1][(ptrdiff_t)(y + srcSliceY) * dstStride[1] + x] = l->ramp[l->bU[U + HEADROOM] + Y];
                dst[2][(ptrdiff_t)(y + srcSliceY) * dstStride[2] + x] = l->ramp[l->rV[V + HEADROOM] + Y];
                continue;
            }
            put_rgb(d + bp * x, l, Y, U, V, layout);
            if (pa && layout >= 2)
                d[bp * x + (layout == 2 || layout == 4 ? 0 : 3)] = pa[x];
        }
    }
    return srcSliceH;
}

/* hScale8To15_c: libswscale/swscale.c:128-142 */
void ffo_hscale8to15(int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *pos, int fs)
{
    for (int i = 0; i < dstW; i++) {
        int acc = 0;
        for (int j = 0; j < fs; j++)
            acc += (int)src[pos[i] + j] * filter[fs * i + j];
        acc >>= 7;
        dst[i] = (int16_t)(acc < (1 << 15) - 1 ? acc : (1 << 15) - 1);
    }
}

/* yuv2planeX_8_c: libswscale/output.c:468-483 (unsigned accumulate, arithmetic >> of the int) */
void ffo_yuv2planeX8(const int16_t *filter, int fs, const int16_t *const *src, uint8_t *dest, int dstW,
                     const uint8_t *dither, int offset)
{
    for (int i = 0; i < dstW; i++) {
        uint32_t acc = (uint32_t)dither[(i + offset) & 7] << 12;
        for (int j = 0; j < fs; j++)
            acc += (uint32_t)(src[j][i] * filter[j]);
        dest[i] = clip_u8((int32_t)acc >> 19);
    }
}

/* yuv2plane1_8_c: libswscale/output.c:485-493 */
void ffo_yuv2plane1_8(const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset)
{
    for (int i = 0; i < dstW; i++)
        dest[i] = clip_u8((src[i] + dither[(i + offset) & 7]) >> 7);
}

/* yuv2nv12cX_c: libswscale/output.c:495-529; `swap` = NV21 byte order (isSwappedChroma) */
void ffo_yuv2nv12cX(int swap, const uint8_t *dither, const int16_t *filter, int fs, const int16_t *const *u,
                    const int16_t *const *v, uint8_t *dest, int chrDstW)
{
    for (int i = 0; i < chrDstW; i++) {
        uint32_t au = (uint32_t)dither[i & 7] << 12;
        uint32_t av = (uint32_t)dither[(i + 3) & 7] << 12;
        for (int j = 0; j < fs; j++) {
            au += (uint32_t)(u[j][i] * filter[j]);
            av += (uint32_t)(v[j][i] * filter[j]);
        }
        dest[2 * i + (swap ? 1 : 0)] = clip_u8((int32_t)au >> 19);
        dest[2 * i + (swap ? 0 : 1)] = clip_u8((int32_t)av >> 19);
    }
}

/* yuv2rgb_X_c_template + yuv2rgb_write for RGB24/BGR24: libswscale/output.c:1789-1840,1697-1714 */
static void rgb24_X(const FfoYuv2RgbLuts *l, const int16_t *lf, const int16_t *const *lum, int lfs,
                    const int16_t *cf, const int16_t *const *cu, const int16_t *const *cv, int cfs, uint8_t *dest,
                    int dstW, int bgr)
{
    for (int i = 0; i < dstW >> 1; i++) {
        uint32_t y1 = 1 << 18, y2 = 1 << 18, u = 1 << 18, v = 1 << 18;
        for (int j = 0; j < lfs; j++) {
            y1 += (uint32_t)(lum[j][2 * i] * (int)lf[j]);
            y2 += (uint32_t)(lum[j][2 * i + 1] * (int)lf[j]);
        }
        for (int j = 0; j < cfs; j++) {
            u += (uint32_t)(cu[j][i] * (int)cf[j]);
            v += (uint32_t)(cv[j][i] * (int)cf[j]);
        }
        int U = (int32_t)u >> 19, V = (int32_t)v >> 19;
        put_rgb(dest + 2 * px_bytes(bgr) * i,                 l, (int32_t)y1 >> 19, U, V, bgr);
        put_rgb(dest + 2 * px_bytes(bgr) * i + px_bytes(bgr), l, (int32_t)y2 >> 19, U, V, bgr);
    }
}

/* yuv2rgb_2_c_template: libswscale/output.c:1843-1881 */
static void rgb24_2(const FfoYuv2RgbLuts *l, const int16_t *const lum[2], const int16_t *const cu[2],
                    const int16_t *const cv[2], uint8_t *dest, int dstW, int yalpha, int uvalpha, int bgr)
{
    int ya1 = 4096 - yalpha, uva1 = 4096 - uvalpha;
    for (int i = 0; i < dstW >> 1; i++) {
        int Y1 = (lum[0][2 * i] * ya1 + lum[1][2 * i] * yalpha) >> 19;
        int Y2 = (lum[0][2 * i + 1] * ya1 + lum[1][2 * i + 1] * yalpha) >> 19;
        int U = (cu[0][i] * uva1 + cu[1][i] * uvalpha) >> 19;
        int V = (cv[0][i] * uva1 + cv[1][i] * uvalpha) >> 19;
        put_rgb(dest + 2 * px_bytes(bgr) * i,                 l, Y1, U, V, bgr);
        put_rgb(dest + 2 * px_bytes(bgr) * i + px_bytes(bgr), l, Y2, U, V, bgr);
    }
}

/* yuv2rgb_1_c_template: libswscale/output.c:1883-1939 */
static void rgb24_1(const FfoYuv2RgbLuts *l, const int16_t *lum, const int16_t *const cu[2],
                    const int16_t *const cv[2], uint8_t *dest, int dstW, int uvalpha, int bgr)
{
    int uva1 = 4096 - uvalpha;
    for (int i = 0; i < dstW >> 1; i++) {
        int Y1 = (lum[2 * i] + 64) >> 7;
        int Y2 = (lum[2 * i + 1] + 64) >> 7;
        int U, V;
        if (!uvalpha) {
            U = (cu[0][i] + 64) >> 7;
            V = (cv[0][i] + 64) >> 7;
        } else {
            U = (cu[0][i] * uva1 + cu[1][i] * uvalpha + (128 << 11)) >> 19;
            V = (cv[0][i] * uva1 + cv[1][i] * uvalpha + (128 << 11)) >> 19;
        }
        put_rgb(dest + 2 * px_bytes(bgr) * i,                 l, Y1, U, V, bgr);
        put_rgb(dest + 2 * px_bytes(bgr) * i + px_bytes(bgr), l, Y2, U, V, bgr);
    }
}

/* the three packed-output members, one line (yuv2packedX / yuv2packed2 / yuv2packed1 of a packed-RGB context) */
/* yuv2rgb_write_full (libswscale/output.c:1998-2051) for the 24- and 32-bit layouts: Y, U, V in the writers' 10-bit-shifted scale */
static void put_rgb_full(uint8_t *d, const int *k, int Y, int U, int V, int bgr)
{
    uint32_t y = (uint32_t)(Y - k[1]) * (uint32_t)k[0] + (1U << 21);
    int R = (int)(y + (uint32_t)V * (uint32_t)k[2]);
    int G = (int)(y + (uint32_t)V * (uint32_t)k[3] + (uint32_t)U * (uint32_t)k[4]);
    int B = (int)(y + (uint32_t)U * (uint32_t)k[5]);
    if ((R | G | B) & 0xC0000000) { /* av_clip_uintp2(., 30) */
        R = R & ~((1 << 30) - 1) ? (~R >> 31) & ((1 << 30) - 1) : R;
        G = G & ~((1 << 30) - 1) ? (~G >> 31) & ((1 << 30) - 1) : G;
        B = B & ~((1 << 30) - 1) ? (~B >> 31) & ((1 << 30) - 1) : B;
    }
    const uint8_t r = (uint8_t)(R >> 22), g = (uint8_t)(G >> 22), b = (uint8_t)(B >> 22);
    switch (bgr) {
    case 0: d[0] = r; d[1] = g; d[2] = b; break;
    case 1: d[0] = b; d[1] = g; d[2] = r; break;
    case 2: d[0] = 255; d[1] = r; d[2] = g; d[3] = b; break;
    case 3: d[0] = r; d[1] = g; d[2] = b; d[3] = 255; break;
    case 4: d[0] = 255; d[1] = b; d[2] = g; d[3] = r; break;
    default: d[0] = b; d[1] = g; d[2] = r; d[3] = 255; break;
    }
}
/* one line of yuv2rgb_full_X / _2 / _1 (output.c:2160-2310): a chroma sample per pixel; mode 0: X, 1: the two-row blend (yalpha,
 * uvalpha), 2: one luma row (uvalpha 0: one chroma row, else the blend) */
static void rgb_full_line(const int *k, int mode, const int16_t *lf, const int16_t *const *lum, int lfs, const int16_t *cf,
                          const int16_t *const *cu, const int16_t *const *cv, int cfs, int yalpha, int uvalpha, uint8_t *dest, int dstW, int bgr)
{
    for (int i = 0; i < dstW; i++) {
        int Y, U, V;
        if (mode == 0) {
            uint32_t y = 1 << 9, u = (1 << 9) - (128 << 19), v = (1 << 9) - (128 << 19);
            for (int j = 0; j < lfs; j++)
                y += (uint32_t)(lum[j][i] * (int)lf[j]);
            for (int j = 0; j < cfs; j++) {
                u += (uint32_t)(cu[j][i] * (int)cf[j]);
                v += (uint32_t)(cv[j][i] * (int)cf[j]);
            }
            Y = (int32_t)y >> 10; U = (int32_t)u >> 10; V = (int32_t)v >> 10;
        } else if (mode == 1) {
            Y = (lum[0][i] * (4096 - yalpha) + lum[1][i] * yalpha) >> 10;
            U = (cu[0][i] * (4096 - uvalpha) + cu[1][i] * uvalpha - (128 << 19)) >> 10;
            V = (cv[0][i] * (4096 - uvalpha) + cv[1][i] * uvalpha - (128 << 19)) >> 10;
        } else {
            Y = lum[0][i] * 4;
            if (!uvalpha) {
                U = (cu[0][i] - (128 << 7)) * 4;
                V = (cv[0][i] - (128 << 7)) * 4;
            } else {
                U = (cu[0][i] * (4096 - uvalpha) + cu[1][i] * uvalpha - (128 << 19)) >> 10;
                V = (cv[0][i] * (4096 - uvalpha) + cv[1][i] * uvalpha - (128 << 19)) >> 10;
            }
        }
        put_rgb_full(dest + px_bytes(bgr) * i, k, Y, U, V, bgr);
    }
}

void ffo_yuv2rgb_X(const FfoYuv2RgbLuts *l, const int16_t *lf, const int16_t *const *lum, int lfs, const int16_t *cf,
                   const int16_t *const *cu, const int16_t *const *cv, int cfs, uint8_t *dest, int dstW, int layout)
{
    rgb24_X(l, lf, lum, lfs, cf, cu, cv, cfs, dest, dstW, layout);
}
void ffo_yuv2rgb_2(const FfoYuv2RgbLuts *l, const int16_t *const lum[2], const int16_t *const cu[2], const int16_t *const cv[2], uint8_t *dest,
                   int dstW, int yalpha, int uvalpha, int layout)
{
    rgb24_2(l, lum, cu, cv, dest, dstW, yalpha, uvalpha, layout);
}
void ffo_yuv2rgb_1(const FfoYuv2RgbLuts *l, const int16_t *lum, const int16_t *const cu[2], const int16_t *const cv[2], uint8_t *dest, int dstW,
                   int uvalpha, int layout)
{
    rgb24_1(l, lum, cu, cv, dest, dstW, uvalpha, layout);
}

/* ---------------------------------------------------------------------------------------------
 * Whole-frame scaled conversion: the net effect of ff_swscale() (libswscale/swscale.c:263-567) with
 * its slice ring buffers (slice.c) for 8-bit sources: out = V(H(in)) with table-driven indices.
 *   H: lum_h_scale/chr_h_scale (hscale.c:39,168) after nv12ToUV_c de-interleave (input.c:936)
 *   V: lum_planar_vscale / chr_planar_vscale / packed_vscale dispatch (vscale.c:41-171)
 * dither is sws_pb_64 for <= 8-bit sources (swscale.c:54,385-387).
 * ------------------------------------------------------------------------------------------- */
static int is_nv(int fmt) { return fmt == FFO_PIX_FMT_NV12 || fmt == FFO_PIX_FMT_NV21; }
static int rgb_layout(int fmt)
{
    switch (fmt) {
    case FFO_PIX_FMT_RGB24: return 0;
    case FFO_PIX_FMT_BGR24: return 1;
    case FFO_PIX_FMT_ARGB:  return 2;
    case FFO_PIX_FMT_RGBA:  return 3;
    case FFO_PIX_FMT_ABGR:  return 4;
    case FFO_PIX_FMT_BGRA:  return 5;
    }
    return -1;
}
static int is_rgb(int fmt) { return rgb_layout(fmt) >= 0; }

/* solve_range_convert / init_range_convert_constants (libswscale/swscale.c:568-624) */
static void range_solve(unsigned src_min, unsigned src_max, unsigned dst_min, unsigned dst_max, int src_shift, int mult_shift,
                        uint32_t *coeff, int64_t *offset)
{
    const unsigned src_range = (uint16_t)(src_max - src_min), dst_range = (uint16_t)(dst_max - dst_min);
    const int total_shift = mult_shift + src_shift;
    const uint64_t q = ((uint64_t)dst_range << total_shift) / src_range;
    *coeff = (uint32_t)-((-(int64_t)q) >> src_shift); /* AV_CEIL_RSHIFT */
    *offset = ((int64_t)dst_max << total_shift) - ((int64_t)src_max << src_shift) * *coeff + (1U << (mult_shift - 1));
}
void ffo_sws_range_constants(int src_range, int dst_depth, uint32_t *lum_coeff, int64_t *lum_offset, uint32_t *chr_coeff, int64_t *chr_offset)
{
    const int bit_depth = dst_depth > 16 ? 16 : dst_depth;
    const int src_bits = bit_depth <= 14 ? 15 : 19, src_shift = src_bits - bit_depth, mult_shift = bit_depth <= 14 ? 14 : 18;
    const unsigned mpeg_min = 16U << (bit_depth - 8), mpeg_max_lum = 235U << (bit_depth - 8), mpeg_max_chr = 240U << (bit_depth - 8);
    const unsigned jpeg_max = (1U << bit_depth) - 1;
    if (src_range) {
        range_solve(0, jpeg_max, mpeg_min, mpeg_max_lum, src_shift, mult_shift, lum_coeff, lum_offset);
        range_solve(0, jpeg_max, mpeg_min, mpeg_max_chr, src_shift, mult_shift, chr_coeff, chr_offset);
    } else {
        range_solve(mpeg_min, mpeg_max_lum, 0, jpeg_max, src_shift, mult_shift, lum_coeff, lum_offset);
        range_solve(mpeg_min, mpeg_max_chr, 0, jpeg_max, src_shift, mult_shift, chr_coeff, chr_offset);
    }
}
/* lumRangeToJpeg_c / lumRangeFromJpeg_c and the chroma pair on one int16 line (swscale.c:160-207) */
static void range15_line(int16_t *dst, int width, uint32_t coeff_, int64_t offset_, int to_jpeg)
{
    const uint16_t coeff = (uint16_t)coeff_;
    const int32_t offset = (int32_t)offset_;
    for (int i = 0; i < width; i++) {
        const int v = (dst[i] * coeff + offset) >> 14;
        dst[i] = (int16_t)(to_jpeg && v > (1 << 15) - 1 ? (1 << 15) - 1 : v);
    }
}

/*
 * The alpha byte of a 32-bit packed RGB target whose SOURCE has an alpha plane (needAlpha, libswscale/utils.c:1398): the plane goes
 * through the luma's horizontal bank (lum_h_scale on plane 3, hscale.c:63-79) and the packed writer forms A beside Y from the same
 * lines with the luma's vertical coefficients — each of the writers in its own way (libswscale/output.c):
 *   yuv2rgb_X     :1823-1835   (sum + (1 << 18)) >> 19, and the PAIR (A1, A2) clipped when (A1 | A2) & 0x100
 *   yuv2rgb_2     :1875-1880   (a0 * (4096 - yalpha) + a1 * yalpha) >> 19, clipped
 *   yuv2rgb_1     :1911-1916   uvalpha == 0: (a * 255 + 16384) >> 15;   :1939-1944  else (a + 64) >> 7; clipped
 *   yuv2rgb_full_X :2193-2200  (sum + (1 << 18)) >> 19;  _2 :2241-2245  (... + (1 << 18)) >> 19;  _1 :2278-2302  (a + 64) >> 7;
 *                              each clipped when A & 0x100
 * Called after ffo_sws_scale_frame() has written the picture (alpha 255): overwrites the alpha bytes.  alpha: the source's plane 3.
 */
int ffo_sws_rgba_alpha(const FfoSwsTables *t, const uint8_t *alpha, int alphaStride, uint8_t *dst, int dstStride)
{
    const int srcH = t->srcH, dstW = t->dstW, dstH = t->dstH, lpitch = dstW + 8;
    const int lay = rgb_layout(t->dstFormat), lfs = t->vLum.size, cfs = t->vChr.size;
    const int abyte = lay == 2 || lay == 4 ? 0 : 3;
    int16_t *ha;
    if (lay < 2 || lay > 5)
        return -1;
    ha = malloc(sizeof(int16_t) * (size_t)lpitch * srcH);
    if (!ha)
        return -1;
    for (int y = 0; y < srcH; y++)
        ffo_hscale8to15(ha + (size_t)y * lpitch, dstW, alpha + (ptrdiff_t)y * alphaStride, t->hLum.filter, t->hLum.pos, t->hLum.size);
    for (int y = 0; y < dstH; y++) {
        const uint16_t *lf = (const uint16_t *)t->vLum.filter + (size_t)y * lfs;
        const uint16_t *cf = (const uint16_t *)t->vChr.filter + (size_t)y * cfs;
        const int16_t *a0 = ha + (size_t)t->vLum.pos[y] * lpitch;
        uint8_t *d = dst + (ptrdiff_t)y * dstStride;
        const int chr_bilin = cfs == 2 && cf[1] + cf[0] == 4096 && cf[1] <= 4096U;
        const int lum_bilin = lfs == 2 && lf[1] + lf[0] == 4096 && lf[1] <= 4096U;
        const int one = lfs == 1 && (cfs == 1 || chr_bilin), two = !one && lum_bilin && chr_bilin;
        const int uvalpha = one && cfs == 2 ? cf[1] : 0;
        const int npx = t->full_chr ? dstW : dstW & ~1;
        for (int i = 0; i < npx; i++) {
            int A;
            if (one) {
                A = t->full_chr || uvalpha ? (a0[i] + 64) >> 7 : (a0[i] * 255 + 16384) >> 15;
            } else if (two) {
                A = (a0[i] * (4096 - (int)lf[1]) + a0[i + lpitch] * (int)lf[1] + (t->full_chr ? 1 << 18 : 0)) >> 19;
            } else {
                uint32_t acc = 1 << 18;
                for (int j = 0; j < lfs; j++)
                    acc += (uint32_t)(a0[i + (size_t)j * lpitch] * (int)(int16_t)lf[j]);
                A = (int32_t)acc >> 19;
            }
            if (t->full_chr) {
                if (A & 0x100)
                    A = clip_u8(A);
            } else if (one || two) {
                A = clip_u8(A);
            } /* (the X writer decides for the PAIR: below) */
            d[4 * i + abyte] = (uint8_t)A;
            if (!t->full_chr && !one && !two && (i & 1)) {
                /* recompute both of the pair unclipped, then clip both if either carries bit 8 */
                int P[2];
                for (int e = 0; e < 2; e++) {
                    uint32_t acc = 1 << 18;
                    for (int j = 0; j < lfs; j++)
                        acc += (uint32_t)(a0[i - 1 + e + (size_t)j * lpitch] * (int)(int16_t)lf[j]);
                    P[e] = (int32_t)acc >> 19;
                }
                if ((P[0] | P[1]) & 0x100) {
                    P[0] = clip_u8(P[0]);
                    P[1] = clip_u8(P[1]);
                }
                d[4 * (i - 1) + abyte] = (uint8_t)P[0];
                d[4 * i + abyte] = (uint8_t)P[1];
            }
        }
    }
    free(ha);
    return dstH;
}

int ffo_sws_scale_frame(const FfoSwsTables *t, const uint8_t *const src[3], const int srcStride[3],
                        uint8_t *const dst[3], const int dstStride[3])
{
    static const uint8_t d64[8] = { 64, 64, 64, 64, 64, 64, 64, 64 };
    const int srcW = t->srcW, srcH = t->srcH, dstW = t->dstW, dstH = t->dstH;
    /* av_pix_fmt_get_chroma_sub_sample(): 4:4:4 has none, 4:2:2 only horizontally (libswscale/utils.c:1265,1393-1394) */
    const int hs = t->srcFormat == FFO_PIX_FMT_YUV444P ? 0 : 1, vs = t->srcFormat == FFO_PIX_FMT_YUV444P || t->srcFormat == FFO_PIX_FMT_YUV422P ? 0 : 1;
    const int chrSrcW = -((-srcW) >> hs), chrSrcH = -((-srcH) >> vs);
    const int chrDstW = t->hChr.n, chrDstH = t->vChr.n;
    const int lpitch = dstW + 8, cpitch = chrDstW + 8;
    int16_t *hl = malloc(sizeof(int16_t) * (size_t)lpitch * srcH);
    int16_t *hu = malloc(sizeof(int16_t) * (size_t)cpitch * chrSrcH);
    int16_t *hv = malloc(sizeof(int16_t) * (size_t)cpitch * chrSrcH);
    uint8_t *tu = malloc((size_t)chrSrcW + 8), *tv = malloc((size_t)chrSrcW + 8);
    const int16_t **rows = malloc(sizeof(*rows) * 3 * (size_t)(t->vLum.size + t->vChr.size + 2));
    FfoYuv2RgbLuts *luts = NULL;
    int ret = -1;

    if (!hl || !hu || !hv || !tu || !tv || !rows)
        goto done;

    for (int y = 0; y < srcH; y++)
        ffo_hscale8to15(hl + (size_t)y * lpitch, dstW, src[0] + (ptrdiff_t)y * srcStride[0], t->hLum.filter,
                        t->hLum.pos, t->hLum.size);
    for (int y = 0; y < chrSrcH; y++) {
        const uint8_t *pu, *pv;
        if (is_nv(t->srcFormat)) {
            const uint8_t *p = src[1] + (ptrdiff_t)y * srcStride[1];
            int sw = t->srcFormat == FFO_PIX_FMT_NV21;
            for (int i = 0; i < chrSrcW; i++) {
                tu[i] = p[2 * i + sw];
                tv[i] = p[2 * i + !sw];
            }
            pu = tu;
            pv = tv;
        } else {
            pu = src[1] + (ptrdiff_t)y * srcStride[1];
            pv = src[2] + (ptrdiff_t)y * srcStride[2];
        }
        ffo_hscale8to15(hu + (size_t)y * cpitch, chrDstW, pu, t->hChr.filter, t->hChr.pos, t->hChr.size);
        ffo_hscale8to15(hv + (size_t)y * cpitch, chrDstW, pv, t->hChr.filter, t->hChr.pos, t->hChr.size);
    }
    /* c->lumConvertRange / chrConvertRange on every horizontal line (ff_swscale via lum_convert / chr_convert, hscale.c) */
    if (t->src_range != t->dst_range && !is_rgb(t->dstFormat)) {
        for (int y = 0; y < srcH; y++)
            range15_line(hl + (size_t)y * lpitch, dstW, t->lum_rc_coeff, t->lum_rc_offset, !t->src_range);
        for (int y = 0; y < chrSrcH; y++) {
            range15_line(hu + (size_t)y * cpitch, chrDstW, t->chr_rc_coeff, t->chr_rc_offset, !t->src_range);
            range15_line(hv + (size_t)y * cpitch, chrDstW, t->chr_rc_coeff, t->chr_rc_offset, !t->src_range);
        }
    }

    if (is_rgb(t->dstFormat)) {
        const int bgr = rgb_layout(t->dstFormat);
        const int lfs = t->vLum.size, cfs = t->vChr.size;
        const int16_t **lr = rows, **ur = rows + lfs + 1, **vr = ur + cfs + 1;
        luts = malloc(sizeof(*luts));
        if (!luts)
            goto done;
        ffo_yuv2rgb_luts_init(luts, &t->k);
        for (int y = 0; y < dstH; y++) {
            /* chrDstVSubSample == 0 for RGB targets, so chroma row index == y */
            const uint16_t *lf = (const uint16_t *)t->vLum.filter + (size_t)y * lfs;
            const uint16_t *cf = (const uint16_t *)t->vChr.filter + (size_t)y * cfs;
            uint8_t *d = dst[0] + (ptrdiff_t)y * dstStride[0];
            for (int j = 0; j < lfs; j++)
                lr[j] = hl + (size_t)(t->vLum.pos[y] + j) * lpitch;
            for (int j = 0; j < cfs; j++) {
                ur[j] = hu + (size_t)(t->vChr.pos[y] + j) * cpitch;
                vr[j] = hv + (size_t)(t->vChr.pos[y] + j) * cpitch;
            }
            if (t->full_chr) { /* the same dispatch of packed_vscale() (vscale.c:126-170) over the full-chroma writers */
                if (lfs == 1 && cfs == 1)
                    rgb_full_line(t->full_coef, 2, NULL, lr, 1, NULL, ur, vr, 1, 0, 0, d, dstW, bgr);
                else if (lfs == 1 && cfs == 2 && cf[1] + cf[0] == 4096 && cf[1] <= 4096U)
                    rgb_full_line(t->full_coef, 2, NULL, lr, 1, NULL, ur, vr, 2, 0, cf[1], d, dstW, bgr);
                else if (lfs == 2 && cfs == 2 && lf[1] + lf[0] == 4096 && lf[1] <= 4096U && cf[1] + cf[0] == 4096 && cf[1] <= 4096U)
                    rgb_full_line(t->full_coef, 1, NULL, lr, 2, NULL, ur, vr, 2, lf[1], cf[1], d, dstW, bgr);
                else
                    rgb_full_line(t->full_coef, 0, (const int16_t *)lf, lr, lfs, (const int16_t *)cf, ur, vr, cfs, 0, 0, d, dstW, bgr);
            } else if (lfs == 1 && cfs == 1) {
                rgb24_1(luts, lr[0], ur, vr, d, dstW, 0, bgr);
            } else if (lfs == 1 && cfs == 2 && cf[1] + cf[0] == 4096 && cf[1] <= 4096U) {
                rgb24_1(luts, lr[0], ur, vr, d, dstW, cf[1], bgr);
            } else if (lfs == 2 && cfs == 2 && lf[1] + lf[0] == 4096 && lf[1] <= 4096U && cf[1] + cf[0] == 4096 &&
                       cf[1] <= 4096U) {
                rgb24_2(luts, lr, ur, vr, d, dstW, lf[1], cf[1], bgr);
            } else {
                rgb24_X(luts, (const int16_t *)lf, lr, lfs, (const int16_t *)cf, ur, vr, cfs, d, dstW, bgr);
            }
        }
    } else {
        const int lfs = t->vLum.size, cfs = t->vChr.size;
        const int16_t **lr = rows, **ur = rows + lfs + 1, **vr = ur + cfs + 1;
        for (int y = 0; y < dstH; y++) {
            for (int j = 0; j < lfs; j++)
                lr[j] = hl + (size_t)(t->vLum.pos[y] + j) * lpitch;
            if (lfs == 1)
                ffo_yuv2plane1_8(lr[0], dst[0] + (ptrdiff_t)y * dstStride[0], dstW, d64, 0);
            else
                ffo_yuv2planeX8(t->vLum.filter + (size_t)y * lfs, lfs, lr, dst[0] + (ptrdiff_t)y * dstStride[0], dstW,
                                d64, 0);
        }
        for (int y = 0; y < chrDstH; y++) {
            const int16_t *cf = t->vChr.filter + (size_t)y * cfs;
            for (int j = 0; j < cfs; j++) {
                ur[j] = hu + (size_t)(t->vChr.pos[y] + j) * cpitch;
                vr[j] = hv + (size_t)(t->vChr.pos[y] + j) * cpitch;
            }
            if (is_nv(t->dstFormat)) {
                ffo_yuv2nv12cX(t->dstFormat == FFO_PIX_FMT_NV21, d64, cf, cfs, ur, vr,
                               dst[1] + (ptrdiff_t)y * dstStride[1], chrDstW);
            } else if (cfs == 1) {
                ffo_yuv2plane1_8(ur[0], dst[1] + (ptrdiff_t)y * dstStride[1], chrDstW, d64, 0);
                ffo_yuv2plane1_8(vr[0], dst[2] + (ptrdiff_t)y * dstStride[2], chrDstW, d64, 3);
            } else {
                ffo_yuv2planeX8(cf, cfs, ur, dst[1] + (ptrdiff_t)y * dstStride[1], chrDstW, d64, 0);
                ffo_yuv2planeX8(cf, cfs, vr, dst[2] + (ptrdiff_t)y * dstStride[2], chrDstW, d64, 3);
            }
        }
    }
    ret = dstH;
done:
    free(hl); free(hu); free(hv); free(tu); free(tv); free(rows); free(luts);
    return ret;
}
