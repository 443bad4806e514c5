/*
 * ffo_sws_rgbin.c — CPU restatement of libswscale's input stage for packed 8-bit RGB sources.  TEST INFRASTRUCTURE ONLY.
 *   rgb24ToY_c / bgr24ToY_c, rgb24ToUV_c, rgb24ToUV_half_c and the rgb32 / bgr32 (+ _1) templates   libswscale/input.c:264-400,1068-1190
 *   input_rgb2yuv_table for SWS_CS_DEFAULT (fill_rgb2yuv_table's closing branch)                     libswscale/utils.c:693-703
 *   chrSrcHSubSample for RGB sources                                                                 libswscale/utils.c:1340-1352
 * Every source line becomes int16 lines of 14-bit samples; the context then scales them with hScale16To15_c at sh = 13
 * (swscale.c:100-128), which is what oracle/ffo_sws_hbd.c does for a 14-bit planar source — with the dither of an 8-bit target off
 * (swscale.c:291 looks at the source FORMAT): ffo_sws_scale_frame_hbd() takes that as sdepth | 0x100.
 * Pinned to the reference's sws_scale() on whole conversions: tests/test_oracle_vs_ref_sws_rgbin.py.
 */
#include <stdint.h>

#include "ffo.h"

#define S 15 /* RGB2YUV_SHIFT */

void ffo_sws_rgb2yuv_default(int32_t t[9])
{
    t[0] =  ((int)(0.299 * 219 / 255 * (1 << S) + 0.5));
    t[1] =  ((int)(0.587 * 219 / 255 * (1 << S) + 0.5));
    t[2] =  ((int)(0.114 * 219 / 255 * (1 << S) + 0.5));
    t[3] = (-(int)(0.169 * 224 / 255 * (1 << S) + 0.5));
    t[4] = (-(int)(0.331 * 224 / 255 * (1 << S) + 0.5));
    t[5] =  ((int)(0.500 * 224 / 255 * (1 << S) + 0.5));
    t[6] =  ((int)(0.500 * 224 / 255 * (1 << S) + 0.5));
    t[7] = (-(int)(0.419 * 224 / 255 * (1 << S) + 0.5));
    t[8] = (-(int)(0.081 * 224 / 255 * (1 << S) + 0.5));
}

/* 1: the chroma converters read pixel pairs (the *_half_c forms) */
int ffo_sws_rgb_half(int srcW, int dstW, int chrDstHSubSample, int flags)
{
    return !(srcW & 1) && !(flags & 0x4000 /* SWS_FULL_CHR_H_INP */) && (dstW >> chrDstHSubSample) <= (srcW >> 1);
}

/* one frame: bpp 3 / 4, (ro, go, bo) the component bytes of a pixel; Y: w samples per row, U / V: w / 2 (half) or w; strides in bytes */
void ffo_sws_rgb_in(const uint8_t *src, ptrdiff_t stride, int w, int h, int bpp, int ro, int go, int bo, int half, const int32_t t[9],
                    uint16_t *Y, ptrdiff_t ys, uint16_t *U, uint16_t *V, ptrdiff_t cs)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + y * stride;
        int16_t *dy = (int16_t *)((uint8_t *)Y + y * ys), *du = (int16_t *)((uint8_t *)U + y * cs), *dv = (int16_t *)((uint8_t *)V + y * cs);
        for (int i = 0; i < w; i++) {
            const int r = s[i * bpp + ro], g = s[i * bpp + go], b = s[i * bpp + bo];
            dy[i] = (int16_t)((t[0] * r + t[1] * g + t[2] * b + (32 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
        }
        if (half) {
            for (int i = 0; i < w / 2; i++) {
                const int r = s[2 * i * bpp + ro] + s[(2 * i + 1) * bpp + ro], g = s[2 * i * bpp + go] + s[(2 * i + 1) * bpp + go],
                          b = s[2 * i * bpp + bo] + s[(2 * i + 1) * bpp + bo];
                du[i] = (int16_t)((unsigned)(t[3] * r + t[4] * g + t[5] * b + (256 << S) + (1 << (S - 6))) >> (S - 5));
                dv[i] = (int16_t)((unsigned)(t[6] * r + t[7] * g + t[8] * b + (256 << S) + (1 << (S - 6))) >> (S - 5));
            }
        } else {
            for (int i = 0; i < w; i++) {
                const int r = s[i * bpp + ro], g = s[i * bpp + go], b = s[i * bpp + bo];
                du[i] = (int16_t)((t[3] * r + t[4] * g + t[5] * b + (256 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
                dv[i] = (int16_t)((t[6] * r + t[7] * g + t[8] * b + (256 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
            }
        }
    }
}
