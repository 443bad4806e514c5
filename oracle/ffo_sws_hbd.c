/*
 * ffo_sws_hbd.c — CPU restatement of the reference's legacy scaler above 8 bits: hScale16To15_c / hScale16To19_c / hScale8To19_c
 * (libswscale/swscale.c:69-160), yuv2plane1 / yuv2planeX at 9..14 and 16 bits (libswscale/output.c:150-200,330-360), the P010 /
 * P016 readers and writers (input.c p010LEToY_c / p010LEToUV_c, output.c yuv2p01xl1 / lX / cX and yuv2nv12cX_16), and the Bayer
 * dither swscale switches on for 8-bit targets fed from deeper sources (swscale.c: should_dither; ff_dither_8x8_128).
 * TEST INFRASTRUCTURE ONLY (see ffo.h).  Pinned against oracle/_ref — the reference's sws_scale() on the real pixel formats — by
 * tests/test_oracle_vs_ref_sws_hbd.py.
 *
 * A format is described by (depth, layout): layout 0 planar little-endian samples in the low bits (yuv4xxp<depth>le, 8-bit planar
 * when depth == 8), 1 semi-planar with the samples in the HIGH bits (p010le / p012le / p016le), 2 semi-planar 8-bit (nv12).
 * Chroma subsampling comes with the tables (hChr.n / vChr.n).  Scaled contexts only: the reference converts equal-size pictures
 * through its special converters (swscale_unscaled.c), which are other arithmetic.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ffo.h"

/* ff_dither_8x8_128: libswscale/swscale.c:42-52: the 8 x 8 ordered-dither matrix (values 0..126; row 8 repeats row 0) */
static const uint8_t dither_8x8_128[9][8] = {
    {  36, 68,  60, 92,  34, 66,  58, 90, },
    { 100,  4, 124, 28,  98,  2, 122, 26, },
    {  52, 84,  44, 76,  50, 82,  42, 74, },
    { 116, 20, 108, 12, 114, 18, 106, 10, },
    {  32, 64,  56, 88,  38, 70,  62, 94, },
    {  96,  0, 120, 24, 102,  6, 126, 30, },
    {  48, 80,  40, 72,  54, 86,  46, 78, },
    { 112, 16, 104,  8, 118, 22, 110, 14, },
    {  36, 68,  60, 92,  34, 66,  58, 90, },
};

static inline int clipu(int v, int bits) { const int m = (1 << bits) - 1; return v < 0 ? 0 : v > m ? m : v; }

/* one source sample as the horizontal scaler sees it */
static inline int src_at(const uint8_t *row, int depth, int layout, int chan, int i)
{
    if (depth == 8)
        return layout == 2 ? row[2 * i + chan] : row[i];
    const uint16_t *p = (const uint16_t *)row;
    if (layout == 1) /* p01x: interleaved for chroma (chan 0 / 1), samples in the high bits (input.c: >> (16 - depth)) */
        return p[chan < 0 ? i : 2 * i + chan] >> (16 - depth);
    return p[i];
}

/* hScale{8,16}To{15,19}_c: `wide` = the 19-bit path (target depth 16) */
static void hscale(int32_t *dst, int dstW, const uint8_t *row, int depth, int layout, int chan, const int16_t *filter, const int32_t *pos,
                   int fs, int wide)
{
    const int sh = depth == 8 ? (wide ? 3 : 7) : (wide ? depth - 1 - 4 : depth - 1);
    const int lim = wide ? (1 << 19) - 1 : (1 << 15) - 1;
    for (int i = 0; i < dstW; i++) {
        int val = 0; /* the reference accumulates in int with wrap-around on overflow: unsigned arithmetic reproduces it */
        unsigned acc = 0;
        for (int j = 0; j < fs; j++)
            acc += (unsigned)(src_at(row, depth, layout, chan, pos[i] + j) * (int)filter[fs * i + j]);
        val = (int)acc >> sh;
        dst[i] = val < lim ? val : lim;
    }
}

/* c->lumConvertRange / chrConvertRange on one horizontal line: the 15-bit forms work on int16 lines with a 16-bit coefficient and a
 * 32-bit offset (lumRangeToJpeg_c ..., swscale.c:160-207), the 19-bit ones on int32 lines with 64-bit products
 * (lumRangeToJpeg16_c ..., :209-255) */
static void range_line(int32_t *dst, int width, uint32_t coeff, int64_t offset, int to_jpeg, int wide)
{
    for (int i = 0; i < width; i++) {
        if (!wide) {
            const int v = ((int16_t)dst[i] * (int)(uint16_t)coeff + (int32_t)offset) >> 14;
            dst[i] = (int16_t)(to_jpeg && v > (1 << 15) - 1 ? (1 << 15) - 1 : v);
        } else {
            const int v = (int)(((int64_t)dst[i] * coeff + offset) >> 18);
            dst[i] = to_jpeg && v > (1 << 19) - 1 ? (1 << 19) - 1 : v;
        }
    }
}

/* one output sample from its vertical taps */
static inline int vout(const int32_t *const *rows, const int16_t *vf, int vfs, int x, int ddepth, int wide, int dither)
{
    if (ddepth == 8) {
        if (vfs == 1)
            return clipu((rows[0][x] + dither) >> 7, 8);
        unsigned acc = (unsigned)dither << 12;
        for (int j = 0; j < vfs; j++)
            acc += (unsigned)(rows[j][x] * (int)vf[j]);
        return clipu((int)acc >> 19, 8);
    }
    if (wide) { /* yuv2plane1_16 / yuv2planeX_16 */
        if (vfs == 1)
            return clipu((rows[0][x] + 4) >> 3, 16);
        unsigned acc = (1u << 14) - 0x40000000u;
        for (int j = 0; j < vfs; j++)
            acc += (unsigned)rows[j][x] * (unsigned)(int)vf[j];
        int v = (int)acc >> 15;
        v = v < -32768 ? -32768 : v > 32767 ? 32767 : v;
        return 0x8000 + v;
    }
    if (vfs == 1) {
        const int shift = 15 - ddepth;
        return clipu((rows[0][x] + (1 << (shift - 1))) >> shift, ddepth);
    }
    const int shift = 11 + 16 - ddepth;
    unsigned acc = 1u << (shift - 1);
    for (int j = 0; j < vfs; j++)
        acc += (unsigned)(rows[j][x] * (int)vf[j]);
    return clipu((int)acc >> shift, ddepth);
}

static inline void put(uint8_t *row, int depth, int layout, int chan, int i, int v)
{
    if (depth == 8) {
        if (layout == 2) row[2 * i + chan] = (uint8_t)v; else row[i] = (uint8_t)v;
        return;
    }
    uint16_t *p = (uint16_t *)row;
    if (layout == 1)
        p[chan < 0 ? i : 2 * i + chan] = (uint16_t)(v << (16 - depth));
    else
        p[i] = (uint16_t)v;
}

/* src / dst: plane pointers as the format has them (planar: Y, U, V; semi-planar: Y, UV); dlayout 3 (round 6): ONE packed 8-bit RGB plane of
 * t->dstFormat, ddepth 8 */
int ffo_sws_scale_frame_hbd(const FfoSwsTables *t, int sdepth, int slayout, int ddepth, int dlayout, const uint8_t *const src[3],
                            const int srcStride[3], uint8_t *const dst[3], const int dstStride[3])
{
    const int srcH = t->srcH, dstW = t->dstW, dstH = t->dstH;
    const int chrDstW = t->hChr.n, chrDstH = t->vChr.n;
    const int chrSrcH = chrDstH == dstH ? /* vChr positions tell: */ 0 : 0;
    (void)chrSrcH;
    const int wide = ddepth == 16;
    /* sdepth | 0x100: the lines are a packed RGB source's converter output (ffo_sws_rgbin.c): 14-bit samples, and no dither — swscale.c's
     * should_dither = isNBPS(src) || is16BPS(src) looks at the caller's source format */
    const int from_rgb = sdepth & 0x100;
    sdepth &= 0xff;
    const int dith = ddepth == 8 && sdepth > 8 && !from_rgb;
    /* chroma source rows: the vertical chroma bank's reach */
    int csh = 0;
    for (int y = 0; y < chrDstH; y++)
        if (t->vChr.pos[y] + t->vChr.size > csh)
            csh = t->vChr.pos[y] + t->vChr.size;
    const size_t lp = (size_t)dstW + 8, cp = (size_t)chrDstW + 8;
    int32_t *hl = malloc(sizeof(int32_t) * lp * srcH), *hu = malloc(sizeof(int32_t) * cp * csh), *hv = malloc(sizeof(int32_t) * cp * csh);
    const int32_t **rows = malloc(sizeof(*rows) * (size_t)(t->vLum.size + t->vChr.size + 2));
    if (!hl || !hu || !hv || !rows) { free(hl); free(hu); free(hv); free(rows); return -1; }
    for (int y = 0; y < srcH; y++)
        hscale(hl + y * lp, dstW, src[0] + (ptrdiff_t)y * srcStride[0], sdepth, slayout == 2 ? 0 : slayout, -1, t->hLum.filter, t->hLum.pos,
               t->hLum.size, wide);
    for (int y = 0; y < csh; y++) {
        const int semi = slayout != 0;
        const uint8_t *ru = src[1] + (ptrdiff_t)y * srcStride[1], *rv = semi ? ru : src[2] + (ptrdiff_t)y * srcStride[2];
        hscale(hu + y * cp, chrDstW, ru, sdepth, slayout, semi ? 0 : -1, t->hChr.filter, t->hChr.pos, t->hChr.size, wide);
        hscale(hv + y * cp, chrDstW, rv, sdepth, slayout, semi ? 1 : -1, t->hChr.filter, t->hChr.pos, t->hChr.size, wide);
    }
    if (t->src_range != t->dst_range) {
        for (int y = 0; y < srcH; y++)
            range_line(hl + y * lp, dstW, t->lum_rc_coeff, t->lum_rc_offset, !t->src_range, wide);
        for (int y = 0; y < csh; y++) {
            range_line(hu + y * cp, chrDstW, t->chr_rc_coeff, t->chr_rc_offset, !t->src_range, wide);
            range_line(hv + y * cp, chrDstW, t->chr_rc_coeff, t->chr_rc_offset, !t->src_range, wide);
        }
    }
    if (dlayout == 3) {
        /* a packed 8-bit RGB target (t->dstFormat) fed from the deeper source: the 15-bit lines go to the writers packed_vscale() picks per
         * row (libswscale/vscale.c:126-170; yuv2rgb_X / _2 / _1, output.c:1789-1939) — ffo_sws.c's, on int16 copies of the lines */
        const int lfs = t->vLum.size, cfs = t->vChr.size;
        const int lay = t->dstFormat == 2 ? 0 : t->dstFormat == 3 ? 1 : t->dstFormat == 25 ? 2 : t->dstFormat == 26 ? 3 : t->dstFormat == 27 ? 4 : 5;
        int16_t *l16 = malloc(sizeof(int16_t) * lp * srcH), *u16 = malloc(sizeof(int16_t) * cp * csh), *v16 = malloc(sizeof(int16_t) * cp * csh);
        const int16_t **lr = malloc(sizeof(*lr) * (size_t)(lfs + 2 * cfs + 6)), **ur = lr + lfs + 1, **vr = ur + cfs + 1;
        FfoYuv2RgbLuts *luts = malloc(sizeof(*luts));
        if (!l16 || !u16 || !v16 || !lr || !luts || wide) { free(l16); free(u16); free(v16); free(lr); free(luts); free(hl); free(hu); free(hv); free(rows); return -1; }
        for (size_t i = 0; i < lp * srcH; i++) l16[i] = (int16_t)hl[i];
        for (size_t i = 0; i < cp * csh; i++) { u16[i] = (int16_t)hu[i]; v16[i] = (int16_t)hv[i]; }
        ffo_yuv2rgb_luts_init(luts, &t->k);
        for (int y = 0; y < dstH; y++) {
            const uint16_t *lf = (const uint16_t *)t->vLum.filter + (size_t)y * lfs, *cf = (const uint16_t *)t->vChr.filter + (size_t)y * cfs;
            uint8_t *d = dst[0] + (ptrdiff_t)y * dstStride[0];
            for (int j = 0; j < lfs; j++)
                lr[j] = l16 + (size_t)(t->vLum.pos[y] + j) * lp;
            for (int j = 0; j < cfs; j++) {
                ur[j] = u16 + (size_t)(t->vChr.pos[y] + j) * cp;
                vr[j] = v16 + (size_t)(t->vChr.pos[y] + j) * cp;
            }
            if (lfs == 1 && cfs == 1)
                ffo_yuv2rgb_1(luts, lr[0], ur, vr, d, dstW, 0, lay);
            else if (lfs == 1 && cfs == 2 && cf[1] + cf[0] == 4096 && cf[1] <= 4096U)
                ffo_yuv2rgb_1(luts, lr[0], ur, vr, d, dstW, cf[1], lay);
            else if (lfs == 2 && cfs == 2 && lf[1] + lf[0] == 4096 && lf[1] <= 4096U && cf[1] + cf[0] == 4096 && cf[1] <= 4096U)
                ffo_yuv2rgb_2(luts, lr, ur, vr, d, dstW, lf[1], cf[1], lay);
            else
                ffo_yuv2rgb_X(luts, (const int16_t *)lf, lr, lfs, (const int16_t *)cf, ur, vr, cfs, d, dstW, lay);
        }
        free(l16); free(u16); free(v16); free(lr); free(luts);
        free(hl); free(hu); free(hv); free(rows);
        return 0;
    }
    for (int y = 0; y < dstH; y++) {
        for (int j = 0; j < t->vLum.size; j++)
            rows[j] = hl + (size_t)(t->vLum.pos[y] + j) * lp;
        uint8_t *d = dst[0] + (ptrdiff_t)y * dstStride[0];
        for (int x = 0; x < dstW; x++)
            put(d, ddepth, dlayout == 2 ? 0 : dlayout, -1, x,
                vout(rows, t->vLum.filter + (size_t)y * t->vLum.size, t->vLum.size, x, ddepth, wide, dith ? dither_8x8_128[y & 7][x & 7] : 64));
    }
    for (int y = 0; y < chrDstH; y++) {
        const int semi = dlayout != 0;
        uint8_t *du = dst[1] + (ptrdiff_t)y * dstStride[1], *dv = semi ? du : dst[2] + (ptrdiff_t)y * dstStride[2];
        const int16_t *vf = t->vChr.filter + (size_t)y * t->vChr.size;
        for (int pl = 0; pl < 2; pl++) {
            const int32_t *h = pl ? hv : hu;
            for (int j = 0; j < t->vChr.size; j++)
                rows[j] = h + (size_t)(t->vChr.pos[y] + j) * cp;
            for (int x = 0; x < chrDstW; x++) {
                /* planar 8-bit chroma: U with dither offset 0, V with 3 (vscale.c chroma planes); the interleaved writer
                 * (yuv2nv12cX_c, output.c:495-529) uses chrDither[i & 7] for U and [(i + 3) & 7] for V as well */
                const int dz = dith ? dither_8x8_128[y & 7][(x + (pl ? 3 : 0)) & 7] : 64;
                put(pl ? dv : du, ddepth, dlayout, semi ? pl : -1, x, vout(rows, vf, t->vChr.size, x, ddepth, wide, dz));
            }
        }
    }
    free(hl); free(hu); free(hv); free(rows);
    return 0;
}
