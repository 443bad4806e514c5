/*
 * ffo_mecmp.c — CPU restatement of me_cmp SAD/SATD and the exhaustive block search.
 * TEST INFRASTRUCTURE ONLY (see ffo.h).  Pinned against oracle/_ref and tests/golden.
 */
#include <stdint.h>
#include <stdlib.h>

#include "ffo.h"

/* pix_abs16_c / pix_abs8_c: libavcodec/me_cmp.c:117-143,272-290 */
int ffo_sad(int width, const uint8_t *a, const uint8_t *b, ptrdiff_t stride, int h)
{
    int s = 0;
    for (int y = 0; y < h; y++, a += stride, b += stride)
        for (int x = 0; x < width; x++)
            s += abs(a[x] - b[x]);
    return s;
}

/* in-place length-8 Walsh-Hadamard butterflies with element step `st` */
static void wht8(int *v, int st)
{
    for (int span = 1; span < 8; span <<= 1)
        for (int i = 0; i < 8; i++)
            if (!(i & span)) {
                int a = v[i * st], b = v[(i + span) * st];
                v[i * st] = a + b;
                v[(i + span) * st] = a - b;
            }
}

/*
 * hadamard8_diff8x8_c: libavcodec/me_cmp.c:514-562.  The reference runs three butterfly stages
 * along rows, two along columns and folds the last column stage into |x+y|+|x-y|; that is the sum
 * of absolute values of the full 8x8 Hadamard transform of (src - dst), un-normalised.
 */
int ffo_hadamard8_diff8x8(const uint8_t *dst, const uint8_t *src, ptrdiff_t stride)
{
    int t[64], sum = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            t[8 * y + x] = src[y * stride + x] - dst[y * stride + x];
    for (int y = 0; y < 8; y++)
        wht8(t + 8 * y, 1);
    for (int x = 0; x < 8; x++)
        wht8(t + x, 8);
    for (int i = 0; i < 64; i++)
        sum += abs(t[i]);
    return sum;
}

/* hadamard8_diff16_c via WRAPPER8_16_SQ: libavcodec/me_cmp.c:933-950 */
int ffo_hadamard8_diff16(const uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h)
{
    int s = ffo_hadamard8_diff8x8(dst, src, stride) + ffo_hadamard8_diff8x8(dst + 8, src + 8, stride);
    if (h == 16) {
        dst += 8 * stride;
        src += 8 * stride;
        s += ffo_hadamard8_diff8x8(dst, src, stride) + ffo_hadamard8_diff8x8(dst + 8, src + 8, stride);
    }
    return s;
}

static uint64_t block_cost(const uint8_t *cur, const uint8_t *ref, int linesize, int mb, int kind, int x_mb,
                           int y_mb, int x, int y)
{
    const uint8_t *c = cur + (ptrdiff_t)y_mb * linesize + x_mb;
    const uint8_t *r = ref + (ptrdiff_t)y * linesize + x;
    if (kind == 0)  /* ff_me_cmp_sad: libavfilter/motion_estimation.c:60-76 */
        return (uint64_t)ffo_sad(mb, c, r, linesize, mb);
    /* SATD variant: me_cmp argument order is (blk1 = current, blk2 = candidate) */
    if (mb == 16)
        return (uint64_t)ffo_hadamard8_diff16(c, r, linesize, 16);
    return (uint64_t)ffo_hadamard8_diff8x8(c, r, linesize);
}

/*
 * ff_me_search_esa + vf_mestimate's context: libavfilter/motion_estimation.c:32-40,78-95,
 * libavfilter/vf_mestimate.c:101,119-129.  mv[] must come in as {x_mb, y_mb}.
 */
uint64_t ffo_me_search_esa(const uint8_t *cur, const uint8_t *ref, int linesize, int width, int height, int mb_size,
                           int search_param, int cost_kind, int x_mb, int y_mb, int *mv)
{
    int log2mb = 0;
    while ((1 << log2mb) < mb_size)
        log2mb++;
    const int lim_x = ((width >> log2mb) - 1) << log2mb, lim_y = ((height >> log2mb) - 1) << log2mb;
    int x0 = x_mb - search_param > 0 ? x_mb - search_param : 0;
    int y0 = y_mb - search_param > 0 ? y_mb - search_param : 0;
    int x1 = x_mb + search_param < lim_x ? x_mb + search_param : lim_x;
    int y1 = y_mb + search_param < lim_y ? y_mb + search_param : lim_y;
    uint64_t best = block_cost(cur, ref, linesize, mb_size, cost_kind, x_mb, y_mb, x_mb, y_mb);
    if (!best)
        return best;
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) {
            uint64_t c = block_cost(cur, ref, linesize, mb_size, cost_kind, x_mb, y_mb, x, y);
            if (c < best) {
                best = c;
                mv[0] = x;
                mv[1] = y;
            }
        }
    return best;
}

/* SEARCH_MV(esa) over one frame pair: libavfilter/vf_mestimate.c:119-129 */
void ffo_me_esa_frame(const uint8_t *cur, const uint8_t *ref, int linesize, int width, int height, int mb_size,
                      int search_param, int cost_kind, int16_t *mv_out, uint32_t *cost_out)
{
    int log2mb = 0;
    while ((1 << log2mb) < mb_size)
        log2mb++;
    const int bw = width >> log2mb, bh = height >> log2mb;
    for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
            int mv[2] = { bx << log2mb, by << log2mb };
            uint64_t c = ffo_me_search_esa(cur, ref, linesize, width, height, mb_size, search_param, cost_kind,
                                           mv[0], mv[1], mv);
            mv_out[2 * (by * bw + bx)]     = (int16_t)mv[0];
            mv_out[2 * (by * bw + bx) + 1] = (int16_t)mv[1];
            cost_out[by * bw + bx] = (uint32_t)c;
        }
}

/* the half-pel SADs, SSE and NSSE: pix_abs{16,8}_{x2,y2,xy2}_c (libavcodec/me_cmp.c:184-370), sse{16,8}_c (:53-104), nsse{16,8}_c
 * (:387-440, the context-free weight 8).  kind = FFHIP_ME_SAD_X2 (2) .. FFHIP_ME_NSSE (6) */
int ffo_me_cmp_other(int kind, int width, const uint8_t *a, const uint8_t *b, ptrdiff_t stride, int h)
{
    int r = 0;
    if (kind == 6) {
        int score2 = 0;
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < width; x++)
                r += (a[x] - b[x]) * (a[x] - b[x]);
            if (y + 1 < h)
                for (int x = 0; x < width - 1; x++) {
                    int da = a[x] - a[x + stride] - a[x + 1] + a[x + stride + 1];
                    int db = b[x] - b[x + stride] - b[x + 1] + b[x + stride + 1];
                    score2 += (da < 0 ? -da : da) - (db < 0 ? -db : db);
                }
            a += stride;
            b += stride;
        }
        return r + (score2 < 0 ? -score2 : score2) * 8;
    }
    for (int y = 0; y < h; y++, a += stride, b += stride)
        for (int x = 0; x < width; x++) {
            int q = b[x], d;
            if (kind == 5) { r += (a[x] - q) * (a[x] - q); continue; }
            if (kind == 2) q = (b[x] + b[x + 1] + 1) >> 1;
            else if (kind == 3) q = (b[x] + b[x + stride] + 1) >> 1;
            else q = (b[x] + b[x + 1] + b[x + stride] + b[x + stride + 1] + 2) >> 2;
            d = a[x] - q;
            r += d < 0 ? -d : d;
        }
    return r;
}
