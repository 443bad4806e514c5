/*
 * ffref_shim.c — flat accessors onto the real reference; see ffref.h.  TEST INFRASTRUCTURE ONLY.
 * Includes the reference's headers where they lie (-I/root/reference); contains no reference code.
 */
#include "config.h"
#include <string.h>
#include "libavutil/cpu.h"
#include "libavutil/log.h"
#include "libavutil/mem.h"
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "libavutil/tx.h"
#include "libavutil/float_dsp.h"
#include "libswscale/swscale.h"
#include "libswscale/swscale_internal.h"
#include "libavcodec/h264dsp.h"
#include "libavcodec/h264qpel.h"
#include "libavcodec/h264chroma.h"
#include "libavcodec/me_cmp.h"
#include "libavcodec/hevc/dsp.h"
#include "libavcodec/vp9dsp.h"
#include "libavfilter/motion_estimation.h"
#include "ffref.h"

static void pure_c(void) { av_force_cpu_flags(0); av_log_set_level(AV_LOG_ERROR); }

/* ---- swscale ---- */
void *ffref_sws_create(int srcW, int srcH, int srcFmt, int dstW, int dstH, int dstFmt, int flags, int threads)
{
    pure_c();
    SwsContext *sws = sws_alloc_context();
    if (!sws)
        return NULL;
    av_opt_set_int(sws, "srcw", srcW, 0);
    av_opt_set_int(sws, "srch", srcH, 0);
    av_opt_set_int(sws, "src_format", srcFmt, 0);
    av_opt_set_int(sws, "dstw", dstW, 0);
    av_opt_set_int(sws, "dsth", dstH, 0);
    av_opt_set_int(sws, "dst_format", dstFmt, 0);
    av_opt_set_int(sws, "sws_flags", flags, 0);
    av_opt_set_int(sws, "threads", threads, 0);
    if (sws_init_context(sws, NULL, NULL) < 0) {
        sws_freeContext(sws);
        return NULL;
    }
    return sws;
}
/* the same with c->opts.src_range / dst_range given (1 = full range): what sws_setColorspaceDetails() or a J format sets */
void *ffref_sws_create_ranges(int srcW, int srcH, int srcFmt, int dstW, int dstH, int dstFmt, int flags, int threads, int src_range, int dst_range)
{
    pure_c();
    SwsContext *sws = sws_alloc_context();
    if (!sws)
        return NULL;
    av_opt_set_int(sws, "srcw", srcW, 0);
    av_opt_set_int(sws, "srch", srcH, 0);
    av_opt_set_int(sws, "src_format", srcFmt, 0);
    av_opt_set_int(sws, "dstw", dstW, 0);
    av_opt_set_int(sws, "dsth", dstH, 0);
    av_opt_set_int(sws, "dst_format", dstFmt, 0);
    av_opt_set_int(sws, "sws_flags", flags, 0);
    av_opt_set_int(sws, "threads", threads, 0);
    av_opt_set_int(sws, "src_range", src_range, 0);
    av_opt_set_int(sws, "dst_range", dst_range, 0);
    if (sws_init_context(sws, NULL, NULL) < 0) {
        sws_freeContext(sws);
        return NULL;
    }
    return sws;
}
void ffref_sws_free(void *ctx) { sws_freeContext(ctx); }
/* sws_setColorspaceDetails() on the live context: matrix row `cs` (SWS_CS_*) for the source, the default row for the target */
int ffref_sws_set_colorspace(void *ctx, int cs, int src_range, int brightness, int contrast, int saturation)
{
    return sws_setColorspaceDetails(ctx, sws_getCoefficients(cs), src_range, sws_getCoefficients(SWS_CS_DEFAULT), 0, brightness, contrast,
                                    saturation);
}
/* the four ints of sws_getCoefficients(cs) (libswscale/yuv2rgb.c:47-66): what c->srcColorspaceTable holds */
void ffref_sws_coefficients(int cs, int out[4]) { memcpy(out, sws_getCoefficients(cs), 4 * sizeof(int)); }
int ffref_sws_scale(void *ctx, const uint8_t *const src[], const int srcStride[], int y, int h,
                    uint8_t *const dst[], const int dstStride[])
{
    return sws_scale(ctx, src, srcStride, y, h, dst, dstStride);
}
static SwsInternal *inner(void *ctx)
{
    SwsInternal *c = sws_internal(ctx);
    if (c->nb_slice_ctx)            /* threaded parent: tables live in the slice contexts */
        c = sws_internal(c->slice_ctx[0]);
    return c;
}
int ffref_sws_filter(void *ctx, int which, const int16_t **filter, const int32_t **pos, int *n)
{
    SwsInternal *c = inner(ctx);
    switch (which) {
    case 0: *filter = c->hLumFilter; *pos = c->hLumFilterPos; *n = c->opts.dst_w;  return c->hLumFilterSize;
    case 1: *filter = c->hChrFilter; *pos = c->hChrFilterPos; *n = c->chrDstW;     return c->hChrFilterSize;
    case 2: *filter = c->vLumFilter; *pos = c->vLumFilterPos; *n = c->opts.dst_h;  return c->vLumFilterSize;
    case 3: *filter = c->vChrFilter; *pos = c->vChrFilterPos; *n = c->chrDstH;     return c->vChrFilterSize;
    }
    return -1;
}
int ffref_sws_is_unscaled(void *ctx) { return inner(ctx)->convert_unscaled != NULL; }
/* the flags the context ended up with (SWS_FULL_CHR_H_INT may have been forced, utils.c:1270-1290) and the six coefficients of the
 * full-chroma writers (yuv2rgb.c:786-791) */
int ffref_sws_flags(void *ctx) { return (int)((SwsContext *)ctx)->flags; }
void ffref_sws_full_coeffs(void *ctx, int out[6])
{
    SwsInternal *c = inner(ctx);
    out[0] = c->yuv2rgb_y_coeff;   out[1] = c->yuv2rgb_y_offset;  out[2] = c->yuv2rgb_v2r_coeff;
    out[3] = c->yuv2rgb_v2g_coeff; out[4] = c->yuv2rgb_u2g_coeff; out[5] = c->yuv2rgb_u2b_coeff;
}
void ffref_sws_yuv2rgb_tables(void *ctx, const uint8_t **rV, const int **gU, const int **gV, const uint8_t **bU)
{
    SwsInternal *c = inner(ctx);
    for (int i = 0; i < 256 + 2 * YUVRGB_TABLE_HEADROOM; i++) {
        rV[i] = c->table_rV[i];
        bU[i] = c->table_bU[i];
    }
    *gU = (const int *)c->table_gU;   /* caller treats as opaque; only used for pointer diffs */
    *gV = c->table_gV;
}
int ffref_pix_fmt(const char *name) { return av_get_pix_fmt(name); }
void ffref_sws_hyscale(void *ctx, int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter,
                       const int32_t *pos, int fs)
{
    SwsInternal *c = inner(ctx);
    c->hyScale(c, dst, dstW, src, filter, pos, fs);
}
void ffref_sws_yuv2planeX(void *ctx, const int16_t *filter, int fs, const int16_t **src, uint8_t *dest,
                          int dstW, const uint8_t *dither, int offset)
{
    inner(ctx)->yuv2planeX(filter, fs, src, dest, dstW, dither, offset);
}
void ffref_sws_yuv2plane1(void *ctx, const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset)
{
    inner(ctx)->yuv2plane1(src, dest, dstW, dither, offset);
}
void ffref_sws_yuv2nv12cX(void *ctx, int dstFormat, const uint8_t *chrDither, const int16_t *chrFilter, int fs,
                          const int16_t **chrU, const int16_t **chrV, uint8_t *dest, int dstW)
{
    inner(ctx)->yuv2nv12cX(dstFormat, chrDither, chrFilter, fs, chrU, chrV, dest, dstW);
}

/* the packed-output members of a context whose target is packed RGB (yuv2rgb_{1,2,X}_c_template via the YUV2RGBWRAPPER macros) */
void ffref_sws_yuv2packedX(void *ctx, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize, const int16_t *chrFilter,
                           const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize, uint8_t *dest, int dstW, int y)
{
    SwsInternal *c = inner(ctx);
    c->yuv2packedX(c, lumFilter, lumSrc, lumFilterSize, chrFilter, chrUSrc, chrVSrc, chrFilterSize, NULL, dest, dstW, y);
}
void ffref_sws_yuv2packed2(void *ctx, const int16_t *lumSrc[2], const int16_t *chrUSrc[2], const int16_t *chrVSrc[2], uint8_t *dest, int dstW,
                           int yalpha, int uvalpha, int y)
{
    SwsInternal *c = inner(ctx);
    c->yuv2packed2(c, lumSrc, chrUSrc, chrVSrc, NULL, dest, dstW, yalpha, uvalpha, y);
}
void ffref_sws_yuv2packed1(void *ctx, const int16_t *lumSrc, const int16_t *chrUSrc[2], const int16_t *chrVSrc[2], uint8_t *dest, int dstW,
                           int uvalpha, int y)
{
    SwsInternal *c = inner(ctx);
    c->yuv2packed1(c, lumSrc, chrUSrc, chrVSrc, NULL, dest, dstW, uvalpha, y);
}

/* ---- h264dsp / qpel / me_cmp ---- */
static H264DSPContext   h264;
static H264DSPContext   h264_422;
static H264QpelContext  qpel;
static MECmpContext     mecmp;
static H264ChromaContext chroma;
static HEVCDSPContext hevc;
static int dsp_ready;
static void dsp_init(void)
{
    if (dsp_ready)
        return;
    pure_c();
    ff_h264dsp_init(&h264, 8, 1);
    ff_h264dsp_init(&h264_422, 8, 2);
    ff_h264qpel_init(&qpel, 8);
    ff_me_cmp_init(&mecmp, NULL);
    ff_h264chroma_init(&chroma, 8);
    ff_hevc_dsp_init(&hevc, 8);
    dsp_ready = 1;
}
/* the depth the ffref_hevc_* calls below run at: HEVCDSPContext is re-initialised the way the decoder does per SPS
 * (ff_hevc_dsp_init(&s->hevcdsp, sps->bit_depth), hevc/hevcdec.c); pixels are uint16_t above 8 bits */
void ffref_hevc_set_bit_depth(int bit_depth)
{
    dsp_init();
    ff_hevc_dsp_init(&hevc, bit_depth);
}
void ffref_h264_idct(int which, uint8_t *dst, int16_t *block, ptrdiff_t stride)
{
    dsp_init();
    switch (which) {
    case 0: h264.idct_add(dst, block, stride); break;
    case 1: h264.idct8_add(dst, block, stride); break;
    case 2: h264.idct_dc_add(dst, block, stride); break;
    case 3: h264.idct8_dc_add(dst, block, stride); break;
    }
}
/* ---- CPU-baseline runners (bench.py): the reference's own pointers over a batch, split statically over pthreads ---- */
#include <pthread.h>
typedef struct { int which; uint8_t *dst; ptrdiff_t stride; const int32_t *off; int16_t *blk; int lo, hi; } IdctJob;
static void *idct_worker(void *p)
{
    IdctJob *j = p;
    const int step = j->which == 1 || j->which == 3 ? 64 : 16;
    for (int i = j->lo; i < j->hi; i++) {
        uint8_t *d = j->dst + j->off[i];
        int16_t *b = j->blk + (size_t)i * step;
        switch (j->which) {
        case 0: h264.idct_add(d, b, j->stride); break;
        case 1: h264.idct8_add(d, b, j->stride); break;
        case 2: h264.idct_dc_add(d, b, j->stride); break;
        default: h264.idct8_dc_add(d, b, j->stride); break;
        }
    }
    return NULL;
}
/* which as ffref_h264_idct(); block i is added at dst + off[i]; blocks are disjoint, so any split is the serial result */
int ffref_h264_idct_batch(int which, uint8_t *dst, ptrdiff_t stride, const int32_t *off, int16_t *blk, int n, int threads)
{
    dsp_init();
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_t th[1024];
    IdctJob job[1024];
    const int per = (n + threads - 1) / threads;
    int started = 0;
    for (int t = 0; t < threads; t++) {
        int lo = t * per, hi = lo + per > n ? n : lo + per;
        if (lo >= hi) break;
        job[t] = (IdctJob){ which, dst, stride, off, blk, lo, hi };
        if (threads == 1) { idct_worker(&job[t]); return 0; }
        if (pthread_create(&th[t], NULL, idct_worker, &job[t])) break;
        started++;
    }
    for (int t = 0; t < started; t++)
        pthread_join(th[t], NULL);
    return started;
}

/* The same split with PERSISTENT threads and a clock inside: one untimed warm-up pass (page faults, caches, thread start), then
 * whole passes until `min_seconds` have gone by.  Every pass runs the reference's function over every block again (after the
 * first pass the coefficient blocks are the zeros the function itself leaves behind: same instructions, same stores).
 * *seconds = wall time of the timed passes, *passes = how many; returns the number of worker threads. */
#include <time.h>
typedef struct { IdctJob job; pthread_barrier_t *bar; volatile int *stop; double t_end; long passes; double t_done; } IdctLoop;
static double now_s(void);
static void *idct_loop_worker(void *p)
{
    IdctLoop *l = p;
    /* NUMA: the caller's arrays were first touched by ONE thread (one memory node); a worker of a 2-socket box would stream its
     * share over the socket link.  Each worker therefore times its share on thread-local copies it has first-touched itself —
     * its coefficient blocks and the picture rows they land in — which is the placement a frame-threaded decoder gets. */
    IdctJob *j = &l->job;
    const int step = j->which == 1 || j->which == 3 ? 64 : 16, side = j->which == 1 || j->which == 3 ? 8 : 4;
    const size_t nblk = (size_t)(j->hi - j->lo) * step;
    int32_t omin = INT32_MAX, omax = 0;
    for (int i = j->lo; i < j->hi; i++) {
        if (j->off[i] < omin) omin = j->off[i];
        if (j->off[i] > omax) omax = j->off[i];
    }
    const size_t npic = (size_t)(omax - omin) + (size_t)side * j->stride;
    int16_t *lb = malloc(nblk * sizeof(int16_t));
    uint8_t *lp = malloc(npic);
    int32_t *lo = malloc((size_t)(j->hi - j->lo) * sizeof(int32_t));
    if (lb && lp && lo) {
        memcpy(lb, j->blk + (size_t)j->lo * step, nblk * sizeof(int16_t));
        memcpy(lp, j->dst + omin, npic);
        for (int i = j->lo; i < j->hi; i++)
            lo[i - j->lo] = j->off[i] - omin;
        j->blk = lb - (size_t)j->lo * step;
        j->dst = lp;
        j->off = lo - j->lo;
    }
    /* one barrier-bracketed warm-up pass, then free-running passes until the deadline: a barrier per pass costs a 256-thread box
     * milliseconds of futex traffic per round, an order more than the 1.2 ms a thread's share of one pass takes */
    pthread_barrier_wait(l->bar);
    idct_worker(&l->job);
    pthread_barrier_wait(l->bar);
    pthread_barrier_wait(l->bar); /* the caller has set t_end */
    do {
        idct_worker(&l->job);
        l->passes++;
        l->t_done = now_s();
    } while (l->t_done < l->t_end);
    pthread_barrier_wait(l->bar);
    free(lb);
    free(lp);
    free(lo);
    return NULL;
}
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}
int ffref_h264_idct_batch_timed(int which, uint8_t *dst, ptrdiff_t stride, const int32_t *off, int16_t *blk, int n, int threads,
                                double min_seconds, double *seconds, int *passes)
{
    dsp_init();
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    static pthread_t th[1024];
    static IdctLoop loop[1024];
    const int per = (n + threads - 1) / threads;
    int want = 0;
    for (int t = 0; t < threads; t++)
        if (t * per < n)
            want++;
    pthread_barrier_t bar;
    volatile int stop = 0;
    (void)stop;
    if (pthread_barrier_init(&bar, NULL, want + 1))
        return -1;
    int started = 0;
    for (int t = 0; t < want; t++) {
        int lo = t * per, hi = lo + per > n ? n : lo + per;
        loop[t] = (IdctLoop){ { which, dst, stride, off, blk, lo, hi }, &bar, &stop };
        if (pthread_create(&th[t], NULL, idct_loop_worker, &loop[t]))
            break;
        started++;
    }
    if (started != want) { /* cannot release a barrier sized for more threads: give up loudly */
        stop = 1;
        for (int t = 0; t < started; t++)
            pthread_cancel(th[t]);
        return -1;
    }
    pthread_barrier_wait(&bar); /* warm-up pass */
    pthread_barrier_wait(&bar);
    const double t0 = now_s();
    for (int t = 0; t < started; t++) {
        loop[t].t_end = t0 + min_seconds;
        loop[t].passes = 0;
    }
    pthread_barrier_wait(&bar);
    pthread_barrier_wait(&bar);
    /* threads finish their last pass at different times: the rate is blocks done / the time the slowest needed */
    double t1 = t0, blocks = 0;
    for (int t = 0; t < started; t++) {
        if (loop[t].t_done > t1)
            t1 = loop[t].t_done;
        blocks += (double)loop[t].passes * (loop[t].job.hi - loop[t].job.lo);
    }
    const int np = (int)(blocks / n + 0.5);
    for (int t = 0; t < started; t++)
        pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    *seconds = (t1 - t0) * ((double)np * n / (blocks > 0 ? blocks : 1)); /* the time np whole passes take at the measured rate */
    *passes = np;
    return started;
}

typedef struct { void *ctx; const uint8_t *const *src; const int *ss; uint8_t *const *dst; const int *ds; int h, reps; } SwsJob;
static void *sws_worker(void *p)
{
    SwsJob *j = p;
    for (int r = 0; r < j->reps; r++)
        sws_scale(j->ctx, j->src, j->ss, 0, j->h, j->dst, j->ds);
    return NULL;
}
/* frame-parallel scaling: `threads` single-threaded contexts, each converting its own frame `reps` times (the way a
 * batch of independent frames uses all host cores; ffref_sws_create(..., threads) is the slice-threaded alternative) */
int ffref_sws_scale_frames_mt(void *const *ctxs, const uint8_t *const *const *srcs, const int *ss, uint8_t *const *const *dsts,
                              const int *ds, int srcH, int threads, int reps)
{
    pthread_t th[1024];
    SwsJob job[1024];
    if (threads > 1024) threads = 1024;
    int started = 0;
    for (int t = 0; t < threads; t++) {
        job[t] = (SwsJob){ ctxs[t], srcs[t], ss, dsts[t], ds, srcH, reps };
        if (pthread_create(&th[t], NULL, sws_worker, &job[t])) break;
        started++;
    }
    for (int t = 0; t < started; t++)
        pthread_join(th[t], NULL);
    return started;
}

void ffref_h264_idct_multi(int which, uint8_t *dst, const int *blockoffset, int16_t *block, ptrdiff_t stride,
                           const uint8_t *nnzc)
{
    dsp_init();
    switch (which) {
    case 0: h264.idct_add16(dst, blockoffset, block, stride, nnzc); break;
    case 1: h264.idct8_add4(dst, blockoffset, block, stride, nnzc); break;
    case 2: h264.idct_add16intra(dst, blockoffset, block, stride, nnzc); break;
    }
}
void ffref_h264_idct_add8(uint8_t **dst, const int *blockoffset, int16_t *block, ptrdiff_t stride,
                          const uint8_t *nnzc, int chroma_format_idc)
{
    dsp_init();
    (chroma_format_idc == 2 ? &h264_422 : &h264)->idct_add8(dst, blockoffset, block, stride, nnzc);
}
void ffref_h264_luma_dc_dequant_idct(int16_t *output, int16_t *input, int qmul) { dsp_init(); h264.luma_dc_dequant_idct(output, input, qmul); }
void ffref_h264_chroma_dc_dequant_idct(int16_t *block, int qmul) { dsp_init(); h264.chroma_dc_dequant_idct(block, qmul); }
void ffref_h264_add_pixels_clear(int n, uint8_t *dst, int16_t *block, ptrdiff_t stride)
{
    dsp_init();
    if (n == 8) h264.add_pixels8_clear(dst, block, stride); else h264.add_pixels4_clear(dst, block, stride);
}
void ffref_h264_loop_filter(int which, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0)
{
    dsp_init();
    switch (which) {
    case 0: h264.v_loop_filter_luma(pix, stride, alpha, beta, tc0); break;
    case 1: h264.h_loop_filter_luma(pix, stride, alpha, beta, tc0); break;
    case 2: h264.v_loop_filter_chroma(pix, stride, alpha, beta, tc0); break;
    case 3: h264.h_loop_filter_chroma(pix, stride, alpha, beta, tc0); break;
    case 4: h264.v_loop_filter_luma_intra(pix, stride, alpha, beta); break;
    case 5: h264.h_loop_filter_luma_intra(pix, stride, alpha, beta); break;
    case 6: h264.v_loop_filter_chroma_intra(pix, stride, alpha, beta); break;
    case 7: h264.h_loop_filter_chroma_intra(pix, stride, alpha, beta); break;
    }
}
void ffref_h264_qpel(int avg, int size_idx, int mcxy, uint8_t *dst, const uint8_t *src, ptrdiff_t stride)
{
    dsp_init();
    (avg ? qpel.avg_h264_qpel_pixels_tab : qpel.put_h264_qpel_pixels_tab)[size_idx][mcxy](dst, src, stride);
}
void ffref_h264_chroma(int avg, int idx, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h, int x, int y)
{
    dsp_init();
    (avg ? chroma.avg_h264_chroma_pixels_tab : chroma.put_h264_chroma_pixels_tab)[idx](dst, src, stride, h, x, y);
}
void ffref_h264_weight(int idx, uint8_t *block, ptrdiff_t stride, int height, int log2_denom, int weight, int offset)
{
    dsp_init();
    h264.weight_pixels_tab[idx](block, stride, height, log2_denom, weight, offset);
}
void ffref_h264_biweight(int idx, uint8_t *dst, uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd,
                         int weights, int offset)
{
    dsp_init();
    h264.biweight_pixels_tab[idx](dst, src, stride, height, log2_denom, weightd, weights, offset);
}
/* the depth the ffref_h264_* calls run at: the three H.264 tables are re-initialised the way the decoder does per SPS
 * (ff_h264dsp_init(&h->h264dsp, sps->bit_depth_luma, sps->chroma_format_idc) & co, libavcodec/h264_slice.c); pixels are uint16_t and
 * coefficients int32_t above 8 bits.  bit_depth 8 / 9 / 10 / 12 / 14 are the depths the reference instantiates (h264dsp.c:135-147) */
void ffref_h264_set_bit_depth(int bit_depth)
{
    dsp_init();
    ff_h264dsp_init(&h264, bit_depth, 1);
    ff_h264dsp_init(&h264_422, bit_depth, 2);
    ff_h264qpel_init(&qpel, bit_depth);
    ff_h264chroma_init(&chroma, bit_depth);
}
/* the members ffref_h264_loop_filter() does not reach: kind as there (bit 0 h_, bit 1 chroma, bit 2 intra), variant 1 = MBAFF
 * (h_ only), 2 = 4:2:2 (h_ chroma only), 3 = 4:2:2 MBAFF */
void ffref_h264_loop_filter_variant(int kind, int variant, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0)
{
    dsp_init();
    const H264DSPContext *c = variant >= 2 ? &h264_422 : &h264;
    const int mbaff = variant & 1;
    switch (kind) {
    case 1: (mbaff ? c->h_loop_filter_luma_mbaff : c->h_loop_filter_luma)(pix, stride, alpha, beta, tc0); break;
    case 3: (mbaff ? c->h_loop_filter_chroma_mbaff : c->h_loop_filter_chroma)(pix, stride, alpha, beta, tc0); break;
    case 5: (mbaff ? c->h_loop_filter_luma_mbaff_intra : c->h_loop_filter_luma_intra)(pix, stride, alpha, beta); break;
    case 7: (mbaff ? c->h_loop_filter_chroma_mbaff_intra : c->h_loop_filter_chroma_intra)(pix, stride, alpha, beta); break;
    default: ffref_h264_loop_filter(kind, pix, stride, alpha, beta, tc0); break;
    }
}
void ffref_h264_chroma_dc_dequant_idct_422(int16_t *block, int qmul) { dsp_init(); h264_422.chroma_dc_dequant_idct(block, qmul); }
void ffref_fdsp(int op, float *dst, const float *src0, const float *src1, const float *src2, float mul, int len)
{
    static AVFloatDSPContext *f;
    pure_c();
    if (!f)
        f = avpriv_float_dsp_alloc(0);
    switch (op) {
    case 0: f->vector_fmul(dst, src0, src1, len); break;
    case 1: f->vector_fmac_scalar(dst, src0, mul, len); break;
    case 2: f->vector_fmul_scalar(dst, src0, mul, len); break;
    case 3: f->vector_fmul_window(dst, src0, src1, src2, len); break;
    case 4: f->vector_fmul_add(dst, src0, src1, src2, len); break;
    case 5: f->vector_fmul_reverse(dst, src0, src1, len); break;
    case 6: f->butterflies_float(dst, (float *)src0, len); break;
    }
}
void ffref_hevc_idct(int idx, int16_t *coeffs, int col_limit)
{
    dsp_init();
    hevc.idct[idx](coeffs, col_limit);
}
void ffref_hevc_idct_dc(int idx, int16_t *coeffs)
{
    dsp_init();
    hevc.idct_dc[idx](coeffs);
}
void ffref_hevc_transform_4x4_luma(int16_t *coeffs)
{
    dsp_init();
    hevc.transform_4x4_luma(coeffs);
}
void ffref_hevc_add_residual(int idx, uint8_t *dst, const int16_t *res, ptrdiff_t stride)
{
    dsp_init();
    hevc.add_residual[idx](dst, res, stride);
}
void ffref_hevc_mc(int chroma, int uni, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, int mx, int my,
                   int width)
{
    static const int wtab[10] = { 2, 4, 6, 8, 12, 16, 24, 32, 48, 64 }; /* ff_hevc_pel_weight, hevc/hevcdec.c */
    int idx = 0;
    dsp_init();
    while (idx < 9 && wtab[idx] < width)
        idx++;
    if (uni)
        (chroma ? hevc.put_hevc_epel_uni : hevc.put_hevc_qpel_uni)[idx][!!my][!!mx](dst, dststride, src, srcstride, height, mx, my, width);
    else
        (chroma ? hevc.put_hevc_epel : hevc.put_hevc_qpel)[idx][!!my][!!mx](dst, src, srcstride, height, mx, my, width);
}
void ffref_hevc_mc_w(int chroma, int mode, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                     int height, int denom, int wx0, int wx1, int ox, int mx, int my, int width)
{
    static const int wtab[10] = { 2, 4, 6, 8, 12, 16, 24, 32, 48, 64 };
    int idx = 0;
    dsp_init();
    while (idx < 9 && wtab[idx] < width)
        idx++;
    if (mode == 2)
        (chroma ? hevc.put_hevc_epel_uni_w : hevc.put_hevc_qpel_uni_w)[idx][!!my][!!mx](dst, dststride, src, srcstride, height, denom, wx0, ox,
                                                                                        mx, my, width);
    else if (mode == 3)
        (chroma ? hevc.put_hevc_epel_bi : hevc.put_hevc_qpel_bi)[idx][!!my][!!mx](dst, dststride, src, srcstride, src2, height, mx, my, width);
    else
        (chroma ? hevc.put_hevc_epel_bi_w : hevc.put_hevc_qpel_bi_w)[idx][!!my][!!mx](dst, dststride, src, srcstride, src2, height, denom, wx0,
                                                                                      wx1, ox, mx, my, width);
}
void ffref_hevc_sao_band(int idx, uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *offset_val,
                         int left_class, int width, int height)
{
    dsp_init();
    hevc.sao_band_filter[idx](dst, src, stride_dst, stride_src, offset_val, left_class, width, height);
}
void ffref_hevc_sao_edge(int idx, uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, const int16_t *offset_val, int eo, int width,
                         int height)
{
    dsp_init();
    hevc.sao_edge_filter[idx](dst, src, stride_dst, offset_val, eo, width, height);
}
/* ---- vp9dsp (ff_vp9dsp_init(dsp, bpp, bitexact)): one context per depth, the ffref_vp9_* calls run at the selected one; above
 * 8 bits pixels are uint16_t and itxfm_add's block points at int32 coefficients ---- */
static int vp9_bd = 8;
static VP9DSPContext *vp9_ctx(void)
{
    static VP9DSPContext ctx[3];
    static int ready[3];
    const int i = vp9_bd == 8 ? 0 : vp9_bd == 10 ? 1 : 2;
    pure_c();
    if (!ready[i]) {
        ff_vp9dsp_init(&ctx[i], vp9_bd, 1);
        ready[i] = 1;
    }
    return &ctx[i];
}
void ffref_vp9_set_bit_depth(int bit_depth) { vp9_bd = bit_depth; }
void ffref_vp9_itxfm_add(int tx, int txtp, uint8_t *dst, ptrdiff_t stride, int16_t *block, int eob)
{
    VP9DSPContext *const v9 = vp9_ctx();
    v9->itxfm_add[tx][txtp](dst, stride, block, eob);
}
void ffref_vp9_mc(int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width, int height,
                  int mx, int my)
{
    int idx = 0;
    VP9DSPContext *const v9 = vp9_ctx();
    while ((64 >> idx) > width)
        idx++;
    v9->mc[idx][filter][avg][!!mx][!!my](dst, dststride, src, srcstride, height, mx, my);
}
/* which 0: loop_filter_8[a][dir], 1: loop_filter_16[dir], 2: loop_filter_mix2[a][b][dir] */
void ffref_vp9_loop_filter(int which, int a, int b, int dir, uint8_t *dst, ptrdiff_t stride, int E, int I, int H)
{
    VP9DSPContext *const v9 = vp9_ctx();
    if (which == 0)
        v9->loop_filter_8[a][dir](dst, stride, E, I, H);
    else if (which == 1)
        v9->loop_filter_16[dir](dst, stride, E, I, H);
    else
        v9->loop_filter_mix2[a][b][dir](dst, stride, E, I, H);
}
void ffref_vp9_intra_pred(int tx, int mode, uint8_t *dst, ptrdiff_t stride, const uint8_t *left, const uint8_t *top)
{
    VP9DSPContext *const v9 = vp9_ctx();
    v9->intra_pred[tx][mode](dst, stride, left, top);
}
void ffref_vp9_smc(int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width, int height,
                   int mx, int my, int dx, int dy)
{
    int idx = 0;
    VP9DSPContext *const v9 = vp9_ctx();
    while ((64 >> idx) > width)
        idx++;
    v9->smc[idx][filter][avg](dst, dststride, src, srcstride, height, mx, my, dx, dy);
}
void ffref_hevc_dequant(int16_t *coeffs, int log2_size)
{
    dsp_init();
    hevc.dequant(coeffs, log2_size);
}
void ffref_hevc_transform_rdpcm(int16_t *coeffs, int log2_size, int mode)
{
    dsp_init();
    hevc.transform_rdpcm(coeffs, log2_size, mode);
}
void ffref_hevc_sao_edge_restore(int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t sd, ptrdiff_t ss, int eo, int offset0,
                                 const int *borders, int width, int height, const uint8_t *vert_edge, const uint8_t *horiz_edge,
                                 const uint8_t *diag_edge)
{
    SAOParams sao;
    dsp_init();
    memset(&sao, 0, sizeof(sao));
    sao.eo_class[0] = eo;
    sao.offset_val[0][0] = offset0;
    hevc.sao_edge_restore[variant](dst, src, sd, ss, &sao, borders, width, height, 0, vert_edge, horiz_edge, diag_edge);
}
void ffref_hevc_loop_filter(int which, uint8_t *pix, ptrdiff_t stride, int beta, const int32_t *tc, const uint8_t *no_p, const uint8_t *no_q)
{
    dsp_init();
    switch (which) {
    case 0: hevc.hevc_h_loop_filter_luma(pix, stride, beta, tc, no_p, no_q); break;
    case 1: hevc.hevc_v_loop_filter_luma(pix, stride, beta, tc, no_p, no_q); break;
    case 2: hevc.hevc_h_loop_filter_chroma(pix, stride, tc, no_p, no_q); break;
    case 3: hevc.hevc_v_loop_filter_chroma(pix, stride, tc, no_p, no_q); break;
    case 4: hevc.hevc_h_loop_filter_luma_c(pix, stride, beta, tc, no_p, no_q); break;
    case 5: hevc.hevc_v_loop_filter_luma_c(pix, stride, beta, tc, no_p, no_q); break;
    case 6: hevc.hevc_h_loop_filter_chroma_c(pix, stride, tc, no_p, no_q); break;
    case 7: hevc.hevc_v_loop_filter_chroma_c(pix, stride, tc, no_p, no_q); break;
    }
}
int ffref_me_cmp(int kind, int idx, const uint8_t *blk1, const uint8_t *blk2, ptrdiff_t stride, int h)
{
    dsp_init();
    switch (kind) {
    case 0: return mecmp.sad[idx](NULL, blk1, blk2, stride, h);
    case 1: return mecmp.hadamard8_diff[idx](NULL, blk1, blk2, stride, h);
    case 2: return mecmp.sse[idx](NULL, blk1, blk2, stride, h);
    case 3: return mecmp.pix_abs[idx][1](NULL, blk1, blk2, stride, h); /* _x2 */
    case 4: return mecmp.pix_abs[idx][2](NULL, blk1, blk2, stride, h); /* _y2 */
    case 5: return mecmp.pix_abs[idx][3](NULL, blk1, blk2, stride, h); /* _xy2 */
    case 6: return mecmp.nsse[idx](NULL, blk1, blk2, stride, h);       /* no encoder context: weight 8 */
    }
    return -1;
}
uint64_t ffref_me_search_esa(const uint8_t *cur, const uint8_t *ref, int linesize, int width, int height,
                             int mb_size, int search_param, int x_mb, int y_mb, int *mv)
{
    /* context set-up as vf_mestimate.c config_input does it */
    AVMotionEstContext me;
    int log2 = 0;
    while ((1 << log2) < mb_size) log2++;
    int b_w = width >> log2, b_h = height >> log2;
    memset(&me, 0, sizeof(me));
    ff_me_init_context(&me, mb_size, search_param, width, height, 0, (b_w - 1) << log2, 0, (b_h - 1) << log2);
    me.data_cur = (uint8_t *)cur;
    me.data_ref = (uint8_t *)ref;
    me.linesize = linesize;
    mv[0] = x_mb;
    mv[1] = y_mb;
    return ff_me_search_esa(&me, x_mb, y_mb, mv);
}

/* ---- av_tx ---- */
typedef struct { AVTXContext *s; av_tx_fn fn; } RefTx;
void *ffref_tx_create(int type, int inv, int len, float scale, uint64_t flags)
{
    pure_c();
    RefTx *t = av_mallocz(sizeof(*t));
    if (!t)
        return NULL;
    if (av_tx_init(&t->s, &t->fn, type, inv, len, &scale, flags) < 0) {
        av_free(t);
        return NULL;
    }
    return t;
}
/* the double types read *scale as a double (SCALE_TYPE, libavutil/tx_double.c) */
void *ffref_tx_create_d(int type, int inv, int len, double scale, uint64_t flags)
{
    pure_c();
    RefTx *t = av_mallocz(sizeof(*t));
    if (!t)
        return NULL;
    if (av_tx_init(&t->s, &t->fn, type, inv, len, &scale, flags) < 0) {
        av_free(t);
        return NULL;
    }
    return t;
}
void ffref_tx_run(void *ctx, void *out, void *in, ptrdiff_t stride)
{
    RefTx *t = ctx;
    t->fn(t->s, out, in, stride);
}
void ffref_tx_free(void *ctx)
{
    RefTx *t = ctx;
    if (t) {
        av_tx_uninit(&t->s);
        av_free(t);
    }
}


/* ---- sws_scale_frame() on refcounted frames: the entry through which a context made with threads > 1 actually threads ---------------
 * (sws_scale() on such a context runs slice_ctx[0] alone, libswscale/swscale.c:1626-1643; sws_scale_frame -> sws_frame_start /
 * sws_send_slice / sws_receive_slice reaches ff_sws_slice_worker on every slice thread, :1405-1420, :1645-1679).  bench.py's
 * slice-threaded CPU leg and tests/test_oracle_vs_ref.py use it. */
void *ffref_frame_alloc(int w, int h, int fmt)
{
    AVFrame *f = av_frame_alloc();
    if (!f)
        return NULL;
    f->width = w;
    f->height = h;
    f->format = fmt;
    if (av_frame_get_buffer(f, 64) < 0) {
        av_frame_free(&f);
        return NULL;
    }
    return f;
}
void ffref_frame_free(void *f) { AVFrame *p = f; av_frame_free(&p); }
uint8_t *ffref_frame_plane(void *f, int i, int *linesize)
{
    AVFrame *p = f;
    *linesize = p->linesize[i];
    return p->data[i];
}
int ffref_sws_scale_frame(void *ctx, void *dst, void *src) { return sws_scale_frame(ctx, dst, src); }

/* ---- CPU-baseline legs for BASELINE's other configurations (bench.py cpu_baseline; TEST / MEASUREMENT INFRASTRUCTURE) ---------------
 * One runner for all of them: persistent threads, every thread builds its OWN state (buffers first-touched on its own node, its own
 * contexts), runs one untimed warm-up chunk, then free-running chunks until the deadline.  A chunk is a fixed amount of the reference's
 * own work through the reference's own function pointers; rate = units done / the time the slowest thread needed.
 *   leg 0  unscaled yuv420p -> rgb24 3840x2160 (sws_scale -> yuv2rgb_c_24_rgb, libswscale/yuv2rgb.c:530): chunk = 1 frame, unit = pixel
 *   leg 1  put_h264_qpel_pixels_tab[0][mc] on every 16x16 macroblock of a 4K luma plane, mc and the +-24 displacement drawn per block
 *          (libavcodec/h264qpel_template.c): chunk = one macroblock row (240 blocks), unit = pixel
 *   leg 2 / 3  h264 v_ / h_loop_filter_luma on one edge per 16x16 tile of a 4K plane, alpha 40, beta 12, tc0 2 (h264dsp_template.c:104-163):
 *          chunk = one tile row (240 edges), unit = edge
 *   leg 4 / 5  av_tx AV_TX_FLOAT_MDCT len 1024 forward / inverse (libavutil/tx_template.c): chunk = 64 transforms, unit = transform
 *   leg 6  ff_me_search_esa, 16x16, search_param 7, SAD (libavfilter/motion_estimation.c:78-95): chunk = one macroblock row of a 4K pair,
 *          unit = macroblock search
 *   leg 7  the same window walked with MECmpContext.hadamard8_diff[0] as the cost (libavcodec/me_cmp.c:514-562,933-950)
 */
typedef struct BenchState {
    int leg, thread;
    void *ctx;
    uint8_t *a, *b;
    float *fa, *fb;
    int32_t *mc;
    long cursor;
} BenchState;
typedef struct BenchLoop { BenchState st; pthread_barrier_t *bar; double t_end, t_done; long chunks; int ok; } BenchLoop;

static uint32_t bench_rand(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static void bench_fill(uint8_t *p, size_t n, uint32_t seed, int lo, int span)
{
    for (size_t i = 0; i < n; i++)
        p[i] = (uint8_t)(lo + bench_rand(&seed) % (uint32_t)span);
}
enum { BW = 3840, BH = 2160, BPAD = 32, BSTRIDE = BW + 2 * BPAD };

static long bench_units(int leg)
{
    switch (leg) {
    case 0: return (long)BW * BH;
    case 1: return 240L * 256;
    case 2: case 3: return 240;
    case 4: case 5: return 64;
    default: return 240;
    }
}
static int bench_make(BenchState *s)
{
    const size_t plane = (size_t)BSTRIDE * (BH + 2 * BPAD);
    uint32_t seed = 0x9e3779b9u * (uint32_t)(s->thread + 1) + (uint32_t)s->leg;
    switch (s->leg) {
    case 0:
        s->ctx = sws_getContext(BW, BH, AV_PIX_FMT_YUV420P, BW, BH, AV_PIX_FMT_RGB24, SWS_BICUBIC, NULL, NULL, NULL);
        s->a = av_malloc((size_t)BW * BH * 3 / 2);
        s->b = av_malloc((size_t)BW * BH * 3);
        if (!s->ctx || !s->a || !s->b) return 0;
        bench_fill(s->a, (size_t)BW * BH * 3 / 2, seed, 0, 256);
        memset(s->b, 0, (size_t)BW * BH * 3);
        return 1;
    case 1:
        s->a = av_malloc(plane); s->b = av_malloc(plane);
        s->mc = av_malloc(sizeof(int32_t) * 2 * 240 * 135);
        if (!s->a || !s->b || !s->mc) return 0;
        bench_fill(s->a, plane, seed, 0, 256);
        memset(s->b, 0, plane);
        for (int i = 0; i < 240 * 135; i++) {
            const int dy = (int)(bench_rand(&seed) % 49) - 24, dx = (int)(bench_rand(&seed) % 49) - 24;
            s->mc[2 * i] = dy * BSTRIDE + dx;
            s->mc[2 * i + 1] = (int)(bench_rand(&seed) & 15);
        }
        return 1;
    case 2: case 3:
        s->a = av_malloc((size_t)BW * BH);
        if (!s->a) return 0;
        bench_fill(s->a, (size_t)BW * BH, seed, 96, 64);
        return 1;
    case 4: case 5: {
        RefTx *t = ffref_tx_create(AV_TX_FLOAT_MDCT, s->leg == 5, 1024, 1.0f, 0);
        s->ctx = t;
        s->fa = av_malloc(sizeof(float) * 64 * 2048);
        s->fb = av_malloc(sizeof(float) * 64 * 2048);
        if (!t || !s->fa || !s->fb) return 0;
        for (int i = 0; i < 64 * 2048; i++)
            s->fa[i] = (float)(bench_rand(&seed) & 0xffff) * (1.0f / 65536.0f) - 0.5f;
        return 1;
    }
    default:
        s->a = av_malloc((size_t)BW * BH); s->b = av_malloc((size_t)BW * BH);
        if (!s->a || !s->b) return 0;
        bench_fill(s->a, (size_t)BW * BH, seed, 0, 256);
        /* the reference frame = the current one moved by (3, -2), as the GPU leg's (torch.roll) */
        for (int y = 0; y < BH; y++)
            for (int x = 0; x < BW; x++)
                s->b[(size_t)y * BW + x] = s->a[(size_t)((y - 3 + BH) % BH) * BW + (x + 2) % BW];
        return 1;
    }
}
static void bench_drop(BenchState *s)
{
    if (s->leg == 0 && s->ctx) sws_freeContext(s->ctx);
    if ((s->leg == 4 || s->leg == 5) && s->ctx) ffref_tx_free(s->ctx);
    av_free(s->a); av_free(s->b); av_free(s->fa); av_free(s->fb); av_free(s->mc);
}
static void bench_chunk(BenchState *s)
{
    switch (s->leg) {
    case 0: {
        const uint8_t *src[4] = { s->a, s->a + (size_t)BW * BH, s->a + (size_t)BW * BH * 5 / 4, NULL };
        const int ss[4] = { BW, BW / 2, BW / 2, 0 };
        uint8_t *dst[4] = { s->b, NULL, NULL, NULL };
        const int ds[4] = { 3 * BW, 0, 0, 0 };
        sws_scale(s->ctx, src, ss, 0, BH, dst, ds);
        break;
    }
    case 1: {
        const int row = (int)(s->cursor++ % 135);
        for (int x = 0; x < 240; x++) {
            const size_t o = (size_t)(BPAD + 16 * row) * BSTRIDE + BPAD + 16 * x;
            const int32_t *m = s->mc + 2 * (row * 240 + x);
            qpel.put_h264_qpel_pixels_tab[0][m[1]](s->b + o, s->a + o + m[0], BSTRIDE);
        }
        break;
    }
    case 2: case 3: {
        const int row = (int)(s->cursor++ % 135);
        int8_t tc0[4] = { 2, 2, 2, 2 };
        for (int x = 0; x < 240; x++) {
            uint8_t *p = s->a + (size_t)16 * row * BW + 16 * x;
            if (s->leg == 2) h264.v_loop_filter_luma(p + 8 * BW, BW, 40, 12, tc0);
            else             h264.h_loop_filter_luma(p + 8, BW, 40, 12, tc0);
        }
        break;
    }
    case 4: case 5: {
        RefTx *t = s->ctx;
        const int nin = s->leg == 4 ? 2048 : 1024;
        for (int i = 0; i < 64; i++)
            t->fn(t->s, s->fb + (size_t)i * 1024, s->fa + (size_t)i * nin, sizeof(float));
        break;
    }
    default: {
        const int row = (int)(s->cursor++ % 135);
        for (int x = 0; x < 240; x++) {
            int mv[2];
            if (s->leg == 6) {
                ffref_me_search_esa(s->a, s->b, BW, BW, BH, 16, 7, 16 * x, 16 * row, mv);
            } else {
                /* ff_me_search_esa's window and order (motion_estimation.c:78-95) with the Hadamard cost */
                const int x_mb = 16 * x, y_mb = 16 * row, lim_x = BW - 16, lim_y = BH - 16;
                const int x0 = x_mb - 7 > 0 ? x_mb - 7 : 0, y0 = y_mb - 7 > 0 ? y_mb - 7 : 0;
                const int x1 = x_mb + 7 < lim_x ? x_mb + 7 : lim_x, y1 = y_mb + 7 < lim_y ? y_mb + 7 : lim_y;
                const uint8_t *c = s->a + (size_t)y_mb * BW + x_mb;
                int best = mecmp.hadamard8_diff[0](NULL, c, s->b + (size_t)y_mb * BW + x_mb, BW, 16);
                const int best0 = best;
                mv[0] = x_mb; mv[1] = y_mb;
                for (int yy = y0; yy <= (best0 ? y1 : y0 - 1); yy++) /* a zero cost at the block's own position returns at once (:84); nothing later does */
                    for (int xx = x0; xx <= x1; xx++) {
                        const int v = mecmp.hadamard8_diff[0](NULL, c, s->b + (size_t)yy * BW + xx, BW, 16);
                        if (v < best) { best = v; mv[0] = xx; mv[1] = yy; }
                    }
            }
            s->cursor += mv[0] & 0; /* keeps the result live */
        }
        break;
    }
    }
}
static void *bench_worker(void *p)
{
    BenchLoop *l = p;
    l->ok = bench_make(&l->st);
    pthread_barrier_wait(l->bar);
    if (l->ok) bench_chunk(&l->st);
    pthread_barrier_wait(l->bar);
    pthread_barrier_wait(l->bar); /* the caller has set t_end */
    l->t_done = now_s();
    if (l->ok)
        do {
            bench_chunk(&l->st);
            l->chunks++;
            l->t_done = now_s();
        } while (l->t_done < l->t_end);
    pthread_barrier_wait(l->bar);
    bench_drop(&l->st);
    return NULL;
}
/* returns the number of threads that ran (<= 0: failure); *units = work done in *seconds */
int ffref_bench_leg(int leg, int threads, double min_seconds, double *units, double *seconds)
{
    dsp_init();
    if (leg < 0 || leg > 7) return -1;
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    static pthread_t th[1024];
    static BenchLoop loop[1024];
    pthread_barrier_t bar;
    if (pthread_barrier_init(&bar, NULL, threads + 1))
        return -1;
    int started = 0;
    for (int t = 0; t < threads; t++) {
        memset(&loop[t], 0, sizeof(loop[t]));
        loop[t].st.leg = leg;
        loop[t].st.thread = t;
        loop[t].bar = &bar;
        if (pthread_create(&th[t], NULL, bench_worker, &loop[t]))
            break;
        started++;
    }
    if (started != threads) {
        for (int t = 0; t < started; t++)
            pthread_cancel(th[t]);
        return -1;
    }
    pthread_barrier_wait(&bar); /* states built */
    pthread_barrier_wait(&bar); /* warm-up chunk done */
    const double t0 = now_s();
    for (int t = 0; t < started; t++)
        loop[t].t_end = t0 + min_seconds;
    pthread_barrier_wait(&bar);
    pthread_barrier_wait(&bar);
    double t1 = t0, chunks = 0;
    int ran = 0;
    for (int t = 0; t < started; t++) {
        if (loop[t].t_done > t1) t1 = loop[t].t_done;
        chunks += (double)loop[t].chunks;
        ran += loop[t].ok;
    }
    for (int t = 0; t < started; t++)
        pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    *units = chunks * (double)bench_units(leg);
    *seconds = t1 - t0;
    return ran;
}
