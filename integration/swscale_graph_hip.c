/*
 * integration/swscale_graph_hip.c — libswscale/graph.c for hip frames: the intermediate frames of a multi-pass graph in DEVICE memory.
 *
 * A conversion the graph cuts in two (two-dimensional scaling: one pass per filter direction, ops_dispatch.c:745-766) hands its
 * first pass's output to the second through a frame pass_alloc_output() allocates (graph.c:130-175) — with av_buffer_alloc(), host
 * memory, for every device type but Vulkan (pass_alloc_output_hw(), graph.c:105-128).  The patch gives hip the same branch Vulkan
 * has; expressed without touching the reference file: graph.c is compiled unchanged, where it lies, and for the duration of the
 * include
 *   - av_frame_alloc(), called once, at the top of pass_alloc_output() with `pass` in scope, also notes whether the pass belongs to a
 *     graph between two hip frames (SwsGraph.src / .dst .hw_format), and
 *   - av_buffer_alloc(), called once, for a plane of that frame, then returns an AVBufferRef over ffhip_malloc()'ed memory of the
 *     calling thread's current device (the hwcontext made it current: integration/avutil_hwcontext_hip.c).
 * The recipe (oracle/refbuild/Makefile) checks that graph.c calls each of the two exactly once.  The second half of this file: the
 * legacy scaler's passes of such a graph on the device.
 */
#include "config.h"
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "libavutil/buffer.h"
#include "libavutil/frame.h"
#include "libavutil/log.h"
#include "libavutil/pixfmt.h"
#include "libswscale/swscale.h"
#include "libswscale/swscale_internal.h"
#include "avutil_hwcontext_hip.h"
#include "ffhip.h"

struct SwsPass;
static AVBufferRef *ffhip_graph_buffer_alloc(size_t size);
static void ffhip_graph_note_pass(const struct SwsPass *pass);
static SwsInternal *ffhip_graph_internal(const SwsContext *sws);
static int ffhip_graph_swscale(SwsInternal *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                               uint8_t *const dst[], const int dstStride[], int dstSliceY, int dstSliceH);
static void ffhip_graph_free_context(SwsContext **sws);
static inline SwsInternal *ref_sws_internal(const SwsContext *sws) { return sws_internal(sws); }
static SwsBackend ffhip_graph_backends(const SwsContext *ctx, const SwsFormat **src, const SwsFormat **dst, const char *func);

/*
 * The legacy scaler's passes (subsampled formats: the ops backends are not offered those yet, format.c:560-600) on hip frames:
 *   - sws_internal(), the first thing init_legacy_subpass() calls on a legacy context, clears c->convert_unscaled when the graph
 *     is being built by a sws_scale_frame() between two hip frames (integration/swscale_frame_hip.c notes that), so every such pass
 *     runs run_legacy_swscale() -> ff_swscale();
 *   - ff_swscale() called with device pointers is ffhip_sws_scale_batch_dev() on a context built for the legacy context's
 *     formats, sizes, flags and ranges (libffhip's own initFilter() restatement: banks identical to the reference's, pinned by
 *     tests/test_oracle_vs_ref.py); called with host pointers it is the reference's ff_swscale();
 *   - sws_free_context() drops that context.
 * A conversion libffhip does not take is refused (logged, counted: ffhip_integration_hw_refused) and leaves the frame untouched.
 */
#define av_buffer_alloc ffhip_graph_buffer_alloc
#define av_frame_alloc() (ffhip_graph_note_pass(pass), (av_frame_alloc)())
#define sws_internal(s) ffhip_graph_internal(s)
#define ff_swscale ffhip_graph_swscale
#define sws_free_context ffhip_graph_free_context
/* add_legacy_sws_pass() refuses hardware frames (graph.c:569-570: the patch adds `&& hw_format != AV_PIX_FMT_HIP`): its one call of
 * ff_sws_enabled_backends(), right above that test with `src` and `dst` in scope, re-points the two at copies without the hardware
 * format when both are hip frames (the other caller, add_ops_convert_pass(), keeps them: the op backends select by hw_format) */
#define ff_sws_enabled_backends(ctx) ffhip_graph_backends(ctx, &src, &dst, __func__)
#include "libswscale/graph.c"
#undef ff_sws_enabled_backends
#undef sws_free_context
#undef ff_swscale
#undef sws_internal
#undef av_frame_alloc
#undef av_buffer_alloc

extern _Thread_local int ffhip_integration_hip_frames;
static _Thread_local int hip_graph; /* the frame being allocated belongs to a graph between two hip frames */
static long device_intermediates, legacy_launches, legacy_refused;
long ffhip_integration_device_intermediates(void) { return device_intermediates; }
long ffhip_integration_legacy_launches(void) { return legacy_launches; }
long ffhip_integration_legacy_refused(void) { return legacy_refused; }

static void ffhip_graph_note_pass(const struct SwsPass *pass)
{
    const SwsGraph *graph = pass->graph;
    hip_graph = graph->src.hw_format == FFHIP_HW_PIX_FMT && graph->dst.hw_format == FFHIP_HW_PIX_FMT;
}

static void device_free(void *opaque, uint8_t *data)
{
    (void)opaque;
    ffhip_free(data);
}

static AVBufferRef *ffhip_graph_buffer_alloc(size_t size)
{
    void *p = NULL;
    AVBufferRef *ref;
    if (!hip_graph)
        return av_buffer_alloc(size);
    if (ffhip_malloc(&p, size) < 0 || !p)
        return NULL;
    ref = av_buffer_create(p, size, device_free, NULL, 0);
    if (!ref)
        ffhip_free(p);
    else
        __atomic_fetch_add(&device_intermediates, 1, __ATOMIC_RELAXED);
    return ref;
}

/* ---- the legacy scaler's passes on device frames ---- */
static SwsBackend ffhip_graph_backends(const SwsContext *ctx, const SwsFormat **src, const SwsFormat **dst, const char *func)
{
    static _Thread_local SwsFormat s, d;
    if (!strcmp(func, "add_legacy_sws_pass") && (*src)->hw_format == FFHIP_HW_PIX_FMT && (*dst)->hw_format == FFHIP_HW_PIX_FMT) {
        s = **src;
        d = **dst;
        s.hw_format = d.hw_format = AV_PIX_FMT_NONE;
        *src = &s;
        *dst = &d;
    }
    return ff_sws_enabled_backends(ctx);
}

static SwsInternal *ffhip_graph_internal(const SwsContext *sws)
{
    SwsInternal *c = ref_sws_internal(sws);
    if (ffhip_integration_hip_frames && c->convert_unscaled)
        c->convert_unscaled = NULL; /* the special converters take host pointers: every pass of a hip graph goes through ff_swscale() */
    return c;
}

#define NCTX 64
static struct { const SwsInternal *key; FFHipSwsContext *ctx; } ctxs[NCTX];
static pthread_mutex_t ctx_mu = PTHREAD_MUTEX_INITIALIZER;

static FFHipSwsContext *device_context(const SwsInternal *c)
{
    FFHipSwsContext *x = NULL;
    int slot = -1;
    pthread_mutex_lock(&ctx_mu);
    for (int i = 0; i < NCTX; i++) {
        if (ctxs[i].key == c)
            x = ctxs[i].ctx;
        else if (!ctxs[i].key && slot < 0)
            slot = i;
    }
    if (!x && slot >= 0) {
        FFHipSwsHostTables *ht = ffhip_sws_tables_create(c->opts.src_w, c->opts.src_h, c->opts.src_format, c->opts.dst_w, c->opts.dst_h,
                                                         c->opts.dst_format, (int)c->opts.flags);
        FFHipSwsTables t;
        if (ht && ffhip_sws_tables_set_ranges(ht, c->opts.src_range, c->opts.dst_range) >= 0 && ffhip_sws_tables_get(ht, &t) >= 0)
            x = ffhip_sws_from_tables(&t);
        if (ht)
            ffhip_sws_tables_free(ht);
        if (x) {
            ctxs[slot].key = c;
            ctxs[slot].ctx = x;
        }
    }
    pthread_mutex_unlock(&ctx_mu);
    return x;
}

static int ffhip_graph_swscale(SwsInternal *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                               uint8_t *const dst[], const int dstStride[], int dstSliceY, int dstSliceH)
{
    const int sdev = ffhip_pointer_device(src[0]) >= 0, ddev = ffhip_pointer_device(dst[0]) >= 0;
    const size_t zero[4] = { 0, 0, 0, 0 };
    FFHipSwsContext *x;
    if (!sdev && !ddev)
        return ff_swscale(c, src, srcStride, srcSliceY, srcSliceH, dst, dstStride, dstSliceY, dstSliceH);
    if (dstSliceY)
        return dstSliceH; /* the first slice's call scaled the frame: libffhip's scaled contexts are whole-frame */
    x = sdev && ddev ? device_context(c) : NULL;
    if (!x || ffhip_sws_scale_batch_dev(x, 1, (const void *const *)src, srcStride, zero, (void *const *)dst, dstStride, zero, NULL) < 0) {
        if (!__atomic_fetch_add(&legacy_refused, 1, __ATOMIC_RELAXED))
            av_log(NULL, AV_LOG_ERROR, "hip: %s -> %s %dx%d -> %dx%d on device frames is not taken by libffhip (%s); not run\n",
                   av_get_pix_fmt_name(c->opts.src_format), av_get_pix_fmt_name(c->opts.dst_format), c->opts.src_w, c->opts.src_h,
                   c->opts.dst_w, c->opts.dst_h, ffhip_last_error());
        return dstSliceH;
    }
    __atomic_fetch_add(&legacy_launches, 1, __ATOMIC_RELAXED);
    return dstSliceH;
}

static void ffhip_graph_free_context(SwsContext **sws)
{
    if (sws && *sws) {
        const SwsInternal *c = ref_sws_internal(*sws);
        pthread_mutex_lock(&ctx_mu);
        for (int i = 0; i < NCTX; i++)
            if (ctxs[i].key == c) {
                ffhip_sws_freeContext(ctxs[i].ctx);
                ctxs[i].key = NULL;
                ctxs[i].ctx = NULL;
            }
        pthread_mutex_unlock(&ctx_mu);
    }
    sws_free_context(sws);
}
