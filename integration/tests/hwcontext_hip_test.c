/*
 * integration/tests/hwcontext_hip_test.c — the `hip` AVHWDeviceType end to end, through libavutil's own generic entry points:
 *
 *   av_hwdevice_ctx_create -> av_hwframe_ctx_alloc/init (pool in HBM) -> av_hwframe_get_buffer -> av_hwframe_transfer_data (upload)
 *   -> ffhip_sws_scale_batch_dev on AVFrame.data[]/linesize[] of the device frames (no host copy in between)
 *   -> av_hwframe_transfer_data (download) -> compare with the reference's sws_scale() of the same host frame.
 *
 * TEST INFRASTRUCTURE: built by oracle/refbuild (`make hwcontext`) against the reference's libavutil/libswscale objects compiled where
 * they lie; run on the GPU box by tests/test_gpu_hwcontext.py.  usage: hwcontext_hip_test [srcW srcH dstW dstH [srcfmt dstfmt]]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "libavutil/frame.h"
#include "libavutil/hwcontext.h"
#include "libavutil/imgutils.h"
#include "libavutil/lfg.h"
#include "libavutil/pixdesc.h"
#include "libswscale/swscale.h"

#include "ffhip.h"
#include "avutil_hwcontext_hip.h"

#define CHECK(x) do { int r_ = (x); if (r_ < 0) { fprintf(stderr, "FAIL %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, r_, ffhip_last_error()); return 1; } } while (0)

static AVBufferRef *frames_ctx(AVBufferRef *dev, enum AVPixelFormat sw, int w, int h)
{
    AVBufferRef *ref = av_hwframe_ctx_alloc(dev);
    AVHWFramesContext *fc;
    if (!ref)
        return NULL;
    fc            = (AVHWFramesContext *)ref->data;
    fc->format    = FFHIP_HW_PIX_FMT;
    fc->sw_format = sw;
    fc->width     = w;
    fc->height    = h;
    if (av_hwframe_ctx_init(ref) < 0)
        av_buffer_unref(&ref);
    return ref;
}

static AVFrame *host_frame(enum AVPixelFormat fmt, int w, int h)
{
    AVFrame *f = av_frame_alloc();
    f->format = fmt;
    f->width  = w;
    f->height = h;
    return av_frame_get_buffer(f, 0) < 0 ? NULL : f;
}

long ffhip_integration_hw_launches(void);
long ffhip_integration_hw_refused(void);
long ffhip_integration_device_intermediates(void);
long ffhip_integration_legacy_launches(void);
long ffhip_integration_legacy_refused(void);

static void fill_frame(AVFrame *f, enum AVPixelFormat fmt, int w, int h, AVLFG *lfg)
{
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fmt);
    const int np = av_pix_fmt_count_planes(fmt);
    for (int p = 0; p < np; p++) {
        const int rows = (p == 1 || p == 2) ? AV_CEIL_RSHIFT(h, d->log2_chroma_h) : h;
        const int bw = av_image_get_linesize(fmt, w, p);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < bw; x++)
                f->data[p][y * f->linesize[p] + x] = av_lfg_get(lfg) >> 24;
        if (d->comp[0].depth > 8 && !(d->flags & AV_PIX_FMT_FLAG_FLOAT))
            for (int y = 0; y < rows; y++)
                for (int x = 0; x < bw / 2; x++) {
                    uint16_t *s = (uint16_t *)(f->data[p] + y * f->linesize[p]) + x;
                    *s = (*s & ((1 << d->comp[0].depth) - 1)) << d->comp[0].shift;
                }
    }
}

static int rows_differ(const AVFrame *a, const AVFrame *b, enum AVPixelFormat fmt, int w, int h)
{
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fmt);
    int bad = 0;
    for (int p = 0; p < av_pix_fmt_count_planes(fmt); p++) {
        const int rows = (p == 1 || p == 2) ? AV_CEIL_RSHIFT(h, d->log2_chroma_h) : h;
        const int bw = av_image_get_linesize(fmt, w, p);
        for (int y = 0; y < rows; y++)
            bad += !!memcmp(a->data[p] + y * a->linesize[p], b->data[p] + y * b->linesize[p], bw);
    }
    return bad;
}

/*
 * `graph` mode: sws_scale_frame() of the reference's new API on two frames of the hip device — ff_fmt_from_frame sees the hardware
 * format (format.c:351-357), the graph offers its op lists to the backend whose hw_format matches (ops_dispatch.c:113-117: the
 * hip_hw backend of integration/swscale_hw_hip.c), op_pass_run calls it with the device pointers — against the same call on host
 * frames with the C backend.  usage: hwcontext_hip_test graph srcfmt dstfmt w h [dw dh]
 */
static int graph_main(int argc, char **argv)
{
    const enum AVPixelFormat sf = av_get_pix_fmt(argv[2]), df = av_get_pix_fmt(argv[3]);
    const int sw = atoi(argv[4]), sh = atoi(argv[5]), dw = argc > 7 ? atoi(argv[6]) : sw, dh = argc > 7 ? atoi(argv[7]) : sh;
    AVBufferRef *dev = NULL, *sfc, *dfc;
    AVFrame *hs, *hd, *href, *ds = av_frame_alloc(), *dd = av_frame_alloc();
    SwsContext *g = sws_alloc_context(), *r = sws_alloc_context();
    AVLFG lfg;
    long before;
    int ret, bad;
    if (sf == AV_PIX_FMT_NONE || df == AV_PIX_FMT_NONE)
        return 2;
    if (ffhip_device_count() <= 0) {
        printf("SKIP no HIP device\n");
        return 77;
    }
    CHECK(av_hwdevice_ctx_create(&dev, FFHIP_HWDEVICE_TYPE, "0", NULL, 0));
    sfc = frames_ctx(dev, sf, sw, sh);
    dfc = frames_ctx(dev, df, dw, dh);
    if (!sfc || !dfc) {
        fprintf(stderr, "FAIL frames context\n");
        return 1;
    }
    hs = host_frame(sf, sw, sh);
    hd = host_frame(df, dw, dh);
    href = host_frame(df, dw, dh);
    av_lfg_init(&lfg, 0x5eed + sw);
    fill_frame(hs, sf, sw, sh, &lfg);
    CHECK(av_hwframe_get_buffer(sfc, ds, 0));
    CHECK(av_hwframe_get_buffer(dfc, dd, 0));
    CHECK(av_hwframe_transfer_data(ds, hs, 0));
    /* SWS_UNSTABLE: the op lists first (graph.c:727-754), the legacy scaler for what they do not take (subsampled formats) */
    g->flags = r->flags = SWS_BICUBIC | SWS_BITEXACT | SWS_ACCURATE_RND | SWS_UNSTABLE;
    g->threads = r->threads = 1;
    g->backends = SWS_BACKEND_LEGACY | SWS_BACKEND_C | 1 << 6; /* SWS_BACKEND_HIP; _C carries the format tests (the patch adds the new
                                           * bit to SWS_BACKEND_UNSTABLE, format.c:602-609) and is never offered hardware frames */
    r->backends = SWS_BACKEND_LEGACY | SWS_BACKEND_C;
    before = ffhip_integration_hw_launches() + ffhip_integration_legacy_launches();
    if ((ret = sws_scale_frame(g, dd, ds)) < 0) {
        fprintf(stderr, "FAIL sws_scale_frame on hip frames: %d (%s)\n", ret, ffhip_last_error());
        return 1;
    }
    if (ffhip_integration_legacy_refused()) {
        printf("REFUSED by libffhip's scaler: %s -> %s\n", argv[2], argv[3]);
        return 3;
    }
    if (ffhip_integration_hw_refused()) {
        printf("REFUSED a pass on a host pointer: %s -> %s\n", argv[2], argv[3]);
        return 3;
    }
    if (ffhip_integration_hw_launches() + ffhip_integration_legacy_launches() == before) {
        fprintf(stderr, "FAIL the hip_hw backend did not run\n");
        return 1;
    }
    CHECK(av_hwframe_transfer_data(hd, dd, 0));
    if ((ret = sws_scale_frame(r, href, hs)) < 0) {
        fprintf(stderr, "FAIL reference sws_scale_frame: %d\n", ret);
        return 1;
    }
    if ((bad = rows_differ(hd, href, df, dw, dh))) {
        fprintf(stderr, "FAIL %d rows differ from backend_c\n", bad);
        return 1;
    }
    printf("PASS sws_scale_frame on hip frames: %s %dx%d -> %s %dx%d in HBM (%ld op-list launches, %ld legacy-scaler passes, %ld intermediate planes in device memory), bit-exact with backend_c\n",
           argv[2], sw, sh, argv[3], dw, dh, ffhip_integration_hw_launches(), ffhip_integration_legacy_launches(), ffhip_integration_device_intermediates());
    if (getenv("HWTEST_REPEAT")) { /* the API-level rate: n sws_scale_frame() calls on the same two device frames, one launch each */
        const int n = atoi(getenv("HWTEST_REPEAT"));
        struct timespec t0, t1;
        ffhip_stream_synchronize(NULL);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < n; i++)
            if (sws_scale_frame(g, dd, ds) < 0)
                return 1;
        ffhip_stream_synchronize(NULL);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double sec = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        printf("RATE %d sws_scale_frame calls on hip frames: %.1f us per frame, %.1f frames/s, %.1f Mpixel/s out\n", n, 1e6 * sec / n, n / sec,
               1e-6 * n / sec * dw * dh);
    }
    sws_free_context(&g);
    sws_free_context(&r);
    av_frame_free(&hs); av_frame_free(&hd); av_frame_free(&href); av_frame_free(&ds); av_frame_free(&dd);
    av_buffer_unref(&sfc); av_buffer_unref(&dfc); av_buffer_unref(&dev);
    return 0;
}

int main(int argc, char **argv)
{
    const int sw = argc > 4 && strcmp(argv[1], "graph") ? atoi(argv[1]) : 640, sh = argc > 4 ? atoi(argv[2]) : 360;
    const int dw = argc > 4 ? atoi(argv[3]) : 1280, dh = argc > 4 ? atoi(argv[4]) : 720;
    const enum AVPixelFormat sf = argc > 6 ? av_get_pix_fmt(argv[5]) : AV_PIX_FMT_YUV420P, df = argc > 6 ? av_get_pix_fmt(argv[6]) : AV_PIX_FMT_YUV420P;
    const int flags = SWS_BICUBIC | SWS_BITEXACT | SWS_ACCURATE_RND;
    AVBufferRef *dev = NULL, *sfc, *dfc;
    AVHWFramesConstraints *cons;
    enum AVPixelFormat *fmts = NULL;
    AVHIPDeviceContext *hw;
    AVFrame *hs, *hd, *href, *ds, *dd;
    FFHipSwsContext *c;
    struct SwsContext *rc;
    AVLFG lfg;
    const void *sp[4] = { 0 };
    void *dp[4] = { 0 };
    size_t zero[4] = { 0 };
    int bad = 0, nplanes;

    if (argc > 5 && !strcmp(argv[1], "graph"))
        return graph_main(argc, argv);
    if (sf == AV_PIX_FMT_NONE || df == AV_PIX_FMT_NONE) {
        fprintf(stderr, "unknown pixel format\n");
        return 2;
    }
    /* (no device needed for the tables) */
    if (av_hwdevice_find_type_by_name("hip") != AV_HWDEVICE_TYPE_HIP || !av_hwdevice_get_type_name(AV_HWDEVICE_TYPE_HIP) ||
        strcmp(av_hwdevice_get_type_name(AV_HWDEVICE_TYPE_HIP), "hip") || av_hwdevice_find_type_by_name("cuda") != AV_HWDEVICE_TYPE_CUDA ||
        av_hwdevice_find_type_by_name("ohcodec") != AV_HWDEVICE_TYPE_OHCODEC) {
        fprintf(stderr, "FAIL hw_type_names[]\n");
        return 1;
    }
    {
        const AVPixFmtDescriptor *pd = av_pix_fmt_desc_get(AV_PIX_FMT_HIP), *pc = av_pix_fmt_desc_get(AV_PIX_FMT_CUDA);
        if (!pd || !(pd->flags & AV_PIX_FMT_FLAG_HWACCEL) || strcmp(pd->name, "hip") || av_get_pix_fmt("hip") != AV_PIX_FMT_HIP ||
            !pc || strcmp(pc->name, "cuda") || av_get_pix_fmt("cuda") != AV_PIX_FMT_CUDA || av_pix_fmt_desc_get_id(pd) != AV_PIX_FMT_HIP ||
            strcmp(av_get_pix_fmt_name(AV_PIX_FMT_HIP), "hip")) {
            fprintf(stderr, "FAIL av_pix_fmt_descriptors[]\n");
            return 1;
        }
    }
    printf("hip rows of hw_type_names[] / av_pix_fmt_descriptors[]: OK (type %d, format %d)\n", (int)AV_HWDEVICE_TYPE_HIP, (int)AV_PIX_FMT_HIP);
    if (ffhip_device_count() <= 0) {
        printf("SKIP no HIP device\n");
        return 77;
    }
    /* the type is found by name and by enum through the unmodified hwcontext.c, with enum values of its own */
    if (av_hwdevice_iterate_types(AV_HWDEVICE_TYPE_NONE) != AV_HWDEVICE_TYPE_HIP || AV_HWDEVICE_TYPE_HIP == AV_HWDEVICE_TYPE_CUDA) {
        fprintf(stderr, "FAIL the hip type is not in hw_table[]\n");
        return 1;
    }
    CHECK(av_hwdevice_ctx_create(&dev, FFHIP_HWDEVICE_TYPE, "0", NULL, 0));
    hw = ((AVHWDeviceContext *)dev->data)->hwctx;
    if (!hw->stream || hw->device != 0) {
        fprintf(stderr, "FAIL device context not initialised\n");
        return 1;
    }
    cons = av_hwdevice_get_hwframe_constraints(dev, NULL);
    if (!cons || cons->valid_hw_formats[0] != FFHIP_HW_PIX_FMT) {
        fprintf(stderr, "FAIL constraints\n");
        return 1;
    }
    av_hwframe_constraints_free(&cons);
    /* a format outside the list is refused by frames_init */
    if (frames_ctx(dev, AV_PIX_FMT_PAL8, 64, 64)) {
        fprintf(stderr, "FAIL pal8 frames context accepted\n");
        return 1;
    }
    sfc = frames_ctx(dev, sf, sw, sh);
    dfc = frames_ctx(dev, df, dw, dh);
    if (!sfc || !dfc) {
        fprintf(stderr, "FAIL frames context\n");
        return 1;
    }
    CHECK(av_hwframe_transfer_get_formats(sfc, AV_HWFRAME_TRANSFER_DIRECTION_TO, &fmts, 0));
    if (fmts[0] != sf || fmts[1] != AV_PIX_FMT_NONE) {
        fprintf(stderr, "FAIL transfer formats\n");
        return 1;
    }
    av_free(fmts);

    hs = host_frame(sf, sw, sh);
    hd = host_frame(df, dw, dh);
    href = host_frame(df, dw, dh);
    ds = av_frame_alloc();
    dd = av_frame_alloc();
    av_lfg_init(&lfg, 0x5eed);
    nplanes = av_pix_fmt_count_planes(sf);
    for (int p = 0; p < nplanes; p++) {
        const int rows = p ? AV_CEIL_RSHIFT(sh, av_pix_fmt_desc_get(sf)->log2_chroma_h) : sh;
        const int bw = av_image_get_linesize(sf, sw, p);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < bw; x++)
                hs->data[p][y * hs->linesize[p] + x] = av_lfg_get(&lfg) >> 24;
        if (av_pix_fmt_desc_get(sf)->comp[0].depth > 8) /* samples inside the format's range: depth bits at the format's shift */
            for (int y = 0; y < rows; y++)
                for (int x = 0; x < bw / 2; x++) {
                    uint16_t *s = (uint16_t *)(hs->data[p] + y * hs->linesize[p]) + x;
                    *s = (*s & ((1 << av_pix_fmt_desc_get(sf)->comp[0].depth) - 1)) << av_pix_fmt_desc_get(sf)->comp[0].shift;
                }
    }
    CHECK(av_hwframe_get_buffer(sfc, ds, 0));
    CHECK(av_hwframe_get_buffer(dfc, dd, 0));
    if (ds->format != FFHIP_HW_PIX_FMT || !ds->hw_frames_ctx || ds->linesize[0] % 256 || (uintptr_t)ds->data[1] % 256) {
        fprintf(stderr, "FAIL device frame layout\n");
        return 1;
    }
    CHECK(av_hwframe_transfer_data(ds, hs, 0));

    /* the consumer: device pointers and strides straight out of the AVFrames, on the device context's stream */
    c = ffhip_sws_getContext(sw, sh, sf, dw, dh, df, flags);
    if (!c) {
        fprintf(stderr, "FAIL ffhip_sws_getContext: %s\n", ffhip_last_error());
        return 1;
    }
    for (int p = 0; p < 4; p++) {
        sp[p] = ds->data[p];
        dp[p] = dd->data[p];
    }
    CHECK(ffhip_sws_scale_batch_dev(c, 1, sp, ds->linesize, zero, dp, dd->linesize, zero, hw->stream));
    CHECK(av_hwframe_transfer_data(hd, dd, 0)); /* ordered behind the kernels on hw->stream, synchronises */

    /* round trip of the upload alone: download the source frame again */
    {
        AVFrame *back = host_frame(sf, sw, sh);
        CHECK(av_hwframe_transfer_data(back, ds, 0));
        for (int p = 0; p < nplanes; p++) {
            const int rows = p ? AV_CEIL_RSHIFT(sh, av_pix_fmt_desc_get(sf)->log2_chroma_h) : sh;
            const int bw = av_image_get_linesize(sf, sw, p);
            for (int y = 0; y < rows; y++)
                bad += !!memcmp(back->data[p] + y * back->linesize[p], hs->data[p] + y * hs->linesize[p], bw);
        }
        if (bad) {
            fprintf(stderr, "FAIL upload/download round trip: %d rows differ\n", bad);
            return 1;
        }
        av_frame_free(&back);
    }

    rc = sws_getContext(sw, sh, sf, dw, dh, df, flags, NULL, NULL, NULL);
    if (!rc || sws_scale(rc, (const uint8_t *const *)hs->data, hs->linesize, 0, sh, href->data, href->linesize) != dh) {
        fprintf(stderr, "FAIL reference sws_scale\n");
        return 1;
    }
    nplanes = av_pix_fmt_count_planes(df);
    for (int p = 0; p < nplanes; p++) {
        const int rows = p ? AV_CEIL_RSHIFT(dh, av_pix_fmt_desc_get(df)->log2_chroma_h) : dh;
        const int bw = av_image_get_linesize(df, dw, p);
        for (int y = 0; y < rows; y++)
            bad += !!memcmp(hd->data[p] + y * hd->linesize[p], href->data[p] + y * href->linesize[p], bw);
    }
    if (bad) {
        fprintf(stderr, "FAIL %d output rows differ from the reference's sws_scale\n", bad);
        return 1;
    }
    printf("PASS hwcontext hip: %s %dx%d -> %s %dx%d, device frames scaled in HBM, bit-exact with sws_scale\n", av_get_pix_fmt_name(sf), sw, sh,
           av_get_pix_fmt_name(df), dw, dh);
    sws_freeContext(rc);
    ffhip_sws_freeContext(c);
    av_frame_free(&hs); av_frame_free(&hd); av_frame_free(&href); av_frame_free(&ds); av_frame_free(&dd);
    av_buffer_unref(&sfc); av_buffer_unref(&dfc); av_buffer_unref(&dev);
    return 0;
}
