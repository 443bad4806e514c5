/*
 * integration/tests/sws_unscaled_hip_test.c — the frame-level SwsFunc hook (integration/swscale_unscaled_hip.c) through libswscale's
 * PUBLIC entry points, on host frames: sws_getContext() / sws_scale() / sws_setColorspaceDetails() / sws_freeContext() of the
 * reference's libraries (compiled where they lie), once with cpu flags 0 (the C converter) and once with AV_CPU_FLAG_HIP forced.
 *
 * For every case: the hip context's c->convert_unscaled is NOT the C context's pointer (the hook is installed), the pictures are
 * byte-identical (guard bytes right of every line and an untouched odd tail included), every sws_scale() call went through the hook and
 * none fell back to the C converter.  Cases: whole frames, source slices of 2..64 lines, bottom-up pictures (negative strides),
 * a colour matrix / range / brightness / contrast / saturation set AFTER the context was made, a pair the hook leaves alone.
 *
 * TEST INFRASTRUCTURE: built by oracle/refbuild (`make checkasm`), run on the GPU box by tests/test_gpu_sws_hook.py.
 * usage: sws_unscaled_hip_test [srcfmt dstfmt w h]   (no arguments: the built-in list, BASELINE configs[0] = yuv420p -> rgb24 1920x1080 first)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "libavutil/cpu.h"
#include "libavutil/imgutils.h"
#include "libavutil/lfg.h"
#include "libavutil/log.h"
#include "libavutil/mem.h"
#include "libavutil/pixdesc.h"
#include "libswscale/swscale.h"
#include "libswscale/swscale_internal.h"

#include "ffhip.h"
#include "hip_cpu.h"

long ff_sws_hip_unscaled_calls(const SwsInternal *c, long *fallbacks);

#define GUARD 32

typedef struct Pic {
    uint8_t *buf[4], *data[4];
    int      linesize[4], rows[4], wbytes[4];
} Pic;

static int pic_alloc(Pic *p, enum AVPixelFormat fmt, int w, int h, int flip)
{
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fmt);
    int n = av_pix_fmt_count_planes(fmt);
    memset(p, 0, sizeof(*p));
    for (int i = 0; i < n; i++) {
        const int chroma = (i == 1 || i == 2) && !(d->flags & AV_PIX_FMT_FLAG_RGB);
        p->wbytes[i] = av_image_get_linesize(fmt, w, i);
        p->rows[i] = chroma ? AV_CEIL_RSHIFT(h, d->log2_chroma_h) : h;
        p->linesize[i] = FFALIGN(p->wbytes[i] + GUARD, 64);
        p->buf[i] = av_malloc((size_t)p->linesize[i] * p->rows[i]);
        if (!p->buf[i])
            return -1;
        p->data[i] = p->buf[i];
        if (flip) { /* a bottom-up picture: the first line is the last in memory */
            p->data[i] = p->buf[i] + (size_t)p->linesize[i] * (p->rows[i] - 1);
            p->linesize[i] = -p->linesize[i];
        }
    }
    return n;
}

static void pic_free(Pic *p)
{
    for (int i = 0; i < 4; i++)
        av_freep(&p->buf[i]);
}

static void pic_fill(Pic *p, int n, AVLFG *lfg, int constant)
{
    for (int i = 0; i < n; i++) {
        const size_t sz = (size_t)abs(p->linesize[i]) * p->rows[i];
        for (size_t k = 0; k < sz; k++)
            p->buf[i][k] = constant >= 0 ? constant : av_lfg_get(lfg) >> 11;
    }
}

static int pic_equal(const Pic *a, const Pic *b, int n)
{
    for (int i = 0; i < n; i++)
        if (memcmp(a->buf[i], b->buf[i], (size_t)abs(a->linesize[i]) * a->rows[i]))
            return 0;
    return 1;
}

/* one conversion with the given cpu flags; slice_h == 0: one call for the frame */
static int convert(int cpu_flags, enum AVPixelFormat sf, enum AVPixelFormat df, int w, int h, const Pic *src, Pic *dst, int slice_h,
                   int colorspace, int expect_hook, SwsFunc *func, double *ms)
{
    SwsContext *sws;
    SwsInternal *c;
    long calls = 0, fb = 0;
    int ncalls = 0, r = 0;
    struct timespec t0, t1;
    av_force_cpu_flags(cpu_flags);
    sws = sws_getContext(w, h, sf, w, h, df, SWS_BICUBIC, NULL, NULL, NULL);
    if (!sws)
        return -1;
    c = sws_internal(sws);
    if (func)
        *func = c->convert_unscaled;
    if (colorspace) {
        /* after the init, as a player does once the stream's colour properties are known: BT.709, full-range source, and a touch of
         * brightness / contrast / saturation (libswscale/utils.c:848-1000 re-derives the tables on the live context) */
        if (sws_setColorspaceDetails(sws, sws_getCoefficients(SWS_CS_ITU709), 1, sws_getCoefficients(SWS_CS_DEFAULT), 0,
                                     colorspace > 1 ? 3 << 11 : 0, colorspace > 1 ? (1 << 16) + 5000 : 1 << 16,
                                     colorspace > 1 ? (1 << 16) - 9000 : 1 << 16) < 0) {
            fprintf(stderr, "sws_setColorspaceDetails failed\n");
            r = -1;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int y = 0; y < h && r >= 0; ) {
        const int sh = slice_h ? FFMIN(slice_h, h - y) : h;
        const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(sf);
        const uint8_t *sp[4] = { 0 };
        for (int i = 0; i < 4 && src->data[i]; i++)
            sp[i] = src->data[i] + (ptrdiff_t)(i == 1 || i == 2 ? y >> d->log2_chroma_h : y) * src->linesize[i];
        r = sws_scale(sws, sp, src->linesize, y, sh, dst->data, dst->linesize);
        ncalls++;
        if (r != sh) {
            fprintf(stderr, "sws_scale(y %d, h %d) returned %d\n", y, sh, r);
            r = -1;
        }
        y += sh;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (ms)
        *ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    calls = ff_sws_hip_unscaled_calls(c, &fb);
    if (r >= 0 && expect_hook && (calls != ncalls || fb != 0)) {
        fprintf(stderr, "hook: %ld of %d calls went through hip_convert_unscaled, %ld fell back to C (%s)\n", calls, ncalls, fb, ffhip_last_error());
        r = -2;
    }
    if (r >= 0 && !expect_hook && calls != -1) {
        fprintf(stderr, "hook installed where it should not be\n");
        r = -2;
    }
    sws_freeContext(sws); /* releases the libffhip context through c->hw_priv (utils.c:2257) */
    return r < 0 ? r : 0;
}

static int run_case(const char *sfn, const char *dfn, int w, int h, int slice_h, int flip, int colorspace, int expect_hook)
{
    const enum AVPixelFormat sf = av_get_pix_fmt(sfn), df = av_get_pix_fmt(dfn);
    Pic src, ref, out;
    AVLFG lfg;
    SwsFunc fc = NULL, fh = NULL;
    double ms_c = 0, ms_h = 0;
    int ns, nd, ok = 0;
    if (sf == AV_PIX_FMT_NONE || df == AV_PIX_FMT_NONE)
        return 2;
    av_lfg_init(&lfg, 0xF0F00001u ^ (unsigned)(w * 131 + h));
    ns = pic_alloc(&src, sf, w, h, flip);
    nd = pic_alloc(&ref, df, w, h, flip);
    if (ns < 0 || nd < 0 || pic_alloc(&out, df, w, h, flip) < 0)
        return 2;
    pic_fill(&src, ns, &lfg, -1);
    pic_fill(&ref, nd, &lfg, 0xA5); /* guard bytes, and whatever a converter leaves untouched */
    pic_fill(&out, nd, &lfg, 0xA5);
    if (convert(0, sf, df, w, h, &src, &ref, slice_h, colorspace, 0, &fc, &ms_c) < 0 ||
        convert(AV_CPU_FLAG_HIP, sf, df, w, h, &src, &out, slice_h, colorspace, expect_hook, &fh, &ms_h) < 0)
        ok = 0;
    else if (expect_hook && fc == fh) {
        fprintf(stderr, "c->convert_unscaled is the C function under AV_CPU_FLAG_HIP\n");
        ok = 0;
    } else
        ok = pic_equal(&ref, &out, nd);
    printf("%s %s -> %s %dx%d slice %d%s%s: %s  (C %.2f ms, hip incl. PCIe %.2f ms)\n", ok ? "OK  " : "FAIL", sfn, dfn, w, h, slice_h,
           flip ? " bottom-up" : "", colorspace == 2 ? " bt709/full/b-c-s" : colorspace ? " bt709/full" : "",
           expect_hook ? "hip SwsFunc == C" : "left to C", ms_c, ms_h);
    pic_free(&src); pic_free(&ref); pic_free(&out);
    return !ok;
}

int main(int argc, char **argv)
{
    static const char *const dsts[] = { "rgb24", "bgr24", "argb", "rgba", "abgr", "bgra", "gbrp" };
    static const char *const srcs[] = { "yuv420p", "yuv422p", "yuva420p" };
    int fails = 0, n = 0;
    av_log_set_level(AV_LOG_ERROR); /* ("No accelerated colorspace conversion found": the C selection says so for every context) */
    if (ffhip_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", ffhip_last_error());
        return 3;
    }
    if (argc >= 5)
        return run_case(argv[1], argv[2], atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 0, argc > 6 ? atoi(argv[6]) : 0,
                        argc > 7 ? atoi(argv[7]) : 0, 1);
    /* BASELINE configs[0]: sws_scale yuv420p -> rgb24 1920x1080, single frame */
    fails += run_case("yuv420p", "rgb24", 1920, 1080, 0, 0, 0, 1); n++;
    for (int s = 0; s < 3; s++)
        for (int d = 0; d < 7; d++) {
            fails += run_case(srcs[s], dsts[d], 1920, 1080, 0, 0, 0, 1); n++;
            fails += run_case(srcs[s], dsts[d], 354, 290, 0, 0, (s + d) % 3, 1); n++;   /* width % 16 != 0, pitches unaligned to the kernels' vectors */
        }
    fails += run_case("yuv420p", "rgb24", 1280, 720, 16, 0, 0, 1); n++;      /* slices, as a slice-threaded decoder hands them over */
    fails += run_case("yuv420p", "bgra", 1280, 720, 2, 0, 1, 1); n++;
    fails += run_case("yuv422p", "rgb24", 640, 480, 64, 0, 2, 1); n++;
    fails += run_case("yuva420p", "rgba", 640, 480, 6, 0, 0, 1); n++;
    fails += run_case("yuv420p", "rgb24", 1920, 1080, 0, 1, 0, 1); n++;      /* bottom-up: negative strides on both sides */
    fails += run_case("yuv420p", "gbrp", 642, 362, 0, 1, 2, 1); n++;
    fails += run_case("yuva420p", "argb", 642, 362, 10, 1, 1, 1); n++;
    fails += run_case("yuv420p", "rgb24", 3840, 2160, 0, 0, 2, 1); n++;
    /* pairs the hook leaves with the C converter: an odd width (the converter's tail case), a dithered 16-bit target */
    fails += run_case("yuv420p", "rgb24", 353, 290, 0, 0, 0, 0); n++;
    fails += run_case("yuv420p", "rgb565le", 352, 288, 0, 0, 0, 0); n++;
    printf("%d cases, %d failed\n", n, fails);
    return fails != 0;
}
