/*
 * integration/tests/sws_scaled_hip_test.c — the frame-level SwsFunc of SCALED contexts (ff_sws_hip_scaled_hook(),
 * integration/swscale_unscaled_hip.c: installed at the end of ff_sws_init_scale() with the context's own banks) through libswscale's
 * PUBLIC entry points, on host frames: sws_getContext() / sws_scale() / sws_setColorspaceDetails() / sws_freeContext() of the
 * reference's libraries (compiled where they lie), once with cpu flags 0 (ff_swscale()'s line loop in C) and once with
 * AV_CPU_FLAG_HIP forced (libffhip's fused kernels behind c->convert_unscaled).
 *
 * For every case: the hook is installed (c->convert_unscaled is set where the C context has none), the pictures are byte-identical
 * (guard bytes right of every line included), every sws_scale() call went through the hook and none fell back to ff_swscale().
 * Cases: the conversions a player or an inference pipeline runs — NV12 into RGB at the source's size (the reference has no special
 * converter for it: the scaler), 1080p -> 4K (BASELINE configs[1]'s shape, one frame), 4K -> 1080p into RGB, other ratios, widths that
 * are not multiples of 4, thumbnails, a 10-bit source into an 8-bit target —, source slices, bottom-up pictures, colour details set
 * AFTER the context was made, pairs the hook leaves alone.
 *
 * TEST INFRASTRUCTURE: built by oracle/refbuild (`make checkasm`), run on the GPU box by tests/test_gpu_sws_hook.py.
 * usage: sws_scaled_hip_test [srcfmt sw sh dstfmt dw dh [flags slice_h flip colorspace]]   (no arguments: the built-in list)
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

long ff_sws_hip_scaled_calls(const SwsInternal *c, long *fallbacks);

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
static int convert(int cpu_flags, enum AVPixelFormat sf, int w, int h, enum AVPixelFormat df, int dw, int dh, int flags, const Pic *src, Pic *dst,
                   int slice_h, int colorspace, int expect_hook, SwsFunc *func, double *ms)
{
    SwsContext *sws;
    SwsInternal *c;
    long calls = 0, fb = 0;
    int ncalls = 0, r = 0;
    struct timespec t0, t1;
    av_force_cpu_flags(cpu_flags);
    sws = sws_getContext(w, h, sf, dw, dh, df, flags, NULL, NULL, NULL);
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
    if (slice_h < 0) {
        /* the frame API with TARGET slices (sws_frame_start / sws_send_slice / sws_receive_slice, swscale.c:1304-1395): the whole source is
         * sent, the target is received in slices of -slice_h lines (the first one in one piece when slice_h == -1: then the rest) */
        AVFrame *fs = av_frame_alloc(), *fd = av_frame_alloc();
        const int step = slice_h == -1 ? FFALIGN(dh / 2, 16) : -slice_h;
        fs->width = w; fs->height = h; fs->format = sf;
        fd->width = dw; fd->height = dh; fd->format = df;
        if (av_frame_get_buffer(fs, 0) < 0 || av_frame_get_buffer(fd, 0) < 0)
            r = -1;
        for (int i = 0; i < 4 && r >= 0 && src->data[i]; i++) {
            const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(sf);
            const int rows = i == 1 || i == 2 ? AV_CEIL_RSHIFT(h, d->log2_chroma_h) : h;
            for (int y = 0; y < rows; y++)
                memcpy(fs->data[i] + (ptrdiff_t)y * fs->linesize[i], src->data[i] + (ptrdiff_t)y * src->linesize[i], FFMIN(abs(fs->linesize[i]), abs(src->linesize[i])));
        }
        if (r >= 0 && (sws_frame_start(sws, fd, fs) < 0 || sws_send_slice(sws, 0, h) < 0))
            r = -1;
        for (int y = 0; y < dh && r >= 0; y += step) {
            r = sws_receive_slice(sws, y, FFMIN(step, dh - y));
            ncalls++;
            if (r < 0)
                fprintf(stderr, "sws_receive_slice(%d, %d) returned %d\n", y, FFMIN(step, dh - y), r);
        }
        sws_frame_end(sws);
        for (int i = 0; i < 4 && r >= 0 && dst->data[i]; i++) {
            const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(df);
            const int rows = i == 1 || i == 2 ? AV_CEIL_RSHIFT(dh, d->log2_chroma_h) : dh;
            for (int y = 0; y < rows; y++)
                memcpy(dst->data[i] + (ptrdiff_t)y * dst->linesize[i], fd->data[i] + (ptrdiff_t)y * fd->linesize[i], FFMIN(abs(fd->linesize[i]), abs(dst->linesize[i])));
        }
        av_frame_free(&fs);
        av_frame_free(&fd);
        expect_hook = expect_hook ? 2 : 0;          /* calls go through the hook, target slices fall back to ff_swscale() by design */
    }
    for (int y = 0; slice_h >= 0 && y < h && r >= 0; ) {
        const int sh = slice_h ? FFMIN(slice_h, h - y) : h;
        const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(sf);
        const uint8_t *sp[4] = { 0 };
        for (int i = 0; i < 4 && src->data[i]; i++)
            sp[i] = src->data[i] + (ptrdiff_t)(i == 1 || i == 2 ? y >> d->log2_chroma_h : y) * src->linesize[i];
        r = sws_scale(sws, sp, src->linesize, y, sh, dst->data, dst->linesize);
        ncalls++;
        if (r < 0) {
            fprintf(stderr, "sws_scale(y %d, h %d) returned %d\n", y, sh, r);
            r = -1;
        }
        y += sh;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (ms)
        *ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    calls = ff_sws_hip_scaled_calls(c, &fb);
    if (r >= 0 && expect_hook == 2 && calls != ncalls) {
        fprintf(stderr, "hook: %ld of %d sws_receive_slice calls went through hip_convert_scaled\n", calls, ncalls);
        r = -2;
    }
    if (r >= 0 && expect_hook == 1 && (calls != ncalls || fb != 0)) {
        fprintf(stderr, "hook: %ld of %d calls went through hip_convert_scaled, %ld fell back to ff_swscale (%s)\n", calls, ncalls, fb, ffhip_last_error());
        r = -2;
    }
    if (r >= 0 && !expect_hook && calls != -1) {
        fprintf(stderr, "hook installed where it should not be\n");
        r = -2;
    }
    sws_freeContext(sws); /* releases the libffhip context through c->hw_priv (utils.c:2257) */
    return r < 0 ? r : 0;
}

static int run_case(const char *sfn, int w, int h, const char *dfn, int dw, int dh, int flags, int slice_h, int flip, int colorspace, int expect_hook)
{
    const enum AVPixelFormat sf = av_get_pix_fmt(sfn), df = av_get_pix_fmt(dfn);
    Pic src, ref, out;
    AVLFG lfg;
    SwsFunc fc = NULL, fh = NULL;
    double ms_c = 0, ms_h = 0;
    int ns, nd, ok = 0;
    if (sf == AV_PIX_FMT_NONE || df == AV_PIX_FMT_NONE)
        return 2;
    av_lfg_init(&lfg, 0xF0F00002u ^ (unsigned)(w * 131 + h + dw));
    ns = pic_alloc(&src, sf, w, h, flip);
    nd = pic_alloc(&ref, df, dw, dh, flip);
    if (ns < 0 || nd < 0 || pic_alloc(&out, df, dw, dh, flip) < 0)
        return 2;
    pic_fill(&src, ns, &lfg, -1);
    if (av_pix_fmt_desc_get(sf)->comp[0].depth > 8) { /* samples of the format's depth (P01x: in the high bits) */
        const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(sf);
        for (int i = 0; i < ns; i++) {
            uint16_t *p = (uint16_t *)src.buf[i];
            const size_t n = (size_t)abs(src.linesize[i]) * src.rows[i] / 2;
            for (size_t k = 0; k < n; k++)
                p[k] = (uint16_t)((p[k] & ((1 << d->comp[0].depth) - 1)) << d->comp[0].shift);
        }
    }
    pic_fill(&ref, nd, &lfg, 0xA5); /* guard bytes, and whatever a converter leaves untouched */
    pic_fill(&out, nd, &lfg, 0xA5);
    if (convert(0, sf, w, h, df, dw, dh, flags, &src, &ref, slice_h, colorspace, 0, &fc, &ms_c) < 0 ||
        convert(AV_CPU_FLAG_HIP, sf, w, h, df, dw, dh, flags, &src, &out, slice_h, colorspace, expect_hook, &fh, &ms_h) < 0)
        ok = 0;
    else if (expect_hook && (fc || !fh)) {
        fprintf(stderr, "c->convert_unscaled: C context %s, hip context %s\n", fc ? "set" : "unset", fh ? "set" : "unset");
        ok = 0;
    } else
        ok = pic_equal(&ref, &out, nd);
    printf("%s %s %dx%d -> %s %dx%d flags %#x slice %d%s%s: %s  (C %.2f ms, hip incl. PCIe %.2f ms)\n", ok ? "OK  " : "FAIL", sfn, w, h, dfn, dw, dh,
           flags, slice_h, flip ? " bottom-up" : "", colorspace == 2 ? " bt709/full/b-c-s" : colorspace ? " bt709/full" : "",
           expect_hook ? "hip SwsFunc == ff_swscale" : "left to C", ms_c, ms_h);
    pic_free(&src); pic_free(&ref); pic_free(&out);
    return !ok;
}

int main(int argc, char **argv)
{
    int fails = 0, n = 0;
    av_log_set_level(AV_LOG_ERROR);
    if (ffhip_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", ffhip_last_error());
        return 3;
    }
    if (argc >= 7)
        return run_case(argv[1], atoi(argv[2]), atoi(argv[3]), argv[4], atoi(argv[5]), atoi(argv[6]), argc > 7 ? (int)strtol(argv[7], NULL, 0) : SWS_BICUBIC,
                        argc > 8 ? atoi(argv[8]) : 0, argc > 9 ? atoi(argv[9]) : 0, argc > 10 ? atoi(argv[10]) : 0, 1);
#define CASE(sf, w, h, df, dw, dh, fl, sl, flip, cs, hook) do { fails += run_case(sf, w, h, df, dw, dh, fl, sl, flip, cs, hook); n++; } while (0)
    /* a decoder's frame for a display or a network: NV12 into RGB at the source's size — no special converter, the scaler */
    CASE("nv12", 1920, 1080, "rgb24", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("nv12", 1920, 1080, "bgra", 1920, 1080, SWS_BICUBIC, 0, 0, 1, 1);
    CASE("nv21", 1080, 1920, "rgb24", 1080, 1920, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("yuv420p", 1280, 720, "rgb24", 1280, 720, SWS_BICUBIC | SWS_ACCURATE_RND, 0, 0, 2, 1);
    /* 4:4:4 into RGB at the source's size: no table converter either, full chroma forced (sws_full444.hip) */
    CASE("yuv444p", 1920, 1080, "rgb24", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("yuv444p", 1281, 720, "bgra", 1281, 720, SWS_BILINEAR, 0, 0, 1, 1);
    /* BASELINE configs[1]'s conversion, one frame */
    CASE("nv12", 1920, 1080, "nv12", 3840, 2160, SWS_BICUBIC, 0, 0, 0, 1);
    /* exact 2x into RGB, 4K -> 1080p into RGB (two stages), other ratios */
    CASE("yuv420p", 1920, 1080, "rgb24", 3840, 2160, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("nv12", 1920, 1080, "bgra", 3840, 2160, SWS_BICUBIC, 0, 0, 2, 1);
    CASE("nv12", 3840, 2160, "rgb24", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("yuv420p", 3840, 2160, "argb", 1920, 1080, SWS_BICUBIC, 0, 0, 1, 1);
    CASE("nv12", 1920, 1080, "nv12", 1280, 720, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("yuv420p", 1920, 1080, "yuv420p", 1280, 720, SWS_BILINEAR, 0, 0, 0, 1);
    CASE("nv12", 1280, 720, "bgra", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("nv12", 3840, 2160, "nv12", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    /* widths that are not multiples of 4 / 8, a thumbnail, a network input */
    CASE("nv12", 1920, 1080, "nv12", 854, 480, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("nv12", 1920, 1080, "rgb24", 854, 480, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("yuv420p", 1920, 1080, "yuv420p", 426, 240, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("nv12", 1920, 1080, "rgb24", 224, 224, SWS_BICUBIC, 0, 0, 0, 1);
    /* above 8 bits: like to like, and a 10-bit source into an 8-bit target (the ordered dither) */
    CASE("p010le", 1920, 1080, "p010le", 3840, 2160, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("yuv420p10le", 1280, 720, "yuv420p10le", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("p010le", 3840, 2160, "nv12", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("yuv420p10le", 1920, 1080, "yuv420p", 1280, 720, SWS_BICUBIC, 0, 0, 0, 1);
    /* round 6: 3:2 and 4:3 on the static-schedule kernels, both ways, 8 and 10 bits (sws_up32.hip, sws_down32.hip) */
    CASE("nv12", 1280, 720, "nv12", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("yuv420p", 1920, 1080, "yuv420p", 2560, 1440, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("p010le", 1920, 1080, "p010le", 2560, 1440, SWS_BILINEAR, 0, 0, 0, 1);
    CASE("p010le", 3840, 2160, "p010le", 2560, 1440, SWS_BICUBIC, 0, 0, 0, 1);
    /* source slices (libffhip collects them and scales when the frame is complete), bottom-up pictures */
    CASE("yuv420p", 640, 360, "rgb24", 1280, 720, SWS_BICUBIC, 16, 0, 0, 1);
    CASE("nv12", 1280, 720, "nv12", 640, 360, SWS_BICUBIC, 64, 0, 0, 1);
    CASE("nv12", 640, 360, "rgb24", 640, 360, SWS_BICUBIC, 0, 1, 0, 1);
    CASE("yuv420p", 1280, 720, "yuv420p", 854, 480, SWS_BICUBIC, 0, 1, 0, 1);
    /* the frame API asking for TARGET slices of a scaled picture: 64-line slices; a first half then the rest */
    CASE("nv12", 640, 360, "nv12", 1280, 720, SWS_BICUBIC, -64, 0, 0, 1);
    CASE("yuv420p", 1280, 720, "rgb24", 640, 360, SWS_BICUBIC, -1, 0, 0, 1);
    CASE("nv12", 640, 360, "rgb24", 640, 360, SWS_BICUBIC, -32, 0, 0, 1);
    /* packed RGB sources (round 6): a screen capture or an image for an encoder — the input converters run on the device, the context is
     * the one of their 14-bit lines; every component order, half and full chroma input, slices, a colourspace set after the context was made */
    CASE("bgra", 1920, 1080, "nv12", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("rgb24", 1280, 720, "yuv420p", 1280, 720, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("rgba", 1920, 1080, "yuv420p", 1280, 720, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("argb", 641, 361, "yuv444p", 641, 361, SWS_BILINEAR, 0, 0, 0, 1);
    CASE("abgr", 640, 360, "yuv420p", 1280, 720, SWS_BICUBIC | SWS_FULL_CHR_H_INP, 64, 0, 0, 1);
    CASE("bgr24", 1280, 720, "nv12", 1280, 720, SWS_BICUBIC | SWS_ACCURATE_RND, 0, 1, 0, 1);
    CASE("rgb24", 640, 360, "yuv420p", 640, 360, SWS_BICUBIC, 0, 0, 1, 1);
    /* 10-bit video for a display (round 6): the 16-bit walker into an intermediate, then the RGB writer; colour details set after the init */
    CASE("p010le", 1920, 1080, "bgra", 1920, 1080, SWS_BICUBIC, 0, 0, 1, 1);
    CASE("yuv420p10le", 3840, 2160, "rgb24", 1920, 1080, SWS_BICUBIC, 0, 0, 0, 1);
    CASE("yuv420p10le", 1280, 720, "rgba", 1920, 1080, SWS_BILINEAR, 0, 0, 2, 1);
    /* ... and bgr24 -> yuv420p at the source's size is the reference's own special converter (ff_rgb24toyv12): the hook finds it installed */
    CASE("bgr24", 640, 360, "yuv420p", 640, 360, SWS_BICUBIC, 0, 0, 0, 0);
    /* SWS_FAST_BILINEAR scales 8-bit sources through ff_hyscale_fast_c, not through the banks: left to ff_swscale() */
    CASE("yuv420p", 640, 360, "yuv420p", 1280, 720, SWS_FAST_BILINEAR, 0, 0, 0, 0);
    CASE("nv12", 640, 360, "rgb24", 1280, 720, SWS_FAST_BILINEAR, 0, 0, 0, 0);
    /* pairs the hook leaves with ff_swscale(): a dithered 16-bit target, a gray source */
    CASE("yuv420p", 352, 288, "rgb565le", 704, 576, SWS_BICUBIC, 0, 0, 0, 0);
    CASE("gray", 352, 288, "gray", 704, 576, SWS_BICUBIC, 0, 0, 0, 0);
    printf("%d cases, %d failed\n", n, fails);
    return fails != 0;
}
