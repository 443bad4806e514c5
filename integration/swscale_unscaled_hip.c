/*
 * integration/swscale_unscaled_hip.c — libswscale/hip/swscale_unscaled.c of the FFmpeg-side patch: the frame-level SwsFunc hook.
 *
 * ff_get_unscaled_swscale() picks a special converter for the context and ends with the per-arch calls
 * (libswscale/swscale_unscaled.c:2392, 2698-2704: ff_get_unscaled_swscale_ppc / _arm / _aarch64); the pointer it leaves in
 * c->convert_unscaled (swscale_internal.h:99-101, 359) is what scale_internal() calls for every slice (swscale.c:1163-1186) and what
 * tests/checkasm/sw_yuv2rgb.c:179-180 compares between arches.  The reference file is compiled where it lies with the function renamed
 * to ff_get_unscaled_swscale_c (oracle/refbuild/Makefile); the function below takes its name and adds the `hip` call behind the chain,
 * shaped like aarch64/swscale_unscaled.c:38-51, 200-260: when the converter just chosen is the table-driven yuv2rgb of
 * swscale_unscaled.c:2425-2431 (ff_yuv2rgb_get_func_ptr(), yuv2rgb.c:562-676) on a format pair libffhip's converter takes, a
 * libffhip context is made for the conversion and hip_convert_unscaled() — SwsFunc's exact signature — displaces the C function,
 * which stays behind as the fallback.
 *
 * Where the context handle lives: SwsInternal.hw_priv (swscale_internal.h:701), the refstruct slot sws_freeContext() releases first
 * (utils.c:2257).  Its one user in the reference is the Vulkan ops backend on the context that owns a hardware graph
 * (vulkan/ops.c:46-53); a context that got a special converter runs no op list (ff_sws_init_single_context() returns at
 * utils.c:1630-1636), so the slot is free — and the hook keeps its hands off a context whose slot is taken.  No field is added to
 * SwsInternal, no side table, and the handle dies with the context through the reference's own free path.
 *
 * The yuv2rgb coefficients are not context fields (locals of ff_yuv2rgb_c_init_tables(), yuv2rgb.c:750-797): they are derived again,
 * by the same arithmetic (ffhip_sws_yuv2rgb_coeffs), from what the context does store — c->srcColorspaceTable, sws->src_range,
 * c->brightness / contrast / saturation — and derived again when sws_setColorspaceDetails() changed those after the init
 * (utils.c:848-1000 re-runs ff_yuv2rgb_c_init_tables() on the live context; the aarch64 wrapper likewise reads c->yuv2rgb_* per call).
 */
#include <string.h>

#include "libavutil/attributes.h"
#include "libavutil/cpu.h"
#include "libavutil/pixdesc.h"
#include "libavutil/refstruct.h"
#include "libswscale/swscale_internal.h"

#include "ffhip.h"
#include "hip_cpu.h"

void ff_get_unscaled_swscale_c(SwsInternal *c);

typedef struct HipUnscaled {
    FFHipSwsContext *ctx;
    SwsFunc          c_func;          /* the converter the reference chose: runs when libffhip refuses a call */
    int              cs[4], range, brightness, contrast, saturation; /* what ctx's coefficients were derived from */
    long             calls, fallbacks;
} HipUnscaled;

static void hip_unscaled_free(AVRefStructOpaque opaque, void *obj)
{
    ffhip_sws_freeContext(((HipUnscaled *)obj)->ctx);
}

static int hip_colorspace_current(const SwsInternal *c, const HipUnscaled *u)
{
    return !memcmp(u->cs, c->srcColorspaceTable, sizeof(u->cs)) && u->range == c->opts.src_range &&
           u->brightness == c->brightness && u->contrast == c->contrast && u->saturation == c->saturation;
}

static void hip_colorspace_note(const SwsInternal *c, HipUnscaled *u)
{
    memcpy(u->cs, c->srcColorspaceTable, sizeof(u->cs));
    u->range = c->opts.src_range;
    u->brightness = c->brightness; u->contrast = c->contrast; u->saturation = c->saturation;
}

/* SwsFunc (swscale_internal.h:99-101): host pointers, a 2-line aligned slice of the source, returns the lines written */
static int hip_convert_unscaled(SwsInternal *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                                uint8_t *const dst[], const int dstStride[])
{
    HipUnscaled *u = c->hw_priv;
    int r = -1;
    if (!hip_colorspace_current(c, u)) {
        FFHipSwsTables t;
        memset(&t, 0, sizeof(t));
        if (ffhip_sws_yuv2rgb_coeffs(&t, c->srcColorspaceTable, c->opts.src_range, c->brightness, c->contrast, c->saturation) < 0 ||
            ffhip_sws_set_yuv2rgb(u->ctx, &t) < 0)
            goto c_path;
        hip_colorspace_note(c, u);
    }
    r = ffhip_sws_scale(u->ctx, src, srcStride, srcSliceY, srcSliceH, dst, dstStride);
c_path:
    __atomic_fetch_add(&u->calls, 1, __ATOMIC_RELAXED);
    if (r >= 0)
        return r;
    __atomic_fetch_add(&u->fallbacks, 1, __ATOMIC_RELAXED);
    return u->c_func(c, src, srcStride, srcSliceY, srcSliceH, dst, dstStride);
}

/* test instrumentation: calls that went through the hook / that fell back to the C converter, for a context the hook took (-1: not) */
long ff_sws_hip_unscaled_calls(const SwsInternal *c, long *fallbacks)
{
    const HipUnscaled *u = c->convert_unscaled == hip_convert_unscaled ? c->hw_priv : NULL;
    if (!u)
        return -1;
    if (fallbacks)
        *fallbacks = u->fallbacks;
    return u->calls;
}

static av_cold void ff_get_unscaled_swscale_hip(SwsInternal *c)
{
    const enum AVPixelFormat src = c->opts.src_format, dst = c->opts.dst_format;
    FFHipSwsTables t;
    HipUnscaled *u;
    int srcf, dstf;

    if (!(av_get_cpu_flags() & AV_CPU_FLAG_HIP) || !c->convert_unscaled || c->hw_priv)
        return;
    /* the table converter's branch, condition for condition (swscale_unscaled.c:2425-2431) ... */
    if (!((src == AV_PIX_FMT_YUV420P || src == AV_PIX_FMT_YUV422P || src == AV_PIX_FMT_YUVA420P) && isAnyRGB(dst) &&
          !(c->opts.flags & SWS_ACCURATE_RND) && (c->opts.dither == SWS_DITHER_BAYER || c->opts.dither == SWS_DITHER_AUTO) &&
          !(c->opts.dst_h & 1)))
        return;
    /* ... on the pairs libffhip's converter takes (ffhip.h: the equal-size converter).  An alpha plane the target has no room for is not
     * read (yuv2rgb.c:640-648: yuva2rgba_c / yuva2argb_c only for the 32-bit targets); the dithered 16-bit-and-below targets and the
     * 48-bit ones stay with the C function */
    switch (dst) {
    case AV_PIX_FMT_RGB24: dstf = FFHIP_PIX_FMT_RGB24; break;
    case AV_PIX_FMT_BGR24: dstf = FFHIP_PIX_FMT_BGR24; break;
    case AV_PIX_FMT_ARGB:  dstf = FFHIP_PIX_FMT_ARGB;  break;
    case AV_PIX_FMT_RGBA:  dstf = FFHIP_PIX_FMT_RGBA;  break;
    case AV_PIX_FMT_ABGR:  dstf = FFHIP_PIX_FMT_ABGR;  break;
    case AV_PIX_FMT_BGRA:  dstf = FFHIP_PIX_FMT_BGRA;  break;
    case AV_PIX_FMT_GBRP:  dstf = FFHIP_PIX_FMT_GBRP;  break;
    default: return;
    }
    switch (src) {
    case AV_PIX_FMT_YUV420P:  srcf = FFHIP_PIX_FMT_YUV420P; break;
    case AV_PIX_FMT_YUV422P:  srcf = FFHIP_PIX_FMT_YUV422P; break;
    case AV_PIX_FMT_YUVA420P: srcf = FFHIP_PIX_FMT_YUV420P; break;   /* the tables name the format without its alpha plane */
    default: return;
    }
    memset(&t, 0, sizeof(t));
    /* src[3] drives the alpha byte exactly where the C selection does (yuv2rgb.c:640-648) */
    t.dst_alpha_fill = CONFIG_SWSCALE_ALPHA && isALPHA(src) && isALPHA(dst) ? 2 : 0;
    t.srcW = c->opts.src_w; t.srcH = c->opts.src_h; t.srcFormat = srcf;
    t.dstW = c->opts.dst_w; t.dstH = c->opts.dst_h; t.dstFormat = dstf;
    t.flags = c->opts.flags;
    /* no banks: the context returns to its caller before initFilter() (utils.c:1625-1637) */
    if (ffhip_sws_yuv2rgb_coeffs(&t, c->srcColorspaceTable, c->opts.src_range, c->brightness, c->contrast, c->saturation) < 0)
        return;
    u = av_refstruct_alloc_ext(sizeof(*u), 0, NULL, hip_unscaled_free);
    if (!u)
        return;
    u->ctx = ffhip_sws_from_tables(&t);
    if (!u->ctx) {                      /* no device, or a shape libffhip leaves to the C converter (an odd width): C stays */
        av_refstruct_unref(&u);
        return;
    }
    hip_colorspace_note(c, u);
    u->c_func           = c->convert_unscaled;
    c->hw_priv          = u;
    c->convert_unscaled = hip_convert_unscaled;
    c->dst_slice_align  = 2;            /* as the C converter asks (swscale_unscaled.c:2430) */
}

av_cold void ff_get_unscaled_swscale(SwsInternal *c)
{
    ff_get_unscaled_swscale_c(c);
    ff_get_unscaled_swscale_hip(c);     /* the patch: one more line behind swscale_unscaled.c:2698-2704 */
}
