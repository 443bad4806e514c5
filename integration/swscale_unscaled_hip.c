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
    SwsFunc          c_func;          /* the converter the reference chose: runs when libffhip refuses a call (scaled contexts: ff_swscale) */
    int              scaled;          /* a scaled context (ff_sws_hip_scaled_hook() below) ... */
    int              graph;           /* ... made by the filter graph (not sws_init_context()): slices arrive in TARGET lines */
    int              cs[4], range, brightness, contrast, saturation; /* what ctx's coefficients were derived from */
    int              rgb_src;         /* a packed RGB source (ffhip_sws_from_tables_rgb_source) and the converter table the context holds */
    int32_t          rgb2yuv[9];
    long             calls, fallbacks;
    /* where the frame in flight runs: libffhip converts when a frame's LAST source slice arrives, so a frame must stay with
     * the side that took its first slice.  0: the next call starts a frame; 1: libffhip holds the earlier slices; 2: the C path has it */
    int              frame_path, frame_lines;
} HipUnscaled;

/* One source slice of a frame: libffhip, or the C function for the WHOLE frame when libffhip refuses the frame's first slice.  A failure
 * after libffhip has taken earlier slices cannot be replayed (the caller's earlier slice buffers are no longer ours to read): it is
 * returned as an error instead of a frame whose upper part was never converted. */
static int hip_slice(SwsInternal *c, HipUnscaled *u, int prepared, const uint8_t *const src[], const int srcStride[], int y, int h,
                     uint8_t *const dst[], const int dstStride[], int *fell_back)
{
    const int first = u->frame_path == 0;
    int r = -1;
    if (first)
        u->frame_path = prepared ? 1 : 2;
    if (u->frame_path == 1) {
        r = ffhip_sws_scale(u->ctx, src, srcStride, y, h, dst, dstStride);
        if (r < 0 && first)
            u->frame_path = 2;
        else if (r < 0)
            r = AVERROR_EXTERNAL;
    }
    *fell_back = u->frame_path == 2;
    u->frame_lines += h;
    if (u->frame_lines >= c->opts.src_h || r == AVERROR_EXTERNAL)
        u->frame_lines = u->frame_path = 0;           /* the frame is complete (scale_internal() resets its sliceDir at the same point) */
    return r;
}

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
    int r, prepared = 1, fell_back = 0;
    if (!hip_colorspace_current(c, u)) {
        FFHipSwsTables t;
        memset(&t, 0, sizeof(t));
        if (ffhip_sws_yuv2rgb_coeffs(&t, c->srcColorspaceTable, c->opts.src_range, c->brightness, c->contrast, c->saturation) < 0 ||
            ffhip_sws_set_yuv2rgb(u->ctx, &t) < 0)
            prepared = 0;
        else
            hip_colorspace_note(c, u);
    }
    /* (a TARGET slice of sws_receive_slice() reaches an unscaled converter with both pointer sets moved to the slice, swscale.c:1163-1179:
     * for a conversion at the source's size that IS the source slice of the same lines) */
    r = hip_slice(c, u, prepared, src, srcStride, srcSliceY, srcSliceH, dst, dstStride, &fell_back);
    __atomic_fetch_add(&u->calls, 1, __ATOMIC_RELAXED);
    if (!fell_back)
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

/*
 * SCALED contexts (and the unscaled ones the reference has no special converter for: NV12 -> RGB goes through the scaler).
 *
 * scale_internal() calls c->convert_unscaled for ANY context that has one (libswscale/swscale.c:1163-1186) and ff_swscale() otherwise,
 * and the filter graph's legacy pass does the same (graph.c:394-404, 497).  ff_sws_init_scale() is the last step of
 * ff_sws_init_single_context() (utils.c:1797), when the four banks stand in c->{h,v}{Lum,Chr}Filter[Pos|Size]: the `hip` arch's
 * ff_sws_init_scale() (integration/swscale_hip.c) ends with a call of the function below, which hands the context's OWN banks, range
 * constants and colour details to libffhip (ffhip_sws_from_tables) and, when libffhip takes the conversion, installs a SwsFunc — so that
 * sws_scale() / sws_scale_frame() on host frames run the fused kernels instead of ff_swscale()'s line loop.  Not installed for slice
 * threading (threads != 1, or a slice context of a threaded parent: the reference then asks for TARGET slices of a scaled picture,
 * swscale.c:1645-1679, graph.c:505-545), for cascaded / gamma contexts (they delegate), when hw_priv is taken, or for an error-diffusion
 * dither.  libffhip collects source slices in order and scales when the frame is complete (ffhip_sws_scale); a call it refuses runs
 * ff_swscale().
 */
static int hip_convert_scaled(SwsInternal *c, const uint8_t *const src[], const int srcStride[], int y, int h,
                              uint8_t *const dst[], const int dstStride[])
{
    HipUnscaled *u = c->hw_priv;
    int r, prepared = 1, fell_back = 0;
    if (isAnyRGB(c->opts.dst_format) && !hip_colorspace_current(c, u)) {
        FFHipSwsTables t;
        memset(&t, 0, sizeof(t));
        if (ffhip_sws_yuv2rgb_coeffs(&t, c->srcColorspaceTable, c->opts.src_range, c->brightness, c->contrast, c->saturation) < 0 ||
            ffhip_sws_set_yuv2rgb(u->ctx, &t) < 0)
            prepared = 0;
        else
            hip_colorspace_note(c, u);
    }
    if (u->rgb_src) {
        /* sws_setColorspaceDetails() after the context was made rewrites c->input_rgb2yuv_table (fill_rgb2yuv_table, utils.c:1002) and may
         * open a range stage behind the converters (ff_sws_init_range_convert): the table is handed over again, the range stage is the C path's */
        if (c->opts.src_range != c->opts.dst_range)
            prepared = 0;
        else if (memcmp(u->rgb2yuv, c->input_rgb2yuv_table, sizeof(u->rgb2yuv))) {
            if (ffhip_sws_set_rgb2yuv(u->ctx, c->input_rgb2yuv_table) < 0)
                prepared = 0;
            else
                memcpy(u->rgb2yuv, c->input_rgb2yuv_table, sizeof(u->rgb2yuv));
        }
    }
    if (u->graph) {
        /* run_legacy_unscaled() (graph.c:394-404): y, h are lines of the pass, i.e. of the TARGET, and the graph runs this pass in one
         * slice (threads == 1 is a condition of the hook): the frame */
        if (y != 0 || h != c->opts.dst_h)
            prepared = 0;
        else
            h = c->opts.src_h;
    } else if (c->frame_src && c->frame_src->data[0] && !(y == 0 && h == c->opts.src_h)) {
        /* sws_receive_slice() on a legacy context asking for a TARGET slice: the frame API is in flight (sws_frame_start()
         * holds the source in c->frame_src), and scale_internal() has treated this context as an unscaled one — y, h are the target slice,
         * every source pointer was moved down by y (chroma: y >> chrSrcVSubSample) lines and every target pointer up to the frame's first
         * line (swscale.c:1163-1179).  Undo both and let ff_swscale() produce the slice from the whole source, as it would have been
         * called without a converter (swscale.c:1184-1186). */
        const uint8_t *s2[4] = { src[0], src[1], src[2], src[3] };
        uint8_t *d2[4] = { dst[0], dst[1], dst[2], dst[3] };
        for (int i = 0; i < 4 && s2[i]; i++) {
            if (i > 0 && usePal(c->opts.src_format))
                break;
            s2[i] -= (y >> ((i == 1 || i == 2) ? c->chrSrcVSubSample : 0)) * (ptrdiff_t)srcStride[i];
        }
        for (int i = 0; i < 4 && d2[i]; i++) {
            if (i > 0 && usePal(c->opts.dst_format))
                break;
            d2[i] += (y >> ((i == 1 || i == 2) ? c->chrDstVSubSample : 0)) * (ptrdiff_t)dstStride[i];
        }
        __atomic_fetch_add(&u->calls, 1, __ATOMIC_RELAXED);
        __atomic_fetch_add(&u->fallbacks, 1, __ATOMIC_RELAXED);
        return ff_swscale(c, s2, srcStride, 0, c->opts.src_h, d2, dstStride, y, h);
    }
    r = hip_slice(c, u, prepared, src, srcStride, y, h, dst, dstStride, &fell_back);
    __atomic_fetch_add(&u->calls, 1, __ATOMIC_RELAXED);
    if (!fell_back)
        return r;
    __atomic_fetch_add(&u->fallbacks, 1, __ATOMIC_RELAXED);
    return ff_swscale(c, src, srcStride, y, h, dst, dstStride, 0, c->opts.dst_h);
}

/* test instrumentation, as ff_sws_hip_unscaled_calls() */
long ff_sws_hip_scaled_calls(const SwsInternal *c, long *fallbacks)
{
    const HipUnscaled *u = c->convert_unscaled == hip_convert_scaled ? c->hw_priv : NULL;
    if (!u)
        return -1;
    if (fallbacks)
        *fallbacks = u->fallbacks;
    return u->calls;
}

static int hip_scaled_format(enum AVPixelFormat f, int target)
{
    switch (f) {
    /* FFHIP_PIX_FMT_* == AV_PIX_FMT_* (include/ffhip.h) */
    case AV_PIX_FMT_YUV420P: case AV_PIX_FMT_YUV422P: case AV_PIX_FMT_YUV444P: case AV_PIX_FMT_NV12: case AV_PIX_FMT_NV21:
    case AV_PIX_FMT_YUV420P9LE: case AV_PIX_FMT_YUV420P10LE: case AV_PIX_FMT_YUV420P12LE: case AV_PIX_FMT_YUV420P14LE: case AV_PIX_FMT_YUV420P16LE:
    case AV_PIX_FMT_YUV422P9LE: case AV_PIX_FMT_YUV422P10LE: case AV_PIX_FMT_YUV422P12LE: case AV_PIX_FMT_YUV422P14LE: case AV_PIX_FMT_YUV422P16LE:
    case AV_PIX_FMT_YUV444P9LE: case AV_PIX_FMT_YUV444P10LE: case AV_PIX_FMT_YUV444P12LE: case AV_PIX_FMT_YUV444P14LE: case AV_PIX_FMT_YUV444P16LE:
    case AV_PIX_FMT_P010LE: case AV_PIX_FMT_P012LE: case AV_PIX_FMT_P016LE:
        return (int)f;
    case AV_PIX_FMT_RGB24: case AV_PIX_FMT_BGR24: case AV_PIX_FMT_ARGB: case AV_PIX_FMT_RGBA: case AV_PIX_FMT_ABGR: case AV_PIX_FMT_BGRA:
        return target ? (int)f : -1;
    default:
        return -1;
    }
}

av_cold void ff_sws_hip_scaled_hook(SwsInternal *c)
{
    const enum AVPixelFormat src = c->opts.src_format, dst = c->opts.dst_format;
    int srcf = hip_scaled_format(src, 0), dstf = hip_scaled_format(dst, 1), rgb_src = 0;
    FFHipSwsTables t;
    HipUnscaled *u;

    /* a packed 8-bit RGB source in front of a YUV target: libffhip runs the input converters itself and takes the context as the one of
     * their 14-bit lines (ffhip_sws_from_tables_rgb_source()); no alpha plane into the target, no vertical chroma drop, equal ranges */
    if (srcf < 0 && dstf >= 0 && !isAnyRGB(dst) && hip_scaled_format(src, 1) >= 0 && !c->needAlpha && !c->chrSrcVSubSample &&
        c->opts.src_range == c->opts.dst_range && !c->readLumPlanar && !c->readChrPlanar) {
        rgb_src = (int)src;
        srcf = c->chrSrcHSubSample ? AV_PIX_FMT_YUV422P14LE : AV_PIX_FMT_YUV444P14LE;
    }

    if (!(av_get_cpu_flags() & AV_CPU_FLAG_HIP) || c->convert_unscaled || c->hw_priv || c->parent || c->nb_slice_ctx ||
        c->opts.threads != 1 || c->cascaded_context[0] || c->opts.gamma_flag || srcf < 0 || dstf < 0 ||
        /* SWS_FAST_BILINEAR: 8-bit sources then scale through ff_hyscale_fast_c / ff_hcscale_fast_c (swscale.c:679, hscale.c:54-77), not
         * through the banks handed over below — ffhip_sws_tables_create() refuses the flag for the same reason */
        (c->opts.flags & SWS_FAST_BILINEAR) || c->hyscale_fast || c->hcscale_fast ||
        (c->opts.dither != SWS_DITHER_AUTO && c->opts.dither != SWS_DITHER_BAYER) || c->srcXYZ || c->dstXYZ || c->src0Alpha || c->dst0Alpha)
        return;
    memset(&t, 0, sizeof(t));
    t.srcW = c->opts.src_w; t.srcH = c->opts.src_h; t.srcFormat = srcf;
    t.dstW = c->opts.dst_w; t.dstH = c->opts.dst_h; t.dstFormat = dstf;
    t.flags = c->opts.flags;
    t.hLum = (FFHipSwsFilter){ c->hLumFilter, c->hLumFilterPos, c->hLumFilterSize, c->opts.dst_w };
    t.hChr = (FFHipSwsFilter){ c->hChrFilter, c->hChrFilterPos, c->hChrFilterSize, c->chrDstW };
    t.vLum = (FFHipSwsFilter){ c->vLumFilter, c->vLumFilterPos, c->vLumFilterSize, c->opts.dst_h };
    t.vChr = (FFHipSwsFilter){ c->vChrFilter, c->vChrFilterPos, c->vChrFilterSize, c->chrDstH };
    if (!t.hLum.filter || !t.hChr.filter || !t.vLum.filter || !t.vChr.filter)
        return;
    /* range conversion between YUV formats: the constants ff_sws_init_range_convert() computed (swscale.c:568-660) */
    t.src_range = c->opts.src_range; t.dst_range = c->opts.dst_range;
    t.lumConvertRange_coeff  = c->lumConvertRange_coeff;  t.chrConvertRange_coeff  = c->chrConvertRange_coeff;
    t.lumConvertRange_offset = c->lumConvertRange_offset; t.chrConvertRange_offset = c->chrConvertRange_offset;
    /* (the coefficient fields are part of every table set — ffhip_sws_from_tables() checks them — and only RGB targets read them) */
    if (ffhip_sws_yuv2rgb_coeffs(&t, c->srcColorspaceTable, c->opts.src_range, c->brightness, c->contrast, c->saturation) < 0)
        return;
    t.full_chr_h_int = isAnyRGB(dst) && (c->opts.flags & SWS_FULL_CHR_H_INT);
    u = av_refstruct_alloc_ext(sizeof(*u), 0, NULL, hip_unscaled_free);
    if (!u)
        return;
    u->ctx = rgb_src ? ffhip_sws_from_tables_rgb_source(&t, rgb_src, c->input_rgb2yuv_table) : ffhip_sws_from_tables(&t);
    if (!u->ctx) {                      /* no device, or a conversion libffhip does not take: ff_swscale() stays */
        av_refstruct_unref(&u);
        return;
    }
    hip_colorspace_note(c, u);
    u->rgb_src = rgb_src;
    memcpy(u->rgb2yuv, c->input_rgb2yuv_table, sizeof(u->rgb2yuv));
    u->scaled = 1;
    u->graph  = !c->is_legacy_init;     /* sws_init_context() sets it before it gets here (utils.c:1892); the graph's contexts are not made by it */
    c->hw_priv          = u;
    c->convert_unscaled = hip_convert_scaled;
}

av_cold void ff_get_unscaled_swscale(SwsInternal *c)
{
    ff_get_unscaled_swscale_c(c);
    ff_get_unscaled_swscale_hip(c);     /* the patch: one more line behind swscale_unscaled.c:2698-2704 */
}
