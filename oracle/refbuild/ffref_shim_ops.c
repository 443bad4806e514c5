/*
 * oracle/refbuild/ffref_shim_ops.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * The `hip` SwsOpBackend as a maintainer would add it to libswscale, living inside the reference build so that
 * the reference's OWN machinery drives it: ff_sws_op_list_generate() builds the real SwsOpList, the optimizer and the splitter
 * (ops_dispatch.c:717-770) cut it, ff_sws_ops_translate() lowers it, and op_pass_run() (ops_dispatch.c:403-500) calls the compiled
 * function with its tails and padding.  The five functions behind the backend are BOUND at run time (ffref_sws_hip_bind): tests hand
 * in libffhip's entry points (GPU) or the oracle's restatement of them (CPU pin) — libffref.so links neither.
 *
 * The reference's backend list (libswscale/ops.c:42-54) is a const array; the recipe compiles that file with its array renamed
 * (Makefile: -Dff_sws_op_backends=ffref_unused_op_backends) and this file supplies the list with `hip` in front of `c`.
 */
#include <string.h>

#include "libavutil/frame.h"
#include "libavutil/mem.h"
#include "libavutil/pixdesc.h"
#include "libavutil/refstruct.h"
#include "libswscale/swscale.h"
#include "libswscale/filters.h"
#include "libswscale/ops.h"
#include "libswscale/ops_dispatch.h"
#include "libswscale/ops_internal.h"
#include "libswscale/uops.h"

#include "../../include/ffhip.h"
#include "ffref.h"

/* ---- the layout contract of include/ffhip.h's SwsOpBackend section ---- */
#define SAME(a, b, m) static_assert(offsetof(a, m) == offsetof(b, m), #a "." #m)
static_assert(sizeof(FFHipSwsPixel) == sizeof(SwsPixel), "SwsPixel");
static_assert(sizeof(FFHipSwsFilterWeights) == sizeof(SwsFilterWeights), "SwsFilterWeights");
SAME(FFHipSwsFilterWeights, SwsFilterWeights, filter_size); SAME(FFHipSwsFilterWeights, SwsFilterWeights, weights);
SAME(FFHipSwsFilterWeights, SwsFilterWeights, num_weights); SAME(FFHipSwsFilterWeights, SwsFilterWeights, offsets);
SAME(FFHipSwsFilterWeights, SwsFilterWeights, src_size);    SAME(FFHipSwsFilterWeights, SwsFilterWeights, dst_size);
static_assert(sizeof(FFHipSwsUOp) == sizeof(SwsUOp), "SwsUOp");
SAME(FFHipSwsUOp, SwsUOp, type); SAME(FFHipSwsUOp, SwsUOp, uop); SAME(FFHipSwsUOp, SwsUOp, mask); SAME(FFHipSwsUOp, SwsUOp, par);
SAME(FFHipSwsUOp, SwsUOp, data);
static_assert(sizeof(FFHipSwsUOpParams) == sizeof(SwsUOpParams), "SwsUOpParams");
static_assert(offsetof(FFHipSwsUOpParams, move.dst) == offsetof(SwsUOpParams, move.dst) &&
              offsetof(FFHipSwsUOpParams, move.src) == offsetof(SwsUOpParams, move.src), "SwsMoveUOp");
static_assert(offsetof(FFHipSwsUOpParams, lin.zero) == offsetof(SwsUOpParams, lin.zero), "SwsLinearUOp");
static_assert(offsetof(FFHipSwsUOpParams, dither.size_log2) == offsetof(SwsUOpParams, dither.size_log2), "SwsDitherUOp");
static_assert(offsetof(FFHipSwsUOpParams, clear.zero) == offsetof(SwsUOpParams, clear.zero), "SwsClearUOp");
static_assert(sizeof(FFHipSwsOpExec) == sizeof(SwsOpExec), "SwsOpExec");
SAME(FFHipSwsOpExec, SwsOpExec, in_bump); SAME(FFHipSwsOpExec, SwsOpExec, width); SAME(FFHipSwsOpExec, SwsOpExec, block_size_in);
SAME(FFHipSwsOpExec, SwsOpExec, in_bump_y); SAME(FFHipSwsOpExec, SwsOpExec, in_offset_x);
static_assert(FFHIP_SWS_PIXEL_U8 == SWS_PIXEL_U8 && FFHIP_SWS_PIXEL_F32 == SWS_PIXEL_F32, "SwsPixelType");
static_assert(FFHIP_SWS_UOP_READ_PLANAR == SWS_UOP_READ_PLANAR && FFHIP_SWS_UOP_READ_PLANAR_FH == SWS_UOP_READ_PLANAR_FH &&
              FFHIP_SWS_UOP_READ_PLANAR_FV == SWS_UOP_READ_PLANAR_FV && FFHIP_SWS_UOP_READ_PLANAR_FV_FMA == SWS_UOP_READ_PLANAR_FV_FMA &&
              FFHIP_SWS_UOP_READ_PACKED == SWS_UOP_READ_PACKED && FFHIP_SWS_UOP_READ_NIBBLE == SWS_UOP_READ_NIBBLE &&
              FFHIP_SWS_UOP_READ_BIT == SWS_UOP_READ_BIT && FFHIP_SWS_UOP_READ_PALETTE == SWS_UOP_READ_PALETTE &&
              FFHIP_SWS_UOP_WRITE_PLANAR == SWS_UOP_WRITE_PLANAR && FFHIP_SWS_UOP_WRITE_PACKED == SWS_UOP_WRITE_PACKED &&
              FFHIP_SWS_UOP_WRITE_NIBBLE == SWS_UOP_WRITE_NIBBLE && FFHIP_SWS_UOP_WRITE_BIT == SWS_UOP_WRITE_BIT &&
              FFHIP_SWS_UOP_RW_SHUFFLE == SWS_UOP_RW_SHUFFLE && FFHIP_SWS_UOP_PERMUTE == SWS_UOP_PERMUTE &&
              FFHIP_SWS_UOP_COPY == SWS_UOP_COPY && FFHIP_SWS_UOP_SWAP_BYTES == SWS_UOP_SWAP_BYTES &&
              FFHIP_SWS_UOP_EXPAND_BIT == SWS_UOP_EXPAND_BIT && FFHIP_SWS_UOP_EXPAND_PAIR == SWS_UOP_EXPAND_PAIR &&
              FFHIP_SWS_UOP_EXPAND_QUAD == SWS_UOP_EXPAND_QUAD && FFHIP_SWS_UOP_TO_U8 == SWS_UOP_TO_U8 &&
              FFHIP_SWS_UOP_TO_U16 == SWS_UOP_TO_U16 && FFHIP_SWS_UOP_TO_U32 == SWS_UOP_TO_U32 && FFHIP_SWS_UOP_TO_F32 == SWS_UOP_TO_F32 &&
              FFHIP_SWS_UOP_SCALE == SWS_UOP_SCALE && FFHIP_SWS_UOP_ADD == SWS_UOP_ADD && FFHIP_SWS_UOP_MIN == SWS_UOP_MIN &&
              FFHIP_SWS_UOP_MAX == SWS_UOP_MAX && FFHIP_SWS_UOP_UNPACK == SWS_UOP_UNPACK && FFHIP_SWS_UOP_PACK == SWS_UOP_PACK &&
              FFHIP_SWS_UOP_LSHIFT == SWS_UOP_LSHIFT && FFHIP_SWS_UOP_RSHIFT == SWS_UOP_RSHIFT && FFHIP_SWS_UOP_CLEAR == SWS_UOP_CLEAR &&
              FFHIP_SWS_UOP_LINEAR == SWS_UOP_LINEAR && FFHIP_SWS_UOP_LINEAR_FMA == SWS_UOP_LINEAR_FMA &&
              FFHIP_SWS_UOP_DITHER == SWS_UOP_DITHER && FFHIP_SWS_UOP_LUT_3D == SWS_UOP_LUT_3D &&
              FFHIP_SWS_UOP_TYPE_NB == SWS_UOP_TYPE_NB, "SwsUOpType");
static_assert(FFHIP_SWS_FILTER_SCALE == SWS_FILTER_SCALE, "SWS_FILTER_SCALE");
static_assert(FFHIP_ENOTSUP == AVERROR(ENOTSUP), "ENOTSUP");

/* ---- the backend ---- */
#define SWS_BACKEND_HIP (1 << 6)      /* the next free SwsBackend bit (swscale.h:112-127) */

static struct {
    int  (*compile)(const FFHipSwsUOp *, int, void **);
    void (*free)(void **);
    int  (*block_size)(const void *);
    FFHipSwsOpFunc func;
    void (*set_fallback)(void *, FFHipSwsOpFunc, const void *);
} hip;
static long hip_lists, hip_notsup;

extern const SwsOpBackend backend_c;
extern const SwsOpBackend backend_murder;

/* SwsCompiledOp.priv is what `func` receives, i.e. the bound library's object; the C function compiled beside it (the fallback of
 * the void face) is remembered here until the pass is freed */
static struct { void *u; SwsCompiledOp c; } hip_pairs[4096];

static void hip_free_priv(void *priv)
{
    for (size_t i = 0; i < FF_ARRAY_ELEMS(hip_pairs); i++) {
        if (hip_pairs[i].u == priv) {
            ff_sws_compiled_op_unref(&hip_pairs[i].c);
            hip_pairs[i].u = NULL;
            break;
        }
    }
    hip.free(&priv);
}

static int compile_uops_hip(SwsContext *ctx, const SwsUOpList *uops, SwsCompiledOp *out)
{
    if (!hip.compile)
        return AVERROR(ENOTSUP);
    void *u = NULL;
    int ret = hip.compile((const FFHipSwsUOp *) uops->ops, uops->num_ops, &u);
    if (ret < 0) {
        hip_notsup += ret == AVERROR(ENOTSUP);
        return ret;
    }
    hip_lists++;
    for (size_t i = 0; i < FF_ARRAY_ELEMS(hip_pairs); i++) {
        if (!hip_pairs[i].u) {
            if (backend_c.compile_uops(ctx, uops, &hip_pairs[i].c) >= 0) {
                hip_pairs[i].u = u;
                if (hip.set_fallback)
                    hip.set_fallback(u, (FFHipSwsOpFunc) hip_pairs[i].c.func, hip_pairs[i].c.priv);
            }
            break;
        }
    }
    *out = (SwsCompiledOp) {
        .func        = (SwsOpFunc) hip.func,
        .priv        = u,
        .free        = hip_free_priv,
        .slice_align = 1,
        .block_size  = hip.block_size(u),
    };
    return 0;
}

static int compile_hip(SwsContext *ctx, const SwsOpList *ops, SwsCompiledOp *out)
{
    if (!hip.compile)
        return AVERROR(ENOTSUP);
    SwsUOpList *uops = ff_sws_uop_list_alloc();
    if (!uops)
        return AVERROR(ENOMEM);
    int ret = ff_sws_ops_translate(ctx, ops, 0 /* no FMA, no shuffles: backend_c's own lowering */, uops);
    if (ret >= 0)
        ret = compile_uops_hip(ctx, uops, out);
    ff_sws_uop_list_free(&uops);
    return ret;
}

const SwsOpBackend backend_hip = {
    .name         = "hip",
    .flags        = SWS_BACKEND_HIP,
    .compile      = compile_hip,
    .compile_uops = compile_uops_hip,
    .hw_format    = AV_PIX_FMT_NONE,
};

const SwsOpBackend *const ff_sws_op_backends[] = { &backend_murder, &backend_hip, &backend_c, NULL };

int ffref_sws_hip_bind(void *compile, void *free_, void *block_size, void *func, void *set_fallback)
{
    hip.compile = compile;
    hip.free = free_;
    hip.block_size = block_size;
    hip.func = func;
    hip.set_fallback = set_fallback;
    return 0;
}

long ffref_sws_hip_count(int what)
{
    if (what < 0)
        hip_lists = hip_notsup = 0;
    return what == 1 ? hip_notsup : hip_lists;
}

/* sws_scale_frame() (swscale.c:1405) on caller-owned planes, with the set of op backends restricted to `backends`:
 * SWS_BACKEND_C | SWS_BACKEND_MEMCPY for the reference's answer, FFREF_SWS_BACKEND_HIP | SWS_BACKEND_MEMCPY for the bound backend */
int ffref_sws_frame_convert(int backends, int flags, int scaler, int dither, int threads,
                            int sw, int sh, int sfmt, const uint8_t *const src[4], const int sstride[4],
                            int dw, int dh, int dfmt, uint8_t *const dst[4], const int dstride[4])
{
    SwsContext *ctx = sws_alloc_context();
    AVFrame *s = av_frame_alloc(), *d = av_frame_alloc();
    int ret = AVERROR(ENOMEM);
    if (!ctx || !s || !d)
        goto end;
    ctx->flags    = flags | SWS_UNSTABLE;
    ctx->backends = backends;
    ctx->threads  = threads;
    if (scaler >= 0)
        ctx->scaler = scaler;
    if (dither >= 0)
        ctx->dither = dither;
    s->format = sfmt; s->width = sw; s->height = sh;
    d->format = dfmt; d->width = dw; d->height = dh;
    for (int i = 0; i < 4; i++) {
        s->data[i] = (uint8_t *) src[i]; s->linesize[i] = sstride[i];
        d->data[i] = dst[i];             d->linesize[i] = dstride[i];
    }
    ret = sws_scale_frame(ctx, d, s);
end:
    av_frame_free(&s);
    av_frame_free(&d);
    sws_free_context(&ctx);
    return ret;
}

/* ---- micro-op level: what tests/checkasm/sw_ops.c:147-290 does with the reference backend ---- */
int ffref_sws_uops_run_c(const FFHipSwsUOp *uops, int n, const FFHipSwsOpExec *exec, int x_start, int y_start, int x_end, int y_end)
{
    SwsContext *ctx = sws_alloc_context();
    if (!ctx)
        return AVERROR(ENOMEM);
    ctx->flags = SWS_BITEXACT;
    /* backend_c takes references on the constant data of a list (setup_filter_h, setup_dither: av_refstruct_ref), so the list it
     * sees carries refstruct copies of the caller's plain arrays */
    SwsUOp ops[16];
    SwsFilterWeights kern[16];
    void *owned[16] = {0};
    if (n > 16) {
        sws_free_context(&ctx);
        return AVERROR(EINVAL);
    }
    memcpy(ops, uops, n * sizeof(*ops));
    for (int i = 0; i < n; i++) {
        if (ops[i].uop == SWS_UOP_READ_PLANAR_FH || ops[i].uop == SWS_UOP_READ_PLANAR_FV) {
            kern[i] = *ops[i].data.kernel;
            const size_t bytes = sizeof(int) * kern[i].dst_size * kern[i].filter_size;
            owned[i] = av_refstruct_allocz(bytes);
            memcpy(owned[i], kern[i].weights, bytes);
            kern[i].weights = owned[i];
            kern[i].num_weights = (size_t) kern[i].dst_size * kern[i].filter_size;
            ops[i].data.kernel = &kern[i];
        } else if (ops[i].uop == SWS_UOP_DITHER) {
            const size_t bytes = sizeof(SwsPixel) * (1 << ops[i].par.dither.size_log2) * ff_sws_dither_height(&ops[i].par.dither);
            owned[i] = av_refstruct_allocz(bytes);
            memcpy(owned[i], ops[i].data.ptr, bytes);
            ops[i].data.ptr = owned[i];
        }
    }
    SwsUOpList list = { .ops = ops, .num_ops = n, .planes_in = SWS_COMP_ALL, .planes_out = SWS_COMP_ALL };
    for (int i = 0; i < n; i++)
        list.pixel_size_max = FFMAX(list.pixel_size_max, ff_sws_pixel_type_size(list.ops[i].type));
    SwsCompiledOp comp = {0};
    int ret = backend_c.compile_uops(ctx, &list, &comp);
    if (ret >= 0) {
        if (x_start % comp.block_size || x_end % comp.block_size)
            ret = AVERROR(EINVAL);
        else
            comp.func((const SwsOpExec *) exec, comp.priv, x_start / comp.block_size, y_start, x_end / comp.block_size, y_end);
        ff_sws_compiled_op_unref(&comp);
    }
    for (int i = 0; i < n; i++)
        av_refstruct_unref(&owned[i]);
    sws_free_context(&ctx);
    return ret < 0 ? ret : 0;
}

/* ff_sws_filter_generate() (filters.c): the kernel of a scaler for src_size -> dst_size; weights[dst_size * (*filter_size)] */
int ffref_sws_filter_generate(int scaler, int src_size, int dst_size, int *filter_size, int *weights, int weights_cap, int *offsets)
{
    SwsFilterParams par = {
        .scaler = scaler, .scaler_params = { SWS_PARAM_DEFAULT, SWS_PARAM_DEFAULT }, .src_size = src_size, .dst_size = dst_size,
    };
    SwsFilterWeights *k = NULL;
    int ret = ff_sws_filter_generate(NULL, &par, &k);
    if (ret < 0)
        return ret;
    *filter_size = k->filter_size;
    if ((int) k->num_weights > weights_cap) {
        ret = AVERROR(ENOMEM);
    } else {
        memcpy(weights, k->weights, k->num_weights * sizeof(int));
        memcpy(offsets, k->offsets, dst_size * sizeof(int));
    }
    av_refstruct_unref(&k);
    return ret < 0 ? ret : 0;
}

/* the micro-op lists the reference cuts a conversion into (generate + optimize + split + translate), described as text: one line per
 * micro-op (ff_sws_uop_name), lists separated by an empty line.  Returns the number of lists, < 0 on error. */
static char *desc_buf;
static int desc_cap, desc_len, desc_lists;
static int describe(SwsContext *ctx, const SwsOpList *ops, SwsCompiledOp *out)
{
    SwsUOpList *uops = ff_sws_uop_list_alloc();
    int ret = ff_sws_ops_translate(ctx, ops, 0, uops);
    if (ret >= 0) {
        for (int i = 0; i < uops->num_ops; i++) {
            char name[SWS_UOP_NAME_MAX];
            ff_sws_uop_name(&uops->ops[i], name);
            desc_len += snprintf(desc_buf + desc_len, FFMAX(desc_cap - desc_len, 0), "%s\n", name);
        }
        desc_len += snprintf(desc_buf + desc_len, FFMAX(desc_cap - desc_len, 0), "\n");
        desc_lists++;
        *out = (SwsCompiledOp) {0};
    }
    ff_sws_uop_list_free(&uops);
    return ret;
}
static const SwsOpBackend backend_describe = { .name = "describe", .compile = describe };

int ffref_sws_describe_uops(int flags, int scaler, int sw, int sh, int sfmt, int dw, int dh, int dfmt, char *buf, int cap)
{
    SwsGraph *graph = ff_sws_graph_alloc();
    SwsContext *ctx = sws_alloc_context();
    SwsOpList *ops = NULL;
    int ret = AVERROR(ENOMEM);
    if (!graph || !ctx)
        goto end;
    graph->ctx = ctx;
    ctx->flags = flags;
    if (scaler >= 0)
        ctx->scaler = scaler;
    SwsFormat src, dst;
    ff_fmt_from_pixfmt(sfmt, &src);
    ff_fmt_from_pixfmt(dfmt, &dst);
    bool incomplete = ff_infer_colors(&src.color, &dst.color);
    src.width = sw; src.height = sh; dst.width = dw; dst.height = dh;
    desc_buf = buf; desc_cap = cap; desc_len = 0; desc_lists = 0;
    if (cap > 0)
        buf[0] = 0;
    ret = ff_sws_op_list_generate(ctx, &src, &dst, NULL, &ops, &incomplete);
    if (ret >= 0)
        ret = ff_sws_compile_pass(graph, &backend_describe, &ops, SWS_OP_FLAG_OPTIMIZE | SWS_OP_FLAG_DRY_RUN | SWS_OP_FLAG_SPLIT_MEMCPY,
                                  NULL, NULL);
end:
    sws_free_context(&ctx);
    ff_sws_graph_free(&graph);
    return ret < 0 ? ret : desc_lists;
}

/* ---- every micro-op instance backend_c implements (libswscale/uops_macros.h through the macros tests/checkasm/sw_ops.c:641-708 uses) ---- */
#include "libswscale/uops_macros.h"
#define ENTRY(ARG, NAME, ...) { #NAME, { __VA_ARGS__ } },
#define ALL(UOP) SWS_FOR_STRUCT(U8, UOP, ENTRY, 0) SWS_FOR_STRUCT(U16, UOP, ENTRY, 0) SWS_FOR_STRUCT(U32, UOP, ENTRY, 0) \
                 SWS_FOR_STRUCT(F32, UOP, ENTRY, 0)
static const struct { const char *name; SwsUOp uop; } uop_instances[] = {
    ALL(READ_PLANAR) ALL(READ_PLANAR_FH) ALL(READ_PLANAR_FV) ALL(READ_PACKED) ALL(READ_NIBBLE) ALL(READ_BIT) ALL(READ_PALETTE)
    ALL(WRITE_PLANAR) ALL(WRITE_PACKED) ALL(WRITE_NIBBLE) ALL(WRITE_BIT) ALL(PERMUTE) ALL(COPY) ALL(SWAP_BYTES) ALL(EXPAND_BIT)
    ALL(EXPAND_PAIR) ALL(EXPAND_QUAD) ALL(TO_U8) ALL(TO_U16) ALL(TO_U32) ALL(TO_F32) ALL(SCALE) ALL(ADD) ALL(MIN) ALL(MAX)
    ALL(UNPACK) ALL(PACK) ALL(LSHIFT) ALL(RSHIFT) ALL(CLEAR) ALL(LINEAR) ALL(DITHER) ALL(LUT_3D)
};

int ffref_sws_uop_instance(int idx, FFHipSwsUOp *out, char *name, int cap)
{
    const int n = FF_ARRAY_ELEMS(uop_instances);
    if (idx >= 0 && idx < n) {
        memcpy(out, &uop_instances[idx].uop, sizeof(*out));
        snprintf(name, cap, "%s", uop_instances[idx].name);
    }
    return n;
}

/* ---- pixel format helpers for the tests (libavutil/pixdesc.c, imgutils.c) ---- */
#include "libavutil/imgutils.h"

/* linesizes (aligned up to `align`) and plane heights of a w x h picture; returns the number of planes */
int ffref_image_layout(int fmt, int w, int h, int align, int linesize[4], int lines[4])
{
    ptrdiff_t ls[4];
    size_t sizes[4];
    int ret = av_image_fill_linesizes(linesize, fmt, w);
    if (ret < 0)
        return ret;
    int n = 0;
    for (int i = 0; i < 4; i++) {
        linesize[i] = FFALIGN(linesize[i], align);
        ls[i] = linesize[i];
    }
    if ((ret = av_image_fill_plane_sizes(sizes, fmt, h, ls)) < 0)
        return ret;
    for (int i = 0; i < 4; i++) {
        lines[i] = linesize[i] ? (int) (sizes[i] / linesize[i]) : 0;
        n += sizes[i] > 0;
    }
    return n;
}
