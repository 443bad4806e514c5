/*
 * integration/avutil_hwcontext_hip.c — libavutil/hwcontext_hip.c of the FFmpeg-side patch: the `hip` AVHWDeviceType.
 *
 * Frames that live in HBM between filters: an AVHWFramesContext whose pool hands out device allocations of
 * libffhip (ffhip_malloc), plane pointers in AVFrame.data[] exactly as hwcontext_cuda lays them out (libavutil/hwcontext_cuda.c:
 * 156-190: one allocation per frame, planes at av_image_fill_pointers() offsets of the aligned size), and transfer_data_to / _from as
 * pitched copies on the device context's stream.  A consumer — the hip swscale backend, a hip decoder — takes AVFrame.data[] /
 * linesize[] as the device pointers and strides of ffhip_sws_scale_batch_dev() & co, and the frames never cross PCIe in between.
 *
 * The shape follows the HWContextType of libavutil/hwcontext_internal.h:32-100.  The type has enum values of its own
 * (AV_HWDEVICE_TYPE_HIP, AV_PIX_FMT_HIP: avutil_hwcontext_hip.h) and its rows in hw_table[] / hw_type_names[]
 * (libavutil/hwcontext.c:32-100) and av_pix_fmt_descriptors[] (libavutil/pixdesc.c) — added by wrappers that compile the reference
 * files UNCHANGED, where they lie (avutil_hwcontext_table_hip.c, avutil_pixdesc_hip.c): every generic entry point
 * (av_hwdevice_find_type_by_name("hip"), av_hwdevice_ctx_create, av_hwframe_ctx_init, av_hwframe_get_buffer,
 * av_hwframe_transfer_data) dispatches into this file.
 */
#include <string.h>

#include "libavutil/buffer.h"
#include "libavutil/common.h"
#include "libavutil/hwcontext.h"
#include "libavutil/hwcontext_internal.h"
#include "libavutil/imgutils.h"
#include "libavutil/mem.h"
#include "libavutil/pixdesc.h"

#include "ffhip.h"
#include "avutil_hwcontext_hip.h"

#define HIP_FRAME_ALIGNMENT 256 /* rows and planes start on 256-byte boundaries (the kernels' widest accesses are 16 bytes) */

typedef struct HIPFramesContext {
    int shift_width, shift_height;
} HIPFramesContext;

static const enum AVPixelFormat supported_formats[] = {
    AV_PIX_FMT_NV12, AV_PIX_FMT_YUV420P, AV_PIX_FMT_YUV422P, AV_PIX_FMT_YUV444P, AV_PIX_FMT_P010, AV_PIX_FMT_P016, AV_PIX_FMT_YUV420P10,
    AV_PIX_FMT_YUV444P10, AV_PIX_FMT_YUV444P16, AV_PIX_FMT_RGB24, AV_PIX_FMT_BGR24, AV_PIX_FMT_0RGB32, AV_PIX_FMT_0BGR32, AV_PIX_FMT_RGB32,
    AV_PIX_FMT_BGR32,
};

static int hip_err(int r) { return r == FFHIP_ENOMEM ? AVERROR(ENOMEM) : r == FFHIP_ENOSYS ? AVERROR(ENOSYS) : r == FFHIP_EINVAL ? AVERROR(EINVAL) : AVERROR_EXTERNAL; }

static int hip_frames_get_constraints(AVHWDeviceContext *ctx, const void *hwconfig, AVHWFramesConstraints *constraints)
{
    constraints->valid_sw_formats = av_malloc_array(FF_ARRAY_ELEMS(supported_formats) + 1, sizeof(*constraints->valid_sw_formats));
    if (!constraints->valid_sw_formats)
        return AVERROR(ENOMEM);
    for (int i = 0; i < FF_ARRAY_ELEMS(supported_formats); i++)
        constraints->valid_sw_formats[i] = supported_formats[i];
    constraints->valid_sw_formats[FF_ARRAY_ELEMS(supported_formats)] = AV_PIX_FMT_NONE;
    constraints->valid_hw_formats = av_malloc_array(2, sizeof(*constraints->valid_hw_formats));
    if (!constraints->valid_hw_formats)
        return AVERROR(ENOMEM);
    constraints->valid_hw_formats[0] = FFHIP_HW_PIX_FMT;
    constraints->valid_hw_formats[1] = AV_PIX_FMT_NONE;
    return 0;
}

static void hip_buffer_free(void *opaque, uint8_t *data)
{
    AVHWFramesContext *ctx = opaque;
    AVHIPDeviceContext *hw = ctx->device_ctx->hwctx;
    int prev;
    if (ffhip_device_push(hw->device, &prev) >= 0) { /* the thread that drops the last reference stays bound as it was */
        ffhip_free(data);
        ffhip_device_pop(prev);
    }
}

static AVBufferRef *hip_pool_alloc(void *opaque, size_t size)
{
    AVHWFramesContext *ctx = opaque;
    AVHIPDeviceContext *hw = ctx->device_ctx->hwctx;
    AVBufferRef *ret = NULL;
    void *data = NULL;
    int prev, r;
    if (ffhip_device_push(hw->device, &prev) < 0)
        return NULL;
    r = ffhip_malloc(&data, size);
    ffhip_device_pop(prev);
    if (r < 0)
        return NULL;
    ret = av_buffer_create(data, size, hip_buffer_free, ctx, 0);
    if (!ret)
        ffhip_free(data);
    return ret;
}

static int hip_frames_init(AVHWFramesContext *ctx)
{
    HIPFramesContext *priv = ctx->hwctx;
    int i;
    for (i = 0; i < FF_ARRAY_ELEMS(supported_formats); i++)
        if (ctx->sw_format == supported_formats[i])
            break;
    if (i == FF_ARRAY_ELEMS(supported_formats)) {
        av_log(ctx, AV_LOG_ERROR, "Pixel format '%s' is not supported\n", av_get_pix_fmt_name(ctx->sw_format));
        return AVERROR(ENOSYS);
    }
    av_pix_fmt_get_chroma_sub_sample(ctx->sw_format, &priv->shift_width, &priv->shift_height);
    if (!ctx->pool) {
        int size = av_image_get_buffer_size(ctx->sw_format, FFALIGN(ctx->width, HIP_FRAME_ALIGNMENT), ctx->height, HIP_FRAME_ALIGNMENT);
        if (size < 0)
            return size;
        ffhwframesctx(ctx)->pool_internal = av_buffer_pool_init2(size, ctx, hip_pool_alloc, NULL);
        if (!ffhwframesctx(ctx)->pool_internal)
            return AVERROR(ENOMEM);
    }
    return 0;
}

static int hip_get_buffer(AVHWFramesContext *ctx, AVFrame *frame)
{
    int res;
    frame->buf[0] = av_buffer_pool_get(ctx->pool);
    if (!frame->buf[0])
        return AVERROR(ENOMEM);
    res = av_image_fill_arrays(frame->data, frame->linesize, frame->buf[0]->data, ctx->sw_format, FFALIGN(ctx->width, HIP_FRAME_ALIGNMENT),
                               ctx->height, HIP_FRAME_ALIGNMENT);
    if (res < 0)
        return res;
    frame->format = FFHIP_HW_PIX_FMT;
    frame->width  = ctx->width;
    frame->height = ctx->height;
    return 0;
}

static int hip_transfer_get_formats(AVHWFramesContext *ctx, enum AVHWFrameTransferDirection dir, enum AVPixelFormat **formats)
{
    enum AVPixelFormat *fmts = av_malloc_array(2, sizeof(*fmts));
    if (!fmts)
        return AVERROR(ENOMEM);
    fmts[0] = ctx->sw_format;
    fmts[1] = AV_PIX_FMT_NONE;
    *formats = fmts;
    return 0;
}

static int hip_transfer(AVHWFramesContext *ctx, AVFrame *dst, const AVFrame *src, int to_device)
{
    HIPFramesContext *priv = ctx->hwctx;
    AVHIPDeviceContext *hw = ctx->device_ctx->hwctx;
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(ctx->sw_format);
    int r, prev;
    if ((r = ffhip_device_push(hw->device, &prev)) < 0)
        return hip_err(r);
    /* The consumers of hip frames that take no stream (the swscale backend's SwsOpFunc, swscale_hw_hip.c; the graph's legacy pass,
     * swscale_graph_hip.c) launch on the legacy default stream, and hw->stream is a non-blocking one: the two are unordered unless
     * said otherwise.  A transfer therefore starts behind everything the default stream has queued (a download reads what a
     * conversion wrote) and, when it does not wait itself, the default stream continues behind it (a conversion reads an upload). */
    if ((r = ffhip_stream_order(NULL, hw->stream)) < 0) {
        ffhip_device_pop(prev);
        return hip_err(r);
    }
    for (int i = 0; i < FF_ARRAY_ELEMS(src->data) && src->data[i]; i++) {
        const int h = src->height >> ((i == 0 || i == 3) ? 0 : priv->shift_height);
        const int bw = av_image_get_linesize(ctx->sw_format, src->width, i); /* bytes of a row of this plane */
        if (bw < 0) {
            ffhip_device_pop(prev);
            return bw;
        }
        r = to_device ? ffhip_memcpy2d_h2d_async(dst->data[i], dst->linesize[i], src->data[i], src->linesize[i], bw, h, hw->stream)
                      : ffhip_memcpy2d_d2h_async(dst->data[i], dst->linesize[i], src->data[i], src->linesize[i], bw, h, hw->stream);
        if (r < 0) {
            ffhip_device_pop(prev);
            return hip_err(r);
        }
    }
    (void)desc;
    if (!to_device || !hw->async_upload) /* a download must have landed before the caller reads it */
        r = ffhip_stream_synchronize(hw->stream);
    else
        r = ffhip_stream_order(hw->stream, NULL);
    ffhip_device_pop(prev);
    return r < 0 ? hip_err(r) : 0;
}
static int hip_transfer_data_to(AVHWFramesContext *ctx, AVFrame *dst, const AVFrame *src) { return hip_transfer(ctx, dst, src, 1); }
static int hip_transfer_data_from(AVHWFramesContext *ctx, AVFrame *dst, const AVFrame *src) { return hip_transfer(ctx, dst, src, 0); }

static void hip_device_uninit(AVHWDeviceContext *device_ctx)
{
    AVHIPDeviceContext *hw = device_ctx->hwctx;
    if (hw->stream && hw->owns_stream && ffhip_set_device(hw->device) >= 0)
        ffhip_stream_destroy(hw->stream);
    hw->stream = NULL;
}

static int hip_device_init(AVHWDeviceContext *ctx)
{
    AVHIPDeviceContext *hw = ctx->hwctx;
    int r;
    if (hw->device < 0 || hw->device >= ffhip_device_count())
        return AVERROR(ENODEV);
    if ((r = ffhip_set_device(hw->device)) < 0)
        return hip_err(r);
    if (!hw->stream) { /* a caller that brings its own stream (av_hwdevice_ctx_alloc + init) keeps it */
        if ((r = ffhip_stream_create(&hw->stream)) < 0)
            return hip_err(r);
        hw->owns_stream = 1;
    }
    return 0;
}

static int hip_device_create(AVHWDeviceContext *device_ctx, const char *device, AVDictionary *opts, int flags)
{
    AVHIPDeviceContext *hw = device_ctx->hwctx;
    hw->device = device ? (int)strtol(device, NULL, 0) : 0; /* "-init_hw_device hip:1": the ordinal, as for cuda */
    if (hw->device < 0 || hw->device >= ffhip_device_count()) {
        av_log(device_ctx, AV_LOG_ERROR, "No HIP device %d (%d usable)\n", hw->device, ffhip_device_count());
        return AVERROR(ENODEV);
    }
    return 0;
}

const HWContextType ff_hwcontext_type_hip = {
    .type                   = FFHIP_HWDEVICE_TYPE,
    .name                   = "HIP",
    .device_hwctx_size      = sizeof(AVHIPDeviceContext),
    .frames_hwctx_size      = sizeof(HIPFramesContext),
    .device_create          = hip_device_create,
    .device_init            = hip_device_init,
    .device_uninit          = hip_device_uninit,
    .frames_get_constraints = hip_frames_get_constraints,
    .frames_init            = hip_frames_init,
    .frames_get_buffer      = hip_get_buffer,
    .transfer_get_formats   = hip_transfer_get_formats,
    .transfer_data_to       = hip_transfer_data_to,
    .transfer_data_from     = hip_transfer_data_from,
    .pix_fmts               = (const enum AVPixelFormat[]){ FFHIP_HW_PIX_FMT, AV_PIX_FMT_NONE },
};
