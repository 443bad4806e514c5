/*
 * integration/swscale_frame_hip.c — the one-line change of libswscale/swscale.c for hip frames.
 *
 * sws_frame_setup() refuses hardware frames of any device type but Vulkan (libswscale/swscale.c:1526-1530); the patch adds
 * `&& dev_ctx->type != AV_HWDEVICE_TYPE_HIP` to that test.  The reference file is compiled unchanged, where it lies: this wrapper
 * makes the name in that test mean the hip device type for the duration of the include (the enum itself is defined before the
 * macro; swscale.c uses the name nowhere else — the recipe checks).
 */
#include "config.h"
#include "libavutil/hwcontext.h"
#include "avutil_hwcontext_hip.h"
#define AV_HWDEVICE_TYPE_VULKAN FFHIP_HWDEVICE_TYPE
/* ... and sws_scale_frame() itself notes, for the duration of the call, that it runs between two frames of the hip device: the graph
 * it builds on this thread (integration/swscale_graph_hip.c) then routes the legacy scaler's passes to the device */
#define sws_scale_frame ffhip_ref_sws_scale_frame
#include "libswscale/swscale.c"
#undef sws_scale_frame
#undef AV_HWDEVICE_TYPE_VULKAN

_Thread_local int ffhip_integration_hip_frames;

int sws_scale_frame(SwsContext *sws, AVFrame *dst, const AVFrame *src);
int sws_scale_frame(SwsContext *sws, AVFrame *dst, const AVFrame *src)
{
    const int prev = ffhip_integration_hip_frames;
    int ret;
    ffhip_integration_hip_frames = src && dst && src->hw_frames_ctx && dst->hw_frames_ctx && src->format == FFHIP_HW_PIX_FMT &&
                                   dst->format == FFHIP_HW_PIX_FMT;
    ret = ffhip_ref_sws_scale_frame(sws, dst, src);
    ffhip_integration_hip_frames = prev;
    return ret;
}
