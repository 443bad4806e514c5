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
#include "libswscale/swscale.c"
