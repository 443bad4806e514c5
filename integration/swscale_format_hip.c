/*
 * integration/swscale_format_hip.c — the one-line change of libswscale/format.c for hip frames.
 *
 * sws_test_hw_format() (libswscale/format.c:616-626) lists the hardware formats sws_scale_frame accepts: AV_PIX_FMT_NONE and,
 * under CONFIG_VULKAN, AV_PIX_FMT_VULKAN; the patch adds `case AV_PIX_FMT_HIP: return 1;`.  The reference file is compiled
 * unchanged: the wrapper switches that one `#if CONFIG_VULKAN` block on (format.c has no other) and lets its case label mean the
 * hip frames' format.
 */
#include "config.h"
#include "libavutil/pixfmt.h"
#include "avutil_hwcontext_hip.h"
#undef CONFIG_VULKAN
#define CONFIG_VULKAN 1
#define AV_PIX_FMT_VULKAN FFHIP_HW_PIX_FMT
#include "libswscale/format.c"
