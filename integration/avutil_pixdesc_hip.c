/*
 * integration/avutil_pixdesc_hip.c — libavutil/pixdesc.c with the AV_PIX_FMT_HIP row in av_pix_fmt_descriptors[].
 *
 * The patch adds `[AV_PIX_FMT_HIP] = { .name = "hip", .flags = AV_PIX_FMT_FLAG_HWACCEL },` next to the CUDA row
 * (libavutil/pixdesc.c:2277-2280) and the enumerator before AV_PIX_FMT_NB.  The reference file is compiled unchanged, where it lies:
 * the table and the three range checks (:203, :3371, :3382, :3462) see an AV_PIX_FMT_NB one larger, and AV_PIX_FMT_CUDA — named once
 * in the file, as its row's designator — expands to the hip row followed by itself.  av_pix_fmt_desc_get(AV_PIX_FMT_HIP),
 * av_get_pix_fmt("hip") and av_get_pix_fmt_name() then work as for any hardware format.
 */
#include "libavutil/pixfmt.h"
#include "libavutil/hwcontext.h"
#include "avutil_hwcontext_hip.h"
#define AV_PIX_FMT_NB (FFHIP_PIX_FMT_NB_REF + 1)
#define AV_PIX_FMT_CUDA FFHIP_PIX_FMT_NB_REF] = { .name = "hip", .flags = AV_PIX_FMT_FLAG_HWACCEL }, [AV_PIX_FMT_CUDA
#include "libavutil/pixdesc.c"
