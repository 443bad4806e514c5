/*
 * integration/avutil_hwcontext_table_hip.c — libavutil/hwcontext.c with the `hip` type in hw_table[] and hw_type_names[].
 *
 * hw_table[] (libavutil/hwcontext.c:32-75) is a static array with one `#if CONFIG_*` row per device type, hw_type_names[] (:77-93)
 * one designated row per enumerator; the patch is one more row in each (`#if CONFIG_HIP  &ff_hwcontext_type_hip,` and
 * `[AV_HWDEVICE_TYPE_HIP] = "hip",`) plus the enum value.  The reference file is compiled unchanged, where it lies:
 *   - hw_table[]: the wrapper switches on the CUDA row for the duration of the include and lets the name in that row mean the hip
 *     type (a row is a pointer; the type's own .type is AV_HWDEVICE_TYPE_HIP).  CONFIG_CUDA is used nowhere else in hwcontext.c;
 *   - hw_type_names[]: AV_HWDEVICE_TYPE_OHCODEC — the last enumerator, named once in the file, as that array's last designator —
 *     expands to itself plus the `hip` row, so av_hwdevice_find_type_by_name("hip") / av_hwdevice_get_type_name() know the type
 *     (both uses are checked by the recipe, oracle/refbuild/Makefile).
 */
#include "config.h"
#undef CONFIG_CUDA
#define CONFIG_CUDA 1
#define ff_hwcontext_type_cuda ff_hwcontext_type_hip
#include "libavutil/hwcontext.h"
#include "avutil_hwcontext_hip.h"
#define AV_HWDEVICE_TYPE_OHCODEC AV_HWDEVICE_TYPE_OHCODEC] = "ohcodec", [FFHIP_HWDEVICE_TYPE_NB_REF] = "hip", [AV_HWDEVICE_TYPE_OHCODEC
#include "libavutil/hwcontext.c"
