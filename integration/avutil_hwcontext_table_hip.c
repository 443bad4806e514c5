/*
 * integration/avutil_hwcontext_table_hip.c — libavutil/hwcontext.c with the `hip` type in hw_table[].
 *
 * hw_table[] (libavutil/hwcontext.c:32-75) is a static array with one `#if CONFIG_*` row per device type; the patch is one more row
 * (`#if CONFIG_HIP  &ff_hwcontext_type_hip,`) plus the enum values.  The reference file is compiled unchanged, where it lies: this
 * wrapper switches on the CUDA row for the duration of the include and lets the name in that row mean the hip type, whose
 * .type is that slot's AV_HWDEVICE_TYPE_CUDA (integration/avutil_hwcontext_hip.h).  CONFIG_CUDA is used nowhere else in hwcontext.c
 * (the recipe checks).
 */
#include "config.h"
#undef CONFIG_CUDA
#define CONFIG_CUDA 1
#define ff_hwcontext_type_cuda ff_hwcontext_type_hip
#include "libavutil/hwcontext.c"
