/*
 * integration/avutil_tx_list_hip.c — libavutil/tx.c with ff_tx_codelet_list_float_hip in codelet_list[].
 *
 * codelet_list[] (libavutil/tx.c:340-351) is a static array with one `#if` per arch; the patch is one more entry:
 *     #if CONFIG_HIP
 *         ff_tx_codelet_list_float_hip,
 *     #endif
 * The reference file is compiled unchanged, where it lies: this wrapper switches on the array's (otherwise unused on this host)
 * aarch64 slot for the duration of the include and lets the name in that slot mean the hip list.  tx.c uses ARCH_AARCH64 nowhere
 * else (checked by the recipe: oracle/refbuild/Makefile greps for it).
 */
#include "config.h"
#undef ARCH_AARCH64
#define ARCH_AARCH64 1
#define ff_tx_codelet_list_float_aarch64 ff_tx_codelet_list_float_hip
#include "libavutil/tx.c"
