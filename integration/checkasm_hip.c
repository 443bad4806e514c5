/*
 * integration/checkasm_hip.c — tests/checkasm/checkasm.c with the `hip` arch in its cpu-flag table.
 *
 * The reference's table (tests/checkasm/checkasm.c:322-385) is one `#if ARCH_* ... #elif` chain of { name, suffix, flag } rows; a
 * maintainer adds `{ "HIP", "hip", AV_CPU_FLAG_HIP }` to it.  Here the reference file is compiled UNCHANGED, where it lies: this
 * wrapper includes it with checkasm_main() renamed, and the renamed hook swaps the table for the row above (the generic-arch build
 * has an empty table) and reports the flag as available when a HIP device is present — then hands over to the real checkasm_main().
 * Every test then runs once for the C reference and once more with AV_CPU_FLAG_HIP forced, exactly as it does for sse2 or neon.
 */
#include "config.h"
#include <checkasm/checkasm.h>

#define checkasm_main ffhip_checkasm_main_hook
#include "tests/checkasm/checkasm.c"
#undef checkasm_main

#include "ffhip.h"
#include "hip_cpu.h"

int checkasm_main(CheckasmConfig *config, int argc, const char *argv[]);
/* oracle/refbuild/ffref_shim_ops.c: the "hip" SwsOpBackend compiled into the reference's ops dispatch; its entry points are bound at
 * run time (NULL: the backend answers ENOTSUP and the dispatch moves on, as for a backend whose cpu flag is off) */
int ffref_sws_hip_bind(void *compile, void *free_, void *block_size, void *func, void *set_fallback);

static const CheckasmCpuInfo hip_cpuflags[] = {
    { "HIP", "hip", AV_CPU_FLAG_HIP },
    { NULL }
};

static void hip_set_cpu_flags(CheckasmCpu flags)
{
    av_force_cpu_flags((int)flags);
    if (flags & AV_CPU_FLAG_HIP)
        ffref_sws_hip_bind((void *)ffhip_sws_uops_compile, (void *)ffhip_sws_uops_free, (void *)ffhip_sws_uops_block_size,
                           (void *)ffhip_sws_uops_func, (void *)ffhip_sws_uops_set_fallback);
    else
        ffref_sws_hip_bind(NULL, NULL, NULL, NULL, NULL);
}

int ffhip_checkasm_main_hook(CheckasmConfig *config, int argc, const char *argv[])
{
    config->cpu_flags = hip_cpuflags;
    config->set_cpu_flags = hip_set_cpu_flags;
    if (ffhip_device_count() > 0)
        config->cpu |= AV_CPU_FLAG_HIP;       /* what av_get_cpu_flags() would report once libavutil/cpu.c probes the arch */
    else
        fprintf(stderr, "checkasm: no HIP device (%s): only the C functions are exercised\n", ffhip_last_error());
    return checkasm_main(config, argc, argv);
}
