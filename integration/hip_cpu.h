/* integration/hip_cpu.h — the one line libavutil/cpu.h gains: a cpu flag for the `hip` arch, so that -cpuflags, av_force_cpu_flags()
 * and checkasm can switch the arch on and off like any other (libavutil/cpu.h:32-95 holds the AV_CPU_FLAG_* bits; 0x1000000 is
 * free on every host arch this library can meet).  "Available" means a usable HIP device: ffhip_device_count() > 0. */
#ifndef FFHIP_INTEGRATION_HIP_CPU_H
#define FFHIP_INTEGRATION_HIP_CPU_H
#define AV_CPU_FLAG_HIP 0x1000000
#endif
