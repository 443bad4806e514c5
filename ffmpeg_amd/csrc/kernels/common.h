/* common.h — shared device/host helpers for the gfx950 kernels of libffhip. */
#ifndef FFHIP_KERNELS_COMMON_H
#define FFHIP_KERNELS_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ffhip_internal.h"

#define FFHIP_WAVE 64

/* Turn a failed HIP call into FFHIP_EIO + ffhip_last_error() text. */
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            ffhip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FFHIP_EIO;                                                                  \
        }                                                                                      \
    } while (0)

#define LAUNCH_CHECK()                                                                         \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) {                                                                \
            ffhip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return FFHIP_EIO;                                                                  \
        }                                                                                      \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
/*
 * gfx950 / ROCm 7.2 hazard (measured, see DESIGN.md "toolchain notes"): hipcc folds
 *     clip_u8(a >> n) | clip_u8(b >> n) << 8
 * into v_ashr_pk_u8_i32, and the upper 16 bits of that instruction's destination are NOT zero on
 * the hardware although the compiler goes on to OR the register into a packed dword as if they
 * were — bytes 2/3 of the packed word came back polluted (447 wrong bytes in a 64x16 frame).
 * The empty asm makes the clamped value opaque so the pattern never forms (no instruction emitted).
 */
__device__ __forceinline__ int clip_u8(int v)
{
    int r = min(max(v, 0), 255);
    asm("" : "+v"(r));
    return r;
}
__device__ __forceinline__ int clip3(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d)
{
    return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}
#endif

/* one lazily created scratch arena per process for the host-pointer (signature-exact) faces; every user holds
 * ffhip_scratch_mutex() from its reserve to its last copy-back */
int   ffhip_scratch_reserve(size_t bytes, void **dev);
#ifdef __cplusplus
#include <mutex>
std::mutex &ffhip_scratch_mutex(void);
#endif
/* called wherever a process-global device resource is created: pins ffhip_set_device() to that device from then on */
void  ffhip_note_device_resources(void);
int   ffhip_have_device(void);

#endif
