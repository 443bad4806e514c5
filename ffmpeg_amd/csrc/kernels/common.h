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

/* the residual row segment z[0..NS-1] (NS = 4, 8, 16, 32) added to NS picture samples at d8 (bd 8: bytes; above: uint16_t, clipped to
 * (1 << bd) - 1), as wide as d8's alignment allows: 16-byte accesses for 16-bit samples, 8-byte ones for bytes (one dword for NS = 4) */
template <int NS>
__device__ __forceinline__ void ffhip_add_row(uint8_t *d8, const int (&z)[NS], int bd)
{
    if (bd > 8) {
        const int maxv = (1 << bd) - 1;
        uint16_t *d = reinterpret_cast<uint16_t *>(d8);
        auto two = [&](uint32_t p, int a, int b) {
            return (uint32_t)min(max((int)(p & 0xFFFF) + a, 0), maxv) | (uint32_t)min(max((int)(p >> 16) + b, 0), maxv) << 16;
        };
        if (NS >= 8 && !(reinterpret_cast<uintptr_t>(d) & 15)) {
#pragma unroll
            for (int q = 0; q < NS / 8; q++) {
                const uint4 p = reinterpret_cast<const uint4 *>(d)[q];
                reinterpret_cast<uint4 *>(d)[q] = make_uint4(two(p.x, z[8 * q], z[8 * q + 1]), two(p.y, z[8 * q + 2], z[8 * q + 3]),
                                                             two(p.z, z[8 * q + 4], z[8 * q + 5]), two(p.w, z[8 * q + 6], z[8 * q + 7]));
            }
        } else if (!(reinterpret_cast<uintptr_t>(d) & 7)) {
#pragma unroll
            for (int q = 0; q < NS / 4; q++) {
                const uint2 p = reinterpret_cast<const uint2 *>(d)[q];
                reinterpret_cast<uint2 *>(d)[q] = make_uint2(two(p.x, z[4 * q], z[4 * q + 1]), two(p.y, z[4 * q + 2], z[4 * q + 3]));
            }
        } else {
#pragma unroll
            for (int k = 0; k < NS; k++)
                d[k] = (uint16_t)min(max((int)d[k] + z[k], 0), maxv);
        }
        return;
    }
    auto four = [&](uint32_t p, const int *zz) {
        return pack4(clip_u8((int)(p & 0xFF) + zz[0]), clip_u8((int)((p >> 8) & 0xFF) + zz[1]), clip_u8((int)((p >> 16) & 0xFF) + zz[2]),
                     clip_u8((int)(p >> 24) + zz[3]));
    };
    if (NS >= 8 && !(reinterpret_cast<uintptr_t>(d8) & 7)) {
#pragma unroll
        for (int q = 0; q < NS / 8; q++) {
            const uint2 p = reinterpret_cast<const uint2 *>(d8)[q];
            reinterpret_cast<uint2 *>(d8)[q] = make_uint2(four(p.x, &z[8 * q]), four(p.y, &z[8 * q + 4]));
        }
    } else if (!(reinterpret_cast<uintptr_t>(d8) & 3)) {
#pragma unroll
        for (int q = 0; q < NS / 4; q++)
            reinterpret_cast<uint32_t *>(d8)[q] = four(reinterpret_cast<const uint32_t *>(d8)[q], &z[4 * q]);
    } else {
#pragma unroll
        for (int k = 0; k < NS; k++)
            d8[k] = (uint8_t)clip_u8((int)d8[k] + z[k]);
    }
}
#endif

/* one lazily created staging arena PER DEVICE for the host-pointer (signature-exact) faces; every user holds
 * ffhip_scratch_mutex() (the current device's) from its reserve to its last copy-back */
int   ffhip_scratch_reserve(size_t bytes, void **dev);
int   ffhip_have_device(void);     /* also binds a thread that never chose a device to the process default (runtime.hip) */
int   ffhip_current_device(void);  /* the calling thread's HIP device */
#ifdef __cplusplus
#include <mutex>
#include <vector>
std::mutex &ffhip_scratch_mutex(void);
std::vector<uint8_t> &ffhip_scratch_bounce(void); /* host bounce buffer of the current device's arena; guarded by its mutex */

/* makes a context's device current for the duration of one of its calls (contexts are bound to the device they were created on) */
struct FFHipDeviceGuard {
    int prev;
    explicit FFHipDeviceGuard(int device);
    ~FFHipDeviceGuard();
    FFHipDeviceGuard(const FFHipDeviceGuard &) = delete;
    FFHipDeviceGuard &operator=(const FFHipDeviceGuard &) = delete;
};

/* per-device one-time setup (function attributes and tables are per device):
 *     static FFHipPerDeviceOnce once;  if (once.enter()) { bool ok = ...; once.leave(ok); } */
struct FFHipPerDeviceOnce {
    std::mutex mu;
    uint64_t done = 0;
    bool enter(void);
    void leave(bool ok);
};
#endif

/* experiment / fault-injection switches exist only in the measurement build (libffhip_measure.so, -DFFHIP_MEASURE): the product
 * library reads no environment variable, so nothing a user exports can change its pixels */
#ifdef FFHIP_MEASURE
#include <stdlib.h>
#define FFHIP_KNOB(name) getenv(name)
#else
#define FFHIP_KNOB(name) ((const char *)0)
#endif

#endif
