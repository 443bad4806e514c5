/*
 * runtime.hip — device discovery, memory helpers and the staging arena behind the host-pointer faces.
 * The gating role of av_get_cpu_flags() in ff_*_init_<arch>() (libavutil/cpu.h) is played by
 * ffhip_device_count(): no usable device => every init returns FFHIP_ENOSYS and the caller keeps C.
 */
#include <mutex>

#include "kernels/common.h"

static int g_count = -2;
static std::mutex g_mu;

extern "C" int ffhip_device_count(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_count == -2) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) {
            ffhip_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
            (void)hipGetLastError();
            n = 0;
        }
        g_count = n;
    }
    return g_count;
}

int ffhip_have_device(void) { return ffhip_device_count() > 0; }

extern "C" int ffhip_set_device(int device)
{
    if (device < 0 || device >= ffhip_device_count())
        return FFHIP_EINVAL;
    HIP_TRY(hipSetDevice(device));
    return 0;
}

extern "C" int ffhip_malloc(void **p, size_t bytes)
{
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    hipError_t e = hipMalloc(p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        ffhip_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? FFHIP_ENOMEM : FFHIP_EIO;
    }
    return 0;
}
extern "C" int ffhip_free(void *p)
{
    if (p)
        HIP_TRY(hipFree(p));
    return 0;
}
extern "C" int ffhip_memcpy_h2d(void *d, const void *s, size_t n)
{
    HIP_TRY(hipMemcpy(d, s, n, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int ffhip_memcpy_d2h(void *d, const void *s, size_t n)
{
    HIP_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int ffhip_stream_synchronize(void *stream)
{
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

/* grow-only arena; the single-call shims are serialised by g_shim_mu in their own files */
static void  *g_scratch;
static size_t g_scratch_sz;
int ffhip_scratch_reserve(size_t bytes, void **dev)
{
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    if (bytes > g_scratch_sz) {
        if (g_scratch)
            HIP_TRY(hipFree(g_scratch));
        g_scratch = NULL;
        g_scratch_sz = 0;
        size_t want = bytes + (bytes >> 1) + 4096;
        hipError_t e = hipMalloc(&g_scratch, want);
        if (e != hipSuccess) {
            ffhip_set_error("scratch hipMalloc(%zu): %s", want, hipGetErrorString(e));
            return FFHIP_ENOMEM;
        }
        g_scratch_sz = want;
    }
    *dev = g_scratch;
    return 0;
}
