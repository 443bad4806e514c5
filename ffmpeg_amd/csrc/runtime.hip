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

/*
 * Process-global device resources (the staging arena below, the deblocking progress pool, the dynamic-LDS attribute latches of
 * the transforms) are created once, on the device that is current at first use.  They remember that device here; selecting
 * another one afterwards would make the host-pointer faces dereference the first GPU's memory, so it is refused (one process
 * per GPU: bind first, then work).
 */
static int g_resource_device = -1;
void ffhip_note_device_resources(void)
{
    int d = -1;
    if (hipGetDevice(&d) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_resource_device < 0)
            g_resource_device = d;
    }
}

extern "C" int ffhip_set_device(int device)
{
    if (device < 0 || device >= ffhip_device_count())
        return FFHIP_EINVAL;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_resource_device >= 0 && g_resource_device != device) {
            ffhip_set_error("ffhip_set_device(%d): this process already holds device resources on device %d "
                            "(call ffhip_set_device before any other entry point)", device, g_resource_device);
            return FFHIP_EINVAL;
        }
    }
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
int ffhip_h264_deblock_check(void);
extern "C" int ffhip_stream_synchronize(void *stream)
{
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return ffhip_h264_deblock_check(); /* a finished deblocking launch that lost a hand-off is reported here, not dropped */
}

/* grow-only arena of the host-pointer faces.  ONE mutex guards it for every user: a face holds ffhip_scratch_mutex() for its
 * whole stage / run / copy-back sequence (the arena may be freed and reallocated by the next caller's reserve). */
static void  *g_scratch;
static size_t g_scratch_sz;
static std::mutex g_scratch_mu;
std::mutex &ffhip_scratch_mutex(void) { return g_scratch_mu; }
int ffhip_scratch_reserve(size_t bytes, void **dev)
{
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    ffhip_note_device_resources();
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

/* ---- achievable-bandwidth probe (bench.py: the box's streaming roofs beside the 8 TB/s spec) ---- */
typedef uint32_t bw_u4 __attribute__((ext_vector_type(4)));
template <int MODE> /* 0 read, 1 write, 2 copy, 3 read n/4 + write n (the 1080p -> 4K scaler's mix), 4 read n/2 + write n (yuv420p -> rgb24's) */
__global__ __launch_bounds__(256) void k_membw(const bw_u4 *__restrict__ src, bw_u4 *__restrict__ dst, size_t n16, uint32_t *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bw_u4 acc = { 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        bw_u4 v = { (uint32_t)i, 1, 2, 3 };
        if (MODE == 0 || MODE == 2)
            v = src[i];
        if (MODE == 3 && (i & 3) == 0)
            v = src[i >> 2];
        if (MODE == 4 && (i & 1) == 0)
            v = src[i >> 1];
        if (MODE == 0)
            acc += v;
        else
            dst[i] = v;
    }
    if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 0x12345)
        sink[0] = 1;
}

extern "C" int ffhip_membw_probe(int pattern, size_t bytes, int reps, double *gbps)
{
    if (pattern < 0 || pattern > 4 || bytes < (1u << 20) || reps < 1 || !gbps)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    bytes &= ~(size_t)63;
    bw_u4 *a = nullptr, *b = nullptr;
    uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int r = FFHIP_ENOMEM;
    float ms = 0;
    auto launch = [&]() {
        const dim3 g(2048), t(256);
        switch (pattern) {
        case 0: hipLaunchKernelGGL((k_membw<0>), g, t, 0, 0, a, b, bytes / 16, sink); break;
        case 1: hipLaunchKernelGGL((k_membw<1>), g, t, 0, 0, a, b, bytes / 16, sink); break;
        case 2: hipLaunchKernelGGL((k_membw<2>), g, t, 0, 0, a, b, bytes / 16, sink); break;
        case 3: hipLaunchKernelGGL((k_membw<3>), g, t, 0, 0, a, b, bytes / 16, sink); break;
        default: hipLaunchKernelGGL((k_membw<4>), g, t, 0, 0, a, b, bytes / 16, sink); break;
        }
    };
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess)
        goto done;
    r = FFHIP_EIO;
    if (hipMemset(a, 1, bytes) != hipSuccess || hipMemset(b, 2, bytes) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess)
        goto done;
    for (int i = 0; i < 2; i++)
        launch();
    if (hipEventRecord(e0, 0) != hipSuccess)
        goto done;
    for (int i = 0; i < reps; i++)
        launch();
    if (hipEventRecord(e1, 0) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess ||
        hipGetLastError() != hipSuccess)
        goto done;
    {
        const double moved = pattern == 2 ? 2.0 * bytes : pattern == 3 ? 1.25 * bytes : pattern == 4 ? 1.5 * bytes : (double)bytes;
        *gbps = moved * reps / (ms * 1e-3) / 1e9;
    }
    r = 0;
done:
    if (r < 0)
        ffhip_set_error("ffhip_membw_probe: %s", r == FFHIP_ENOMEM ? "hipMalloc failed" : "HIP call failed");
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    return r;
}
