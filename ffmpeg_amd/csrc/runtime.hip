/*
 * runtime.hip — device discovery, memory helpers and the staging arena behind the host-pointer faces.
 * The gating role of av_get_cpu_flags() in ff_*_init_<arch>() (libavutil/cpu.h) is played by
 * ffhip_device_count(): no usable device => every init returns FFHIP_ENOSYS and the caller keeps C.
 */
#include <mutex>

#include "kernels/common.h"
#include "kernels/progress_pool.h"

#include <atomic>
#include <vector>

#define FFHIP_MAX_DEVICES 64

static std::atomic<int> g_count{ -2 };
static std::mutex g_mu;

extern "C" int ffhip_device_count(void)
{
    int c = g_count.load(std::memory_order_acquire);
    if (c != -2)
        return c;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_count.load() == -2) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) {
            ffhip_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
            (void)hipGetLastError();
            n = 0;
        }
        if (n > FFHIP_MAX_DEVICES)
            n = FFHIP_MAX_DEVICES;
        g_count.store(n, std::memory_order_release);
    }
    return g_count.load();
}

/*
 * Which device a call runs on.  One process may drive every GPU of the node (the reference's execution model is one process with
 * frame / slice threads: libavcodec/pthread_frame.c, libswscale/swscale.c:1645-1679), so nothing in the library is tied to "the"
 * device:
 *   - contexts (FFHipSwsContext, FFHipTXContext, FFHipH264Picture, FFHipAac*, FFHipSwsUOps, FFHipDeviceSet) remember the device
 *     they were created on and make it current for the duration of each of their calls, whatever the calling thread is bound to;
 *   - context-free entry points (the batched `_dev` faces, the host-pointer shims) run on the calling thread's current device;
 *     the shared resources they use (staging arena, progress-counter pool, coefficient tables, dynamic-LDS attribute latches,
 *     compiled op-list modules) live in per-device tables indexed by it.
 * HIP's current device is per THREAD and a new thread starts on device 0.  The first ffhip_set_device() of the process also sets
 * the process default; a thread that never called ffhip_set_device() (an FFmpeg frame/slice worker calling a shim face) is bound
 * to that default at its first entry, so the workers of a process that chose device N do not silently run on device 0.
 */
static std::atomic<int> g_default_device{ -1 };
static thread_local bool t_bound;

static inline void thread_bind(void)
{
    if (t_bound)
        return;
    const int d = g_default_device.load(std::memory_order_acquire);
    if (d < 0)
        return; /* no process default yet: a worker that enters first must still pick up the one a later ffhip_set_device() makes */
    t_bound = true;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != d)
        (void)hipSetDevice(d);
}

int ffhip_have_device(void)
{
    if (ffhip_device_count() <= 0)
        return 0;
    thread_bind();
    return 1;
}

int ffhip_current_device(void)
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return d < FFHIP_MAX_DEVICES ? d : 0;
}

extern "C" int ffhip_set_device(int device)
{
    if (device < 0 || device >= ffhip_device_count())
        return FFHIP_EINVAL;
    HIP_TRY(hipSetDevice(device));
    t_bound = true;
    int none = -1;
    g_default_device.compare_exchange_strong(none, device);
    return 0;
}

/* ffhip_set_device() for the duration of a callback that must not re-bind the calling thread (buffer-pool callbacks run on whatever
 * thread drops the last reference): *prev receives what ffhip_device_pop() restores */
extern "C" int ffhip_device_push(int device, int *prev)
{
    if (!prev || device < 0 || device >= ffhip_device_count())
        return FFHIP_EINVAL;
    thread_bind();
    int cur = -1;
    *prev = -1;
    if (hipGetDevice(&cur) != hipSuccess)
        return FFHIP_ENOSYS;
    if (cur != device) {
        HIP_TRY(hipSetDevice(device));
        *prev = cur;
    }
    return 0;
}
extern "C" void ffhip_device_pop(int prev)
{
    if (prev >= 0)
        (void)hipSetDevice(prev);
}

/* everything queued on `first` so far happens before anything queued on `then` from now on (either may be NULL: the legacy default
 * stream, which a hipStreamNonBlocking stream is NOT ordered against by itself) */
extern "C" int ffhip_stream_order(void *first, void *then)
{
    if (first == then)
        return 0;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t r = hipEventRecord(e, (hipStream_t)first);
    if (r == hipSuccess)
        r = hipStreamWaitEvent((hipStream_t)then, e, 0);
    (void)hipEventDestroy(e); /* deferred by the runtime until the event has completed */
    if (r != hipSuccess) {
        ffhip_set_error("ffhip_stream_order: %s", hipGetErrorString(r));
        return FFHIP_EIO;
    }
    return 0;
}

extern "C" int ffhip_get_device(void)
{
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_current_device();
}

FFHipDeviceGuard::FFHipDeviceGuard(int device) : prev(-1)
{
    thread_bind();
    int cur = -1;
    if (device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess)
        prev = cur;
}
FFHipDeviceGuard::~FFHipDeviceGuard()
{
    if (prev >= 0)
        (void)hipSetDevice(prev);
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
/* which device an address lives on: the ordinal for device memory, FFHIP_EINVAL for host (or unknown) memory.  What a caller that is
 * handed pointers by a framework (an SwsPass of hardware frames) checks before it launches on them */
extern "C" int ffhip_pointer_device(const void *p)
{
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return FFHIP_EINVAL;
    }
    return a.type == hipMemoryTypeDevice ? a.device : FFHIP_EINVAL;
}

/* pitched copies for frame planes (what an hwcontext's transfer_data_to / _from needs): asynchronous on `stream`; the host side
 * must stay valid until the stream is synchronised (pageable host memory is staged by the runtime) */
extern "C" int ffhip_memcpy2d_h2d_async(void *d, size_t dpitch, const void *s, size_t spitch, size_t width_bytes, size_t rows, void *stream)
{
    if (!d || !s || dpitch < width_bytes || spitch < width_bytes)
        return FFHIP_EINVAL;
    HIP_TRY(hipMemcpy2DAsync(d, dpitch, s, spitch, width_bytes, rows, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}
extern "C" int ffhip_memcpy2d_d2h_async(void *d, size_t dpitch, const void *s, size_t spitch, size_t width_bytes, size_t rows, void *stream)
{
    if (!d || !s || dpitch < width_bytes || spitch < width_bytes)
        return FFHIP_EINVAL;
    HIP_TRY(hipMemcpy2DAsync(d, dpitch, s, spitch, width_bytes, rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}
extern "C" int ffhip_stream_create(void **stream)
{
    if (!stream)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return 0;
}
extern "C" int ffhip_stream_destroy(void *stream)
{
    if (stream)
        HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
extern "C" int ffhip_stream_synchronize(void *stream)
{
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    /* a finished wavefront launch OF THIS STREAM that lost a hand-off is reported here, not dropped */
    return ffhip_progress_check((hipStream_t)stream);
}

/* grow-only staging arena of the host-pointer faces, one per device.  ONE mutex per device guards it for every user: a face
 * holds ffhip_scratch_mutex() for its whole stage / run / copy-back sequence (the arena may be freed and reallocated by the next
 * caller's reserve).  Faces on different devices do not contend. */
struct DeviceArena {
    std::mutex mu;
    void *buf = nullptr;
    size_t size = 0;
    std::vector<uint8_t> bounce;
};
static DeviceArena g_arena[FFHIP_MAX_DEVICES];
std::mutex &ffhip_scratch_mutex(void) { return g_arena[ffhip_current_device()].mu; }
std::vector<uint8_t> &ffhip_scratch_bounce(void) { return g_arena[ffhip_current_device()].bounce; }
int ffhip_scratch_reserve(size_t bytes, void **dev)
{
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    DeviceArena &a = g_arena[ffhip_current_device()];
    if (bytes > a.size) {
        if (a.buf)
            HIP_TRY(hipFree(a.buf));
        a.buf = NULL;
        a.size = 0;
        size_t want = bytes + (bytes >> 1) + 4096;
        hipError_t e = hipMalloc(&a.buf, want);
        if (e != hipSuccess) {
            ffhip_set_error("scratch hipMalloc(%zu): %s", want, hipGetErrorString(e));
            return FFHIP_ENOMEM;
        }
        a.size = want;
    }
    *dev = a.buf;
    return 0;
}

/* run `f` once per device (dynamic-LDS attribute latches, coefficient-table uploads): `once` is a static of the caller */
bool FFHipPerDeviceOnce::enter(void)
{
    mu.lock();
    const int d = ffhip_current_device();
    if (done >> (d & 63) & 1) {
        mu.unlock();
        return false;
    }
    return true; /* locked: the caller does its work, then leave() */
}
void FFHipPerDeviceOnce::leave(bool ok)
{
    const int d = ffhip_current_device();
    if (ok)
        done |= 1ull << (d & 63);
    mu.unlock();
}

/* ---- achievable-bandwidth probe (bench.py: the box's streaming roofs beside the 8 TB/s spec) ----
 * A streaming kernel's rate on these parts depends on how its accesses are issued as much as on the mix (tools/ubench/membw2.hip,
 * profiles/r05_membw2.txt: 4.0 .. 6.1 TB/s of pure writes, 4.7 .. 5.9 TB/s of copy across 146 variants on one box).  So the probe is a
 * SWEEP, and its answer the best variant: U = 1, 2 or 8 accesses of 16 bytes in flight per lane before the first dependent store, plain
 * or non-temporal stores, non-temporal loads, 256 x 4 or 256 x 16 workgroups, grid-stride chunks or one private contiguous slice per
 * workgroup with the slices of an XCD's workgroups (blockIdx % 8) adjacent; pattern 5 is the runtime's own hipMemcpyDtoDAsync.
 * A wave-instruction touches 1 KiB, a workgroup's 4 KiB, consecutive instructions consecutive memory: what a row-streaming kernel does. */
typedef uint32_t bw_u4 __attribute__((ext_vector_type(4)));
/* RD : WR = 1 : K in units of 16 bytes (K = 1 with RD = 0: write-only, WR = 0: read-only) */
template <int K, int U, int NT, int RD, int WR, int PAT>
__global__ __launch_bounds__(256) void k_membw(const bw_u4 *__restrict__ src, bw_u4 *__restrict__ dst, size_t n_rd, uint32_t *sink)
{
    bw_u4 acc = { 0, 0, 0, 0 };
    const size_t per_block = n_rd / gridDim.x / (256 * U) * (256 * U);
    const size_t bid = PAT ? (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const size_t first = PAT ? bid * per_block : (size_t)blockIdx.x * 256 * U;
    const size_t last = PAT ? first + per_block : n_rd;
    const size_t step = PAT ? (size_t)256 * U : (size_t)gridDim.x * 256 * U;
    for (size_t base = first; base + 256 * U <= last; base += step) {
        bw_u4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (RD)
                v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
            else
                v[u] = bw_u4{ (uint32_t)i, 1, 2, 3 };
        }
        if (WR) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const size_t o = (base + (size_t)u * 256) * K + (size_t)k * 256 + threadIdx.x;
                    bw_u4 w = v[u];
                    w.x += k;
                    if (NT)
                        __builtin_nontemporal_store(w, dst + o);
                    else
                        dst[o] = w;
                }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++)
                acc += v[u];
        }
    }
    if (!WR && acc.x + acc.y + acc.z + acc.w == 0x12345)
        sink[0] = 1;
}

template <int K, int RD, int WR>
static void membw_launch(int variant, const bw_u4 *a, bw_u4 *b, size_t n_rd, uint32_t *sink)
{
    /* variant: bits 0-1 U index (1, 2, 8), bit 2 non-temporal, bit 3 256 x 16 workgroups (else 256 x 4), bit 4 private slices */
    const dim3 g(variant & 8 ? 4096 : 1024), t(256);
    const int u = variant & 3, nt = (variant >> 2) & 1, pat = (variant >> 4) & 1;
#define MB(U_, NT_, PAT_) hipLaunchKernelGGL((k_membw<K, U_, NT_, RD, WR, PAT_>), g, t, 0, 0, a, b, n_rd, sink)
#define MBU(NT_, PAT_) do { if (u == 0) MB(1, NT_, PAT_); else if (u == 1) MB(2, NT_, PAT_); else MB(8, NT_, PAT_); } while (0)
    if (nt) { if (pat) MBU(1, 1); else MBU(1, 0); }
    else    { if (pat) MBU(0, 1); else MBU(0, 0); }
#undef MBU
#undef MB
}

/* ---- frame memory with page-table friendly alignment (include/ffhip.h: ffhip_frames_alloc) ------------------------------------ */
namespace {
struct FrameRange { void *va; size_t size, chunk; std::vector<hipMemGenericAllocationHandle_t> h; };
std::mutex g_frames_mu;
std::vector<FrameRange> g_frames;
} // namespace

extern "C" int ffhip_frames_alloc(void **ptr, size_t bytes, size_t chunk)
{
    if (!ptr || !bytes)
        return FFHIP_EINVAL;
    *ptr = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return FFHIP_ENOSYS;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) {
        (void)hipGetLastError();
        ffhip_set_error("ffhip_frames_alloc: no virtual memory management on device %d", dev);
        return FFHIP_ENOSYS;
    }
    if (!chunk)
        chunk = (size_t)16 << 20;
    if ((chunk & (chunk - 1)) || chunk < gran) {
        ffhip_set_error("ffhip_frames_alloc: chunk %zu is not a power of two of at least the granularity %zu", chunk, gran);
        return FFHIP_EINVAL;
    }
    FrameRange r;
    r.chunk = chunk;
    r.size = (bytes + chunk - 1) / chunk * chunk;
    r.va = nullptr;
    if (hipMemAddressReserve(&r.va, r.size, chunk, nullptr, 0) != hipSuccess || !r.va) {
        (void)hipGetLastError();
        ffhip_set_error("ffhip_frames_alloc: hipMemAddressReserve(%zu, alignment %zu) failed", r.size, chunk);
        return FFHIP_ENOMEM;
    }
    bool ok = true;
    const size_t nch = r.size / chunk;
    /* the physical chunks come out of the allocator one after the other (on an empty device: physically consecutive); they are mapped
     * into the range in a fixed pseudo-random ORDER, so that the address bits above the chunk size of neighbouring pieces of the range
     * have nothing to do with each other (FFHIP_FRAMES_ORDER=0, measure build: in order) */
    std::vector<size_t> slot(nch);
    for (size_t i = 0; i < nch; i++)
        slot[i] = i;
    {
        const char *eo = FFHIP_KNOB("FFHIP_FRAMES_ORDER");
        if (!(eo && eo[0] == '0')) {
            uint32_t st = 0x9E3779B9u;
            for (size_t i = nch; i > 1; i--) {
                st = st * 1664525u + 1013904223u;
                const size_t j = (size_t)(((uint64_t)(st >> 8) * i) >> 24);
                const size_t t = slot[i - 1];
                slot[i - 1] = slot[j];
                slot[j] = t;
            }
        }
    }
    r.h.assign(nch, hipMemGenericAllocationHandle_t());
    size_t made = 0;
    for (size_t i = 0; i < nch && ok; i++) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) {
            ok = false;
            break;
        }
        r.h[slot[i]] = h;       /* r.h[k]: the handle mapped at chunk k of the range */
        made++;
        if (hipMemMap((uint8_t *)r.va + slot[i] * chunk, chunk, 0, h, 0) != hipSuccess)
            ok = false;
    }
    if (ok) {
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = hipMemSetAccess(r.va, r.size, &acc, 1) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        for (size_t i = 0; i < made; i++) {
            (void)hipMemUnmap((uint8_t *)r.va + slot[i] * chunk, chunk);
            (void)hipMemRelease(r.h[slot[i]]);
        }
        (void)hipMemAddressFree(r.va, r.size);
        (void)hipGetLastError();
        ffhip_set_error("ffhip_frames_alloc: mapping %zu bytes in chunks of %zu failed", r.size, chunk);
        return FFHIP_ENOMEM;
    }
    *ptr = r.va;
    std::lock_guard<std::mutex> lk(g_frames_mu);
    g_frames.push_back(std::move(r));
    return 0;
}

extern "C" int ffhip_frames_free(void *ptr)
{
    if (!ptr)
        return 0;
    FrameRange r;
    {
        std::lock_guard<std::mutex> lk(g_frames_mu);
        size_t i = 0;
        for (; i < g_frames.size(); i++)
            if (g_frames[i].va == ptr)
                break;
        if (i == g_frames.size()) {
            ffhip_set_error("ffhip_frames_free: %p is not a range of ffhip_frames_alloc", ptr);
            return FFHIP_EINVAL;
        }
        r = std::move(g_frames[i]);
        g_frames.erase(g_frames.begin() + (ptrdiff_t)i);
    }
    (void)hipDeviceSynchronize();
    for (size_t i = 0; i < r.h.size(); i++) {
        (void)hipMemUnmap((uint8_t *)r.va + i * r.chunk, r.chunk);
        (void)hipMemRelease(r.h[i]);
    }
    (void)hipMemAddressFree(r.va, r.size);
    (void)hipGetLastError();
    return 0;
}

extern "C" int ffhip_membw_probe(int pattern, size_t bytes, int reps, double *gbps)
{
    if (pattern < 0 || pattern > 5 || bytes < (1u << 20) || reps < 1 || !gbps)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    bytes &= ~(size_t)((1u << 20) - 1);
    bw_u4 *a = nullptr, *b = nullptr;
    uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int r = FFHIP_ENOMEM;
    double best = 0;
    /* `bytes` is the larger (written, or read for pattern 0) side */
    const int K = pattern == 3 ? 4 : pattern == 4 ? 2 : 1;
    /* the largest variant's grid footprint is 4096 workgroups x 256 lanes x 8 elements of 16 K bytes: a size that is not a multiple of it
     * leaves a tail no variant touches while `moved` counts it — round down when the size allows */
    {
        const size_t quantum = (size_t)4096 * 256 * 8 * 16 * K;
        if (bytes >= quantum)
            bytes -= bytes % quantum;
    }
    const size_t n_rd = bytes / 16 / K;
    const double moved = pattern == 2 || pattern == 5 ? 2.0 * bytes : pattern == 3 ? 1.25 * bytes : pattern == 4 ? 1.5 * bytes : (double)bytes;
    auto launch = [&](int v) {
        switch (pattern) {
        case 0: membw_launch<1, 1, 0>(v, a, b, n_rd, sink); break;
        case 1: membw_launch<1, 0, 1>(v, a, b, n_rd, sink); break;
        case 2: membw_launch<1, 1, 1>(v, a, b, n_rd, sink); break;
        case 3: membw_launch<4, 1, 1>(v, a, b, n_rd, sink); break;
        case 4: membw_launch<2, 1, 1>(v, a, b, n_rd, sink); break;
        default: (void)hipMemcpyDtoDAsync(b, a, bytes, 0); break;
        }
    };
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess)
        goto done;
    r = FFHIP_EIO;
    if (hipMemset(a, 1, bytes) != hipSuccess || hipMemset(b, 2, bytes) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess)
        goto done;
    for (int v = 0; v < (pattern == 5 ? 1 : 32); v++) {
        float ms = 0;
        if ((v & 3) == 3)
            continue;
        for (int i = 0; i < 2; i++)
            launch(v);
        if (hipEventRecord(e0, 0) != hipSuccess)
            goto done;
        for (int i = 0; i < reps; i++)
            launch(v);
        if (hipEventRecord(e1, 0) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess ||
            hipGetLastError() != hipSuccess)
            goto done;
        const double g = moved * reps / (ms * 1e-3) / 1e9;
        if (g > best)
            best = g;
    }
    *gbps = best;
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
