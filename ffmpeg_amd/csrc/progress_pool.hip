/*
 * progress_pool.hip — progress counters of the row-ordered launches in flight.
 *
 * The reference orders these loops by walking macroblock / superblock rows on the CPU (libavcodec/h264_slice.c loop_filter() per
 * row, h264_mb hl_decode_mb() in raster order, vp9block/vp9lpf per superblock row); on the device a row's wave waits for the row
 * above to pass a column, which takes a few counters per launch.  They come from a ring of slots in ONE allocation per device,
 * made on first use.  A slot is FREE, OWNED (handed out, launch not issued yet) or IN FLIGHT (an event recorded behind the launch
 * says when it is free again).  Every slot has a FAIL word in pinned host memory the kernels set on a spin timeout (never in a
 * correct run: a lost hand-off); it is folded into a per-device list of failed STREAMS when the slot is recycled or checked, and
 * ffhip_progress_check(stream) — called by ffhip_stream_synchronize() and the picture pipeline's flush — reports a failure to the
 * stream that issued the launch, once.
 */
#include <mutex>
#include <thread>
#include <vector>

#include "kernels/common.h"
#include "kernels/progress_pool.h"

#define SLOTS 64
enum { S_FREE, S_OWNED, S_FLIGHT };
struct Slot {
    hipEvent_t done;
    hipStream_t stream;
    int state;
};
struct Pool {
    std::mutex mu;
    int *counters = nullptr; /* device */
    int *fail = nullptr;     /* pinned host, one word per slot */
    Slot slot[SLOTS];
    unsigned next = 0;
    std::vector<hipStream_t> failed;
};
static Pool g_pool[64];

static int pool_init(Pool &p)
{
    if (p.counters)
        return 0;
    int *c = nullptr, *f = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c), (size_t)SLOTS * FFHIP_PROGRESS_SLOT_INTS * sizeof(int)));
    if (hipHostMalloc(reinterpret_cast<void **>(&f), SLOTS * sizeof(int), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
        (void)hipFree(c);
        ffhip_set_error("ffhip progress pool: hipHostMalloc failed");
        return FFHIP_ENOMEM;
    }
    for (int i = 0; i < SLOTS; i++) {
        f[i] = 0;
        p.slot[i].state = S_FREE;
        p.slot[i].stream = nullptr;
        if (hipEventCreateWithFlags(&p.slot[i].done, hipEventDisableTiming) != hipSuccess) {
            for (int k = 0; k < i; k++)
                (void)hipEventDestroy(p.slot[k].done);
            (void)hipFree(c);
            (void)hipHostFree(f);
            ffhip_set_error("ffhip progress pool: hipEventCreate failed");
            return FFHIP_EIO;
        }
    }
    p.fail = f;
    p.counters = c;
    return 0;
}

/* hipErrorNotReady must not linger as the thread's "last error": the launch paths read hipGetLastError() right after */
static bool event_done(hipEvent_t ev)
{
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess)
        return true;
    (void)hipGetLastError();
    return e != hipErrorNotReady; /* an event the runtime no longer answers for (its stream was destroyed): that launch is over */
}

/* under p.mu: an in-flight slot whose launch has finished becomes free; its FAIL word moves to the failed-stream list */
static void retire(Pool &p, int i)
{
    if (p.fail[i]) {
        p.fail[i] = 0;
        p.failed.push_back(p.slot[i].stream);
    }
    p.slot[i].state = S_FREE;
}

/* a launch queued on a helper stream on behalf of a caller's stream files its failure under the caller's */
static thread_local bool t_report_set;
static thread_local hipStream_t t_report;
void ffhip_progress_report_to(hipStream_t stream, bool on)
{
    t_report_set = on;
    t_report = stream;
}

int ffhip_progress_acquire(int nints, hipStream_t stream, FFHipProgressSlot *s)
{
    if (nints < 0 || nints > FFHIP_PROGRESS_SLOT_INTS) {
        ffhip_set_error("ffhip: %d progress counters exceed the %d of one pool slot", nints, FFHIP_PROGRESS_SLOT_INTS);
        return FFHIP_EINVAL;
    }
    const int dev = ffhip_current_device();
    Pool &p = g_pool[dev];
    std::unique_lock<std::mutex> lk(p.mu);
    const int r = pool_init(p);
    if (r < 0)
        return r;
    int got = -1;
    for (;;) {
        /* a free slot, or one whose launch has completed (hipEventQuery: no waiting under the lock) */
        for (unsigned k = 0; k < SLOTS && got < 0; k++) {
            const int i = (int)((p.next + k) % SLOTS);
            if (p.slot[i].state == S_FLIGHT && event_done(p.slot[i].done))
                retire(p, i);
            if (p.slot[i].state == S_FREE)
                got = i;
        }
        if (got >= 0)
            break;
        /* every slot is busy: wait for the oldest in-flight launch WITHOUT the lock (other threads keep launching) */
        int w = -1;
        for (unsigned k = 0; k < SLOTS && w < 0; k++)
            if (p.slot[(p.next + k) % SLOTS].state == S_FLIGHT)
                w = (int)((p.next + k) % SLOTS);
        hipEvent_t ev = w >= 0 ? p.slot[w].done : nullptr;
        lk.unlock();
        if (ev) {
            if (hipEventSynchronize(ev) != hipSuccess) { /* seen for events of destroyed streams: fall back to polling (event_done) */
                (void)hipGetLastError();
                std::this_thread::yield();
            }
        } else
            std::this_thread::yield(); /* all 64 slots owned by threads between acquire and release: they are about to record */
        lk.lock();
    }
    p.next = (unsigned)got + 1;
    p.slot[got].state = S_OWNED;
    p.slot[got].stream = t_report_set ? t_report : stream;
    lk.unlock();
    s->prog = p.counters + (size_t)got * FFHIP_PROGRESS_SLOT_INTS;
    s->fail = p.fail + got;
    s->index = got;
    s->device = dev;
    if (nints && hipMemsetAsync(s->prog, 0, (size_t)nints * sizeof(int), stream) != hipSuccess) {
        ffhip_set_error("ffhip: hipMemsetAsync of the progress counters failed");
        (void)ffhip_progress_release(s, stream, false);
        return FFHIP_EIO;
    }
    return 0;
}

int ffhip_progress_release(const FFHipProgressSlot *s, hipStream_t stream, bool launched)
{
    Pool &p = g_pool[s->device];
    std::lock_guard<std::mutex> lk(p.mu);
    Slot &sl = p.slot[s->index];
    if (launched) {
        const hipError_t e = hipEventRecord(sl.done, stream);
        if (e == hipSuccess) {
            sl.state = S_FLIGHT;
            return 0;
        }
        /* nothing marks the end of the launch: drain the stream so the slot cannot be handed out under a running kernel */
        (void)hipStreamSynchronize(stream);
        retire(p, s->index);
        ffhip_set_error("ffhip: hipEventRecord failed: %s", hipGetErrorString(e));
        return FFHIP_EIO;
    }
    /* nothing was launched, but acquire()'s hipMemsetAsync may still be queued on `stream`: a slot handed to another stream now
     * could have its live counters zeroed by it.  The slot stays in flight until the stream has passed this point. */
    if (hipEventRecord(sl.done, stream) == hipSuccess) {
        sl.state = S_FLIGHT;
        return 0;
    }
    (void)hipStreamSynchronize(stream);
    sl.state = S_FREE;
    return 0;
}

int ffhip_progress_check(hipStream_t stream)
{
    Pool &p = g_pool[ffhip_current_device()];
    std::lock_guard<std::mutex> lk(p.mu);
    if (!p.counters)
        return 0;
    for (int i = 0; i < SLOTS; i++)
        if (p.slot[i].state == S_FLIGHT && p.slot[i].stream == stream && event_done(p.slot[i].done))
            retire(p, i);
    bool hit = false;
    for (size_t k = 0; k < p.failed.size();)
        if (p.failed[k] == stream) {
            p.failed.erase(p.failed.begin() + (long)k);
            hit = true;
        } else
            k++;
    if (hit) {
        ffhip_set_error("ffhip: a row-ordered launch (deblocking / intra wavefront / loop filter) on this stream timed out waiting for a row hand-off; its picture is only partly processed");
        return FFHIP_EIO;
    }
    return 0;
}
