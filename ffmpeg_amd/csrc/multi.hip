/*
 * multi.hip — one process, several GPUs: the device set and the frame-batch scatter / gather behind it (include/ffhip.h).
 *
 * The hot path has no cross-frame dependency: a batch of frames / macroblock lists / transforms is cut into
 * contiguous ranges, one per device, and every device runs the single-GPU batch entry points on its own range.  There is no
 * reduction anywhere, so the only inter-GPU traffic is the optional scatter of a batch that originates on one device and the
 * gather of the results.  The reference's counterpart is its thread-level sharding inside one process — frame threads
 * (libavcodec/pthread_frame.c) and swscale's slice threads (libswscale/swscale.c:1645-1679) — which is why this lives behind the
 * C ABI and not only in the python harness (ffmpeg_amd/dist.py keeps the one-process-per-GPU form of the same partition).
 *
 * Transport: hipMemcpyPeerAsync over xGMI, one copy per destination, each on the DESTINATION member's stream so that the seven
 * links of the root are driven concurrently (xGMI is point-to-point: the root's egress, 7 x ~153 GB/s, is the bound).  Ordering
 * is by events: a scatter waits for what the root's stream has produced; after a gather the root's stream waits for every copy.
 */
#include <vector>

#include "kernels/common.h"

struct FFHipDeviceSet {
    std::vector<int> dev;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> ev; /* one per member: "my copies are queued" / "the root's data is ready" */
};

extern "C" void ffhip_shard_range(int64_t n_items, int rank, int world, int64_t *lo, int64_t *hi)
{
    const int64_t n = n_items > 0 ? n_items : 0;
    const int64_t per = world > 0 ? (n + world - 1) / world : n;
    int64_t l = (int64_t)rank * per;
    if (l > n) l = n;
    if (l < 0) l = 0;
    int64_t h = l + per;
    if (h > n) h = n;
    if (lo) *lo = l;
    if (hi) *hi = h;
}

extern "C" void ffhip_shard_frame_pairs(int64_t n_frames, int rank, int world, int64_t *plo, int64_t *phi, int64_t *flo, int64_t *fhi)
{
    int64_t l, h;
    ffhip_shard_range(n_frames > 0 ? n_frames - 1 : 0, rank, world, &l, &h);
    if (plo) *plo = l;
    if (phi) *phi = h;
    if (flo) *flo = l;
    if (fhi) *fhi = h > l ? h + 1 : l; /* a rank's pairs [l, h) read frames [l, h]: its own range plus ONE halo frame */
}

extern "C" void ffhip_device_set_free(FFHipDeviceSet **ps)
{
    if (!ps || !*ps)
        return;
    FFHipDeviceSet *s = *ps;
    for (size_t i = 0; i < s->dev.size(); i++) {
        FFHipDeviceGuard dg(s->dev[i]);
        if (i < s->ev.size() && s->ev[i])
            (void)hipEventDestroy(s->ev[i]);
        if (i < s->stream.size() && s->stream[i])
            (void)hipStreamDestroy(s->stream[i]);
    }
    delete s;
    *ps = nullptr;
}

extern "C" int ffhip_device_set_create(FFHipDeviceSet **ps, const int *devices, int n)
{
    if (!ps)
        return FFHIP_EINVAL;
    *ps = nullptr;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    const int count = ffhip_device_count();
    if (n <= 0 || !devices) /* all of them */
        n = count;
    FFHipDeviceSet *s = new (std::nothrow) FFHipDeviceSet();
    if (!s)
        return FFHIP_ENOMEM;
    for (int i = 0; i < n; i++) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= count) {
            ffhip_set_error("ffhip_device_set_create: device %d of %d", d, count);
            ffhip_device_set_free(&s);
            return FFHIP_EINVAL;
        }
        s->dev.push_back(d);
    }
    for (int i = 0; i < n; i++) {
        FFHipDeviceGuard dg(s->dev[i]);
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            ffhip_set_error("ffhip_device_set_create: stream / event creation on device %d failed", s->dev[i]);
            if (st)
                (void)hipStreamDestroy(st);
            ffhip_device_set_free(&s);
            return FFHIP_EIO;
        }
        s->stream.push_back(st);
        s->ev.push_back(ev);
        /* direct xGMI access to every other member (a second enable of the same pair is not an error here) */
        for (int k = 0; k < n; k++) {
            if (s->dev[k] == s->dev[i])
                continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, s->dev[i], s->dev[k]) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(s->dev[k], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                    ffhip_set_error("hipDeviceEnablePeerAccess(%d -> %d): %s", s->dev[i], s->dev[k], hipGetErrorString(e));
                    ffhip_device_set_free(&s);
                    return FFHIP_EIO;
                }
                (void)hipGetLastError();
            } /* no direct path: hipMemcpyPeerAsync still works, staged by the runtime */
        }
    }
    *ps = s;
    return 0;
}

extern "C" int ffhip_device_set_size(const FFHipDeviceSet *s) { return s ? (int)s->dev.size() : FFHIP_EINVAL; }
extern "C" int ffhip_device_set_device(const FFHipDeviceSet *s, int i)
{
    return s && i >= 0 && i < (int)s->dev.size() ? s->dev[i] : FFHIP_EINVAL;
}
extern "C" void *ffhip_device_set_stream(const FFHipDeviceSet *s, int i)
{
    return s && i >= 0 && i < (int)s->dev.size() ? (void *)s->stream[i] : nullptr;
}
extern "C" int ffhip_device_set_bind(const FFHipDeviceSet *s, int i)
{
    if (!s || i < 0 || i >= (int)s->dev.size())
        return FFHIP_EINVAL;
    return ffhip_set_device(s->dev[i]);
}

extern "C" int ffhip_device_set_synchronize(FFHipDeviceSet *s)
{
    if (!s)
        return FFHIP_EINVAL;
    int r = 0;
    for (size_t i = 0; i < s->dev.size(); i++) {
        FFHipDeviceGuard dg(s->dev[i]);
        const int ri = ffhip_stream_synchronize(s->stream[i]);
        if (ri < 0 && r == 0)
            r = ri;
    }
    return r;
}

static int check_ranges(const FFHipDeviceSet *s, int root, const int64_t *lo, const int64_t *hi, int64_t n_items, const char *who)
{
    if (!s || root < 0 || root >= (int)s->dev.size() || !lo || !hi) {
        ffhip_set_error("%s: bad device set / root / ranges", who);
        return FFHIP_EINVAL;
    }
    for (size_t i = 0; i < s->dev.size(); i++)
        if (lo[i] < 0 || hi[i] < lo[i] || hi[i] > n_items) {
            ffhip_set_error("%s: member %zu range [%lld, %lld) outside [0, %lld)", who, i, (long long)lo[i], (long long)hi[i], (long long)n_items);
            return FFHIP_EINVAL;
        }
    return 0;
}

/* full[lo[i] .. hi[i]) on the root -> shards[i] on member i (ranges may overlap: halos).  Asynchronous: member i's stream holds
 * its copy, ordered behind what the ROOT's stream had queued at the time of the call. */
extern "C" int ffhip_batch_scatter_ranges(FFHipDeviceSet *s, int root, const void *full, size_t item_bytes, int64_t n_items, const int64_t *lo,
                                          const int64_t *hi, void *const *shards)
{
    int r = check_ranges(s, root, lo, hi, n_items, "ffhip_batch_scatter");
    if (r < 0)
        return r;
    if (!full || !shards)
        return FFHIP_EINVAL;
    {
        FFHipDeviceGuard dg(s->dev[root]);
        HIP_TRY(hipEventRecord(s->ev[root], s->stream[root]));
    }
    for (size_t i = 0; i < s->dev.size(); i++) {
        const size_t bytes = (size_t)(hi[i] - lo[i]) * item_bytes;
        if (!bytes)
            continue;
        if (!shards[i])
            return FFHIP_EINVAL;
        const uint8_t *src = static_cast<const uint8_t *>(full) + (size_t)lo[i] * item_bytes;
        FFHipDeviceGuard dg(s->dev[i]);
        if ((int)i != root)
            HIP_TRY(hipStreamWaitEvent(s->stream[i], s->ev[root], 0));
        if (shards[i] == src)
            continue; /* the root's shard in place */
        if (s->dev[i] == s->dev[root])
            HIP_TRY(hipMemcpyAsync(shards[i], src, bytes, hipMemcpyDeviceToDevice, s->stream[i]));
        else
            HIP_TRY(hipMemcpyPeerAsync(shards[i], s->dev[i], src, s->dev[root], bytes, s->stream[i]));
    }
    return 0;
}

/* the inverse: shards[i] on member i -> full[lo[i] .. hi[i]) on the root (ranges must not overlap).  Member i's copy is queued on
 * its own stream (behind the work that produced the shard); the root's stream then waits for all of them. */
extern "C" int ffhip_batch_gather_ranges(FFHipDeviceSet *s, int root, void *full, size_t item_bytes, int64_t n_items, const int64_t *lo,
                                         const int64_t *hi, const void *const *shards)
{
    int r = check_ranges(s, root, lo, hi, n_items, "ffhip_batch_gather");
    if (r < 0)
        return r;
    if (!full || !shards)
        return FFHIP_EINVAL;
    for (size_t i = 0; i < s->dev.size(); i++) {
        const size_t bytes = (size_t)(hi[i] - lo[i]) * item_bytes;
        if (!bytes)
            continue;
        if (!shards[i])
            return FFHIP_EINVAL;
        uint8_t *dst = static_cast<uint8_t *>(full) + (size_t)lo[i] * item_bytes;
        FFHipDeviceGuard dg(s->dev[i]);
        if (shards[i] != dst) {
            if (s->dev[i] == s->dev[root])
                HIP_TRY(hipMemcpyAsync(dst, shards[i], bytes, hipMemcpyDeviceToDevice, s->stream[i]));
            else
                HIP_TRY(hipMemcpyPeerAsync(dst, s->dev[root], shards[i], s->dev[i], bytes, s->stream[i]));
        }
        if ((int)i != root)
            HIP_TRY(hipEventRecord(s->ev[i], s->stream[i]));
    }
    FFHipDeviceGuard dg(s->dev[root]);
    for (size_t i = 0; i < s->dev.size(); i++)
        if ((int)i != root && hi[i] > lo[i])
            HIP_TRY(hipStreamWaitEvent(s->stream[root], s->ev[i], 0));
    return 0;
}

static void even_ranges(const FFHipDeviceSet *s, int64_t n_items, std::vector<int64_t> &lo, std::vector<int64_t> &hi)
{
    const int w = (int)s->dev.size();
    lo.resize(w);
    hi.resize(w);
    for (int i = 0; i < w; i++)
        ffhip_shard_range(n_items, i, w, &lo[i], &hi[i]);
}

extern "C" int ffhip_batch_scatter(FFHipDeviceSet *s, int root, const void *full, size_t item_bytes, int64_t n_items, void *const *shards)
{
    if (!s)
        return FFHIP_EINVAL;
    std::vector<int64_t> lo, hi;
    even_ranges(s, n_items, lo, hi);
    return ffhip_batch_scatter_ranges(s, root, full, item_bytes, n_items, lo.data(), hi.data(), shards);
}

extern "C" int ffhip_batch_gather(FFHipDeviceSet *s, int root, void *full, size_t item_bytes, int64_t n_items, const void *const *shards)
{
    if (!s)
        return FFHIP_EINVAL;
    std::vector<int64_t> lo, hi;
    even_ranges(s, n_items, lo, hi);
    return ffhip_batch_gather_ranges(s, root, full, item_bytes, n_items, lo.data(), hi.data(), shards);
}

/* frames of a sequence for the motion search: member i gets the frames of its pairs, halo frame included */
extern "C" int ffhip_batch_scatter_frames_for_pairs(FFHipDeviceSet *s, int root, const void *frames, size_t frame_bytes, int64_t n_frames,
                                                    void *const *shards)
{
    if (!s)
        return FFHIP_EINVAL;
    const int w = (int)s->dev.size();
    std::vector<int64_t> lo(w), hi(w);
    for (int i = 0; i < w; i++)
        ffhip_shard_frame_pairs(n_frames, i, w, nullptr, nullptr, &lo[i], &hi[i]);
    return ffhip_batch_scatter_ranges(s, root, frames, frame_bytes, n_frames, lo.data(), hi.data(), shards);
}
