/* progress_pool.h — the progress counters of the row-ordered ("wavefront") launches in flight: H.264 frame-order deblocking, the
 * H.264 intra reconstruction wavefront, the VP9 superblock-order loop filter.  Internal to libffhip (progress_pool.hip). */
#ifndef FFHIP_PROGRESS_POOL_H
#define FFHIP_PROGRESS_POOL_H

#include <hip/hip_runtime.h>

#define FFHIP_PROGRESS_SLOT_INTS 8192 /* 2048 until round 4: 64 4K luma planes (34 bands each) then split into launches of 60 + 4 pictures, and the 4 paid a whole latency chain (2.26 ms against 0.96 ms for 32 planes) */

struct FFHipProgressSlot {
    int *prog; /* `nints` zeroed (in stream order) progress words on the current device */
    int *fail; /* the slot's FAIL word, pinned host memory mapped into the device: a kernel sets it on a spin timeout */
    int  index, device;
};

/* A free slot of the current device's pool, its first `nints` (<= FFHIP_PROGRESS_SLOT_INTS) counters zeroed on `stream`.  The pool
 * lock is NOT held when this returns: the slot is simply owned until ffhip_progress_release(). */
int ffhip_progress_acquire(int nints, hipStream_t stream, FFHipProgressSlot *s);
/* launched: an event behind the launch on `stream` marks when the slot may be reused; !launched (an error path): the slot is free
 * again at once. */
int ffhip_progress_release(const FFHipProgressSlot *s, hipStream_t stream, bool launched);
/* FFHIP_EIO (once) if a finished launch that was issued on `stream` lost a hand-off — keyed by stream, so the owner of picture A
 * hears about picture A and nobody else does.  Checks the current device's pool. */
int ffhip_progress_check(hipStream_t stream);
/* on: launches the calling thread queues from now on file a lost hand-off under `stream` whatever stream they run on (the picture
 * layer's chroma wavefront runs on a private second stream; its caller only ever asks about its own); off: back to the launch's. */
void ffhip_progress_report_to(hipStream_t stream, bool on);

#endif
