/* shim_arena.h — what every host-pointer (signature-exact) face shares: the fallback bookkeeping and the scratch arena of one call
 * (shims.hip: the codec DSP tables; sws_api.hip: the swscale per-line members).  Internal to libffhip. */
#ifndef FFHIP_SHIM_ARENA_H
#define FFHIP_SHIM_ARENA_H

#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "common.h"

inline std::atomic<long> g_fallbacks;
inline void shim_note(const char *member, bool have_c)
{
    g_fallbacks++;
    if (!have_c)
        ffhip_set_error("ffhip: host face `%s` could not run on the device and displaced no C function: the call was NOT carried out", member);
}
/* answers a failed face through the displaced pointer */
#define SHIM_FB(tab, member, ...) do { const auto fn_ = __atomic_load_n(&(tab).member, __ATOMIC_RELAXED); shim_note(#member, fn_ != nullptr); \
                                       if (fn_) fn_(__VA_ARGS__); } while (0)

/* members of a context the init is about to overwrite: words of `incoming` that differ from what we install are the caller's
 * C functions (a second init of a table that already holds our faces must not make a face its own fallback) */
template <class T>
inline void fb_snapshot(T &fb, const T &incoming, const T &ours)
{
    static_assert(sizeof(T) % sizeof(void *) == 0, "a context is a table of function pointers");
    void *const *in = reinterpret_cast<void *const *>(&incoming), *const *ou = reinterpret_cast<void *const *>(&ours);
    void **f = reinterpret_cast<void **>(&fb);
    /* an init on one thread may run beside faces answering through the table on another: word-sized atomic stores (SHIM_FB loads
     * the same way), one writer at a time */
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < sizeof(T) / sizeof(void *); i++)
        if (in[i] != ou[i])
            __atomic_store_n(&f[i], in[i], __ATOMIC_RELAXED);
}

/* the scratch arena of one call: lock, reserve, (after the launch) bring everything back at once */
inline std::vector<uint8_t> g_bounce; /* guarded by the arena mutex */
struct Arena {
    std::unique_lock<std::mutex> lk;
    uint8_t *buf = nullptr;
    size_t bytes;
    bool ok = false;
    explicit Arena(size_t n) : lk(ffhip_scratch_mutex()), bytes(n)
    {
        const char *ef = FFHIP_KNOB("FFHIP_FAULT"); /* test hook: every face reports failure before touching anything */
        void *p = nullptr;
        if (!(ef && ef[0] == '1') && ffhip_scratch_reserve(n, &p) >= 0) {
            buf = static_cast<uint8_t *>(p);
            ok = true;
        }
    }
    bool down()
    {
        if (hipStreamSynchronize(0) != hipSuccess)
            return false;
        if (g_bounce.size() < bytes)
            g_bounce.resize(bytes);
        return hipMemcpy(g_bounce.data(), buf, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    }
    const uint8_t *host(const void *dev) const { return g_bounce.data() + (static_cast<const uint8_t *>(dev) - buf); }
};


#endif
