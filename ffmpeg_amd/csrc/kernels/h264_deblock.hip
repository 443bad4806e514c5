/*
 * h264_deblock.hip — H.264 8-bit in-loop deblocking filters.
 *
 * Bit-exact restatement of h264_{v,h}_loop_filter_{luma,chroma}[_intra]_8_c
 * (libavcodec/h264dsp_template.c:104-330): per sample line across an edge,
 * p3 p2 p1 p0 | q0 q1 q2 q3, normal filter (bS < 4) with per-4-line (luma) / per-2-line (chroma) tc0,
 * strong filter (bS == 4) without.  Every read of a line happens before any write of that line.
 *
 * Two faces:
 *   k_h264_loop_filter     n edges whose written pixels are pairwise disjoint (the function-level batch,
 *                          checkasm's 32x16 tiles): 16 lanes per edge, one lane per sample line.
 *   k_h264_deblock_frame   a whole picture in the decoder's order (libavcodec/h264_loopfilter.c:716):
 *                          MBs raster, per MB vertical edges 0..3 then horizontal edges 0..3.  Filters of
 *                          neighbouring MBs overlap, so the order is a true dependency: MB(x,y) needs
 *                          (x-1,y) and (x+1,y-1).  One WAVE walks one MB row left to right with the current
 *                          MB + 4 columns / 4 rows of context in an LDS tile; rows hand off through an
 *                          agent-scope release/acquire progress counter (row y may start MB x once row y-1
 *                          has published x+2).  Results equal the serial order bit for bit.
 */
#include <atomic>
#include <mutex>

#include "common.h"
#include "h264_kernels.h"

struct LfLine { int p3, p2, p1, p0, q0, q1, q2, q3; };

/* Filters one sample line in place; returns the mask of changed taps: bit0 p2, bit1 p1, bit2 p0, bit3 q0,
 * bit4 q1, bit5 q2.  cls: 0 luma, 1 chroma, 2 luma intra, 3 chroma intra. */
__device__ __forceinline__ int lf_line(LfLine &v, int cls, int alpha, int beta, int tc0)
{
    const int p0 = v.p0, p1 = v.p1, p2 = v.p2, q0 = v.q0, q1 = v.q1, q2 = v.q2;
    if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta)
        return 0;
    if (cls == 0) {
        if (tc0 < 0)
            return 0;
        int tc = tc0, m = 4 | 8;
        if (abs(p2 - p0) < beta) {
            if (tc0) {
                v.p1 = p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0);
                m |= 2;
            }
            tc++;
        }
        if (abs(q2 - q0) < beta) {
            if (tc0) {
                v.q1 = q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0);
                m |= 16;
            }
            tc++;
        }
        const int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
        v.p0 = min(max(p0 + delta, 0), 255);
        v.q0 = min(max(q0 - delta, 0), 255);
        return m;
    }
    if (cls == 1) {
        if (tc0 <= 0)
            return 0;
        const int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc0, tc0);
        v.p0 = min(max(p0 + delta, 0), 255);
        v.q0 = min(max(q0 - delta, 0), 255);
        return 4 | 8;
    }
    if (cls == 3) {
        v.p0 = (2 * p1 + p0 + q1 + 2) >> 2;
        v.q0 = (2 * q1 + q0 + p1 + 2) >> 2;
        return 4 | 8;
    }
    /* luma intra */
    int m = 4 | 8;
    if (abs(p0 - q0) < ((alpha >> 2) + 2)) {
        if (abs(p2 - p0) < beta) {
            v.p0 = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
            v.p1 = (p2 + p1 + p0 + q0 + 2) >> 2;
            v.p2 = (2 * v.p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
            m |= 1 | 2;
        } else {
            v.p0 = (2 * p1 + p0 + q1 + 2) >> 2;
        }
        if (abs(q2 - q0) < beta) {
            v.q0 = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
            v.q1 = (p0 + q0 + q1 + q2 + 2) >> 2;
            v.q2 = (2 * v.q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
            m |= 16 | 32;
        } else {
            v.q0 = (2 * q1 + q0 + p1 + 2) >> 2;
        }
    } else {
        v.p0 = (2 * p1 + p0 + q1 + 2) >> 2;
        v.q0 = (2 * q1 + q0 + p1 + 2) >> 2;
    }
    return m;
}

/* load / filter / store one line through any byte pointer; xs = step across the edge */
template <typename P>
__device__ __forceinline__ void lf_apply(P pix, ptrdiff_t xs, int cls, int alpha, int beta, int tc0)
{
    LfLine v;
    const bool luma = !(cls & 1);
    v.p1 = pix[-2 * xs]; v.p0 = pix[-xs]; v.q0 = pix[0]; v.q1 = pix[xs];
    v.p2 = luma ? pix[-3 * xs] : 0; v.q2 = luma ? pix[2 * xs] : 0;
    v.p3 = cls == 2 ? pix[-4 * xs] : 0; v.q3 = cls == 2 ? pix[3 * xs] : 0;
    const int m = lf_line(v, cls, alpha, beta, tc0);
    if (m & 1)  pix[-3 * xs] = (uint8_t)v.p2;
    if (m & 2)  pix[-2 * xs] = (uint8_t)v.p1;
    if (m & 4)  pix[-xs] = (uint8_t)v.p0;
    if (m & 8)  pix[0] = (uint8_t)v.q0;
    if (m & 16) pix[xs] = (uint8_t)v.q1;
    if (m & 32) pix[2 * xs] = (uint8_t)v.q2;
}

/* ---- function-level batch ------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_h264_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n)
{
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int d = threadIdx.x & 15;
    if (e >= n)
        return;
    const FFHipH264Edge ed = edges[e];
    const int kind = ed.kind & 7;
    const bool chroma = kind & 2, intra = kind & 4, vert_edge = kind & 1; /* h_ filters a vertical edge */
    if (chroma && d >= 8)
        return;
    const ptrdiff_t xs = vert_edge ? 1 : stride, ys = vert_edge ? stride : 1;
    const int tc0 = intra ? 0 : ed.tc0[chroma ? d >> 1 : d >> 2];
    lf_apply(base + ed.offset + d * ys, xs, (chroma ? 1 : 0) + (intra ? 2 : 0), ed.alpha, ed.beta, tc0);
}

int ffhip_launch_h264_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    hipLaunchKernelGGL(k_h264_loop_filter, dim3(cdiv(n, 16)), dim3(256), 0, stream, base, stride, edges, n);
    LAUNCH_CHECK();
    return 0;
}

/* ---- frame order ---------------------------------------------------------------------------------- */
#define TP 24 /* LDS tile pitch: 4 context columns + 16 + pad */

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64) void k_h264_deblock_frame(uint8_t *luma, size_t frame_pitch, ptrdiff_t stride, int mb_w, int mb_h,
                                                           const FFHipH264Edge *edges, int *progress)
{
    /* blockIdx.y = frame: frames are independent, each has its own counters (mb_h progress words + a fail flag) */
    luma += (size_t)blockIdx.y * frame_pitch;
    edges += (size_t)blockIdx.y * mb_w * mb_h * 8;
    progress += (size_t)blockIdx.y * (mb_h + 1);
    int *fail = progress + mb_h;
    /* tile[r][c]: r = picture row - (16*my - 4), c = picture column - (16*mx - 4); rows are dword aligned */
    __shared__ __align__(16) uint8_t tile[20 * TP];
    const int my = blockIdx.x, lane = threadIdx.x;
    uint8_t *rowbase = luma + (ptrdiff_t)my * 16 * stride;
    const bool dw_ok = !(((uintptr_t)luma | (size_t)stride) & 3);
    const int pr = lane >> 2, pc = 4 * (lane & 3); /* this lane's dword of a 16x16 block */
    /* the MB's own 16x16 pixels are nobody else's to change before we filter it: fetched one MB ahead */
    uint32_t own = 0;
    auto fetch_own = [&](int mx) {
        const uint8_t *p = rowbase + mx * 16 + (ptrdiff_t)pr * stride + pc;
        if (dw_ok)
            own = *reinterpret_cast<const uint32_t *>(p);
        else
            own = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    };
    /* the MB's 8 edge records (96 bytes) likewise: one dword per lane, handed to the filters through LDS — read from
     * global memory inside the edge loops they were eight dependent round trips per macroblock */
    __shared__ uint32_t edl[24];
    uint32_t edw = 0;
    auto fetch_edges = [&](int mx) {
        if (lane < 24)
            edw = reinterpret_cast<const uint32_t *>(edges + (size_t)(my * mb_w + mx) * 8)[lane];
    };
    /* rows -4..-1 over this MB, written by the wave of row my-1: coherent (agent-scope) loads, because a later MB's
     * context shares cache lines with an earlier one's and no acquire may lie in between (see `known`) */
    auto load_top = [&](int mx) {
        uint32_t v = 0;
        if (lane < 16) {
            const uint8_t *p = rowbase + mx * 16 + (ptrdiff_t)(pr - 4) * stride + pc;
            if (dw_ok)
                v = __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        }
        return v;
    };
    int known = 0;          /* last value seen of progress[my-1] (acquire): polled only when it is not enough */
    bool have_top = false;  /* topv already holds this MB's context rows (fetched while the previous MB was filtered) */
    uint32_t topv = 0;
    fetch_own(0);
    fetch_edges(0);
    for (int mx = 0; mx < mb_w; mx++) {
        uint8_t *mb = rowbase + mx * 16;
        *reinterpret_cast<uint32_t *>(&tile[(pr + 4) * TP + 4 + pc]) = own;
        if (lane < 24)
            edl[lane] = edw;
        if (mx + 1 < mb_w) {
            fetch_own(mx + 1);
            fetch_edges(mx + 1);
        }
        /* ---- wait for the row above: MB (mx+1, my-1) done, i.e. progress[my-1] >= min(mx+2, mb_w) ---- */
        if (my > 0) {
            const int want = min(mx + 2, mb_w);
            if (!have_top || !dw_ok) {
                int spins = 0;
                while (known < want) {
                    /* dword-aligned pictures: everything that crosses rows moves with device-scope (cache-bypassing)
                     * loads and stores, so the hand-off needs ORDER only — the wait on the counter's value before the
                     * context loads are issued — and no agent-scope acquire, whose L2 invalidate (one per macroblock
                     * and wave, on all of an XCD's lines) capped the throughput of a batch */
                    known = dw_ok ? __hip_atomic_load(&progress[my - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : __hip_atomic_load(&progress[my - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (known >= want)
                        break;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 24)) { /* never in a correct run; do not hang the device */
                        if (lane == 0)
                            atomicExch(fail, 1);
                        return;
                    }
                }
                if (!dw_ok) /* byte loads go through L1: always behind a fresh acquire */
                    known = __hip_atomic_load(&progress[my - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                else
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                topv = load_top(mx);
            }
            if (lane < 16)
                *reinterpret_cast<uint32_t *>(&tile[pr * TP + 4 + pc]) = topv;
            /* the next MB's context, if the row above has already published it */
            have_top = false;
            if (dw_ok && mx + 1 < mb_w && known >= min(mx + 3, mb_w)) {
                topv = load_top(mx + 1);
                have_top = true;
            }
        }
        wave_lds_sync();
        const FFHipH264Edge *e = reinterpret_cast<const FFHipH264Edge *>(edl);
        /* ---- vertical edges, left to right: lane = row ---- */
        for (int k = 0; k < 4; k++) {
            const FFHipH264Edge ed = e[k];
            if (lane < 16 && ed.alpha && ed.beta && !(k == 0 && mx == 0)) {
                const bool intra = ed.kind >= 4;
                lf_apply(&tile[(lane + 4) * TP + 4 + 4 * k], 1, intra ? 2 : 0, ed.alpha, ed.beta, intra ? 0 : ed.tc0[lane >> 2]);
            }
            wave_lds_sync();
        }
        /* ---- horizontal edges, top to bottom: lane = column ---- */
        for (int k = 0; k < 4; k++) {
            const FFHipH264Edge ed = e[4 + k];
            if (lane < 16 && ed.alpha && ed.beta && !(k == 0 && my == 0)) {
                const bool intra = ed.kind >= 4;
                lf_apply(&tile[(4 + 4 * k) * TP + 4 + lane], TP, intra ? 2 : 0, ed.alpha, ed.beta, intra ? 0 : ed.tc0[lane >> 2]);
            }
            wave_lds_sync();
        }
        /* ---- write back what this MB may have changed: rows -3..-1 x columns 0..15, rows 0..15 x columns -4..15
         * (column -4 and untouched pixels are rewritten with their own final values; the corner is left alone) ---- */
        if (dw_ok) {
            for (int i = lane; i < 16 * 5 + 3 * 4; i += 64) {
                int r, c;
                if (i < 80) { r = i / 5; c = 4 * (i % 5) - 4; } else { r = (i - 80) / 4 - 3; c = 4 * ((i - 80) & 3); }
                if ((c < 0 && mx == 0) || (r < 0 && my == 0))
                    continue;
                /* device-scope (write-through) stores: the release below then finds no dirty lines of ours to write back
                 * from this XCD's L2 — with plain stores every macroblock's release flushed whatever the whole XCD had
                 * written since the last one, and that flush rate capped the throughput of a batch */
                __hip_atomic_store(reinterpret_cast<uint32_t *>(mb + (ptrdiff_t)r * stride + c),
                                   *reinterpret_cast<const uint32_t *>(&tile[(r + 4) * TP + 4 + c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            for (int i = lane; i < 19 * 19; i += 64) {
                const int r = i / 19 - 3, c = i % 19 - 3;
                if ((r < 0 && c < 0) || (r < 0 && my == 0) || (c < 0 && mx == 0))
                    continue;
                mb[(ptrdiff_t)r * stride + c] = tile[(r + 4) * TP + 4 + c];
            }
        }
        /* ---- the MB's right 4 columns (and those of the context rows) are the next MB's left context ---- */
        wave_lds_sync();
        const uint32_t keep = *reinterpret_cast<const uint32_t *>(&tile[(lane < 20 ? lane : 0) * TP + 4 + 12]);
        wave_lds_sync();
        if (lane < 20)
            *reinterpret_cast<uint32_t *>(&tile[lane * TP]) = keep;
        /* ---- publish (release: this wave's stores above become visible before the counter does) ---- */
        if (dw_ok) {
            /* the write-through stores above are complete (acknowledged) before the counter moves */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            if (lane == 0)
                __hip_atomic_store(&progress[my], mx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == 0) {
            __hip_atomic_store(&progress[my], mx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

/*
 * k_h264_deblock_frame_chroma — one 4:2:0 chroma plane in the decoder's order: per macroblock (8x8 samples) the vertical edges
 * at x = 0 and 4, then the horizontal ones at y = 0 and 4 (filter_mb_dir filters chroma on the even luma edges,
 * h264_loopfilter.c:644-700).  Same wavefront as the luma kernel: one wave per macroblock row, MB (x, y) after (x+1, y-1); the
 * tile is the MB + 4 columns / 2 rows of context; a chroma filter reads two samples and writes one on either side, so the two
 * edges of a direction touch disjoint samples and run side by side (lanes 0-7 / 8-15).  Dword-aligned planes only: everything
 * that crosses rows moves with device-scope loads / stores and the hand-off is order-only, as in the luma kernel's fast path.
 */
#define CTP 16 /* chroma tile pitch: 4 context columns + 8 + 4 */
__global__ __launch_bounds__(64) void k_h264_deblock_frame_chroma(uint8_t *plane, size_t frame_pitch, ptrdiff_t stride, int mb_w, int mb_h,
                                                                  const FFHipH264Edge *edges, int *progress)
{
    plane += (size_t)blockIdx.y * frame_pitch;
    edges += (size_t)blockIdx.y * mb_w * mb_h * 4;
    progress += (size_t)blockIdx.y * (mb_h + 1);
    int *fail = progress + mb_h;
    /* tile[r][c]: r = row - (8 my - 2), c = column - (8 mx - 4) */
    __shared__ __align__(16) uint8_t tile[10 * CTP];
    __shared__ uint32_t edl[12];
    const int my = blockIdx.x, lane = threadIdx.x;
    uint8_t *rowbase = plane + (ptrdiff_t)my * 8 * stride;
    const int pr = lane >> 1, pc = 4 * (lane & 1); /* lanes 0..15: this lane's dword of the 8x8 block */
    uint32_t own = 0, edw = 0;
    auto fetch = [&](int mx) {
        if (lane < 16)
            own = *reinterpret_cast<const uint32_t *>(rowbase + mx * 8 + (ptrdiff_t)pr * stride + pc);
        if (lane < 12)
            edw = reinterpret_cast<const uint32_t *>(edges + (size_t)(my * mb_w + mx) * 4)[lane];
    };
    int known = 0;
    fetch(0);
    for (int mx = 0; mx < mb_w; mx++) {
        uint8_t *mb = rowbase + mx * 8;
        if (lane < 16)
            *reinterpret_cast<uint32_t *>(&tile[(pr + 2) * CTP + 4 + pc]) = own;
        if (lane < 12)
            edl[lane] = edw;
        if (mx + 1 < mb_w)
            fetch(mx + 1);
        if (my > 0) {
            /* rows -2, -1 over this MB belong to the row above: wait until it has finished MB mx + 1 */
            const int want = min(mx + 2, mb_w);
            int spins = 0;
            while (known < want) {
                known = __hip_atomic_load(&progress[my - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (known >= want)
                    break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 24)) { /* never in a correct run; do not hang the device */
                    if (lane == 0)
                        atomicExch(fail, 1);
                    return;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane < 4) {
                const int r = lane >> 1, c = 4 * (lane & 1);
                const uint32_t v = __hip_atomic_load(reinterpret_cast<const uint32_t *>(mb + (ptrdiff_t)(r - 2) * stride + c), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                *reinterpret_cast<uint32_t *>(&tile[r * CTP + 4 + c]) = v;
            }
        }
        wave_lds_sync();
        const FFHipH264Edge *e = reinterpret_cast<const FFHipH264Edge *>(edl);
        const int k = (lane >> 3) & 1, line = lane & 7;
        { /* vertical edges at columns 0 and 4: lane = (edge, row) */
            const FFHipH264Edge ed = e[k];
            if (lane < 16 && ed.alpha && ed.beta && !(k == 0 && mx == 0)) {
                const bool intra = ed.kind >= 4;
                lf_apply(&tile[(line + 2) * CTP + 4 + 4 * k], 1, intra ? 3 : 1, ed.alpha, ed.beta, intra ? 0 : ed.tc0[line >> 1]);
            }
        }
        wave_lds_sync();
        { /* horizontal edges at rows 0 and 4: lane = (edge, column) */
            const FFHipH264Edge ed = e[2 + k];
            if (lane < 16 && ed.alpha && ed.beta && !(k == 0 && my == 0)) {
                const bool intra = ed.kind >= 4;
                lf_apply(&tile[(2 + 4 * k) * CTP + 4 + line], CTP, intra ? 3 : 1, ed.alpha, ed.beta, intra ? 0 : ed.tc0[line >> 1]);
            }
        }
        wave_lds_sync();
        /* write back rows 0..7 x columns -4..7 and row -1 x columns 0..7 (untouched samples keep their values) */
        if (lane < 26) {
            int r, c;
            if (lane < 24) { r = lane / 3; c = 4 * (lane % 3) - 4; } else { r = -1; c = 4 * (lane - 24); }
            if (!((c < 0 && mx == 0) || (r < 0 && my == 0)))
                __hip_atomic_store(reinterpret_cast<uint32_t *>(mb + (ptrdiff_t)r * stride + c),
                                   *reinterpret_cast<const uint32_t *>(&tile[(r + 2) * CTP + 4 + c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        /* the MB's right 4 columns are the next MB's left context */
        wave_lds_sync();
        const uint32_t keep = *reinterpret_cast<const uint32_t *>(&tile[(lane < 10 ? lane : 0) * CTP + 4 + 4]);
        wave_lds_sync();
        if (lane < 10)
            *reinterpret_cast<uint32_t *>(&tile[lane * CTP]) = keep;
        /* publish: the write-through stores above are acknowledged before the counter moves */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0)
            __hip_atomic_store(&progress[my], mx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* progress counters: a small ring of slots in one device allocation made on first use (stream-ordered zeroing
 * per launch); a per-launch hipMallocAsync/hipFreeAsync pair serialises launches across streams */
#define DB_SLOTS 64
#define DB_SLOT_INTS 2048
static int *g_db_pool;
static std::atomic<unsigned> g_db_next;
static std::mutex g_db_mu;

int ffhip_launch_h264_deblock_frames(uint8_t *luma, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                     const FFHipH264Edge *edges, hipStream_t stream)
{
    if (mb_w <= 0 || mb_h <= 0 || nframes <= 0)
        return 0;
    if (mb_h + 1 > DB_SLOT_INTS) {
        ffhip_set_error("ffhip_h264_deblock_frame: %d macroblock rows exceed the supported %d", mb_h, DB_SLOT_INTS - 1);
        return FFHIP_EINVAL;
    }
    {
        std::lock_guard<std::mutex> lk(g_db_mu);
        if (!g_db_pool)
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g_db_pool), (size_t)DB_SLOTS * DB_SLOT_INTS * sizeof(int)));
    }
    const int per_launch = DB_SLOT_INTS / (mb_h + 1); /* frames whose counters fit one pool slot */
    for (int f0 = 0; f0 < nframes; f0 += per_launch) {
        const int nf = nframes - f0 < per_launch ? nframes - f0 : per_launch;
        int *prog = g_db_pool + (size_t)(g_db_next.fetch_add(1) % DB_SLOTS) * DB_SLOT_INTS;
        HIP_TRY(hipMemsetAsync(prog, 0, (size_t)nf * (mb_h + 1) * sizeof(int), stream));
        hipLaunchKernelGGL(k_h264_deblock_frame, dim3(mb_h, nf), dim3(64), 0, stream, luma + (size_t)f0 * frame_pitch, frame_pitch,
                           stride, mb_w, mb_h, edges + (size_t)f0 * mb_w * mb_h * 8, prog);
        LAUNCH_CHECK();
    }
    return 0;
}

int ffhip_launch_h264_deblock_frame(uint8_t *luma, ptrdiff_t stride, int mb_w, int mb_h, const FFHipH264Edge *edges,
                                    hipStream_t stream)
{
    return ffhip_launch_h264_deblock_frames(luma, 0, 1, stride, mb_w, mb_h, edges, stream);
}

int ffhip_launch_h264_deblock_frames_chroma(uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                            const FFHipH264Edge *edges, hipStream_t stream)
{
    if (mb_w <= 0 || mb_h <= 0 || nframes <= 0)
        return 0;
    if (((uintptr_t)plane | (size_t)stride | frame_pitch) & 3) {
        ffhip_set_error("ffhip_h264_deblock_frame_chroma: plane, stride and frame pitch must be 4-byte aligned");
        return FFHIP_EINVAL;
    }
    if (mb_h + 1 > DB_SLOT_INTS) {
        ffhip_set_error("ffhip_h264_deblock_frame_chroma: %d macroblock rows exceed the supported %d", mb_h, DB_SLOT_INTS - 1);
        return FFHIP_EINVAL;
    }
    {
        std::lock_guard<std::mutex> lk(g_db_mu);
        if (!g_db_pool)
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g_db_pool), (size_t)DB_SLOTS * DB_SLOT_INTS * sizeof(int)));
    }
    const int per_launch = DB_SLOT_INTS / (mb_h + 1);
    for (int f0 = 0; f0 < nframes; f0 += per_launch) {
        const int nf = nframes - f0 < per_launch ? nframes - f0 : per_launch;
        int *prog = g_db_pool + (size_t)(g_db_next.fetch_add(1) % DB_SLOTS) * DB_SLOT_INTS;
        HIP_TRY(hipMemsetAsync(prog, 0, (size_t)nf * (mb_h + 1) * sizeof(int), stream));
        hipLaunchKernelGGL(k_h264_deblock_frame_chroma, dim3(mb_h, nf), dim3(64), 0, stream, plane + (size_t)f0 * frame_pitch, frame_pitch,
                           stride, mb_w, mb_h, edges + (size_t)f0 * mb_w * mb_h * 4, prog);
        LAUNCH_CHECK();
    }
    return 0;
}
