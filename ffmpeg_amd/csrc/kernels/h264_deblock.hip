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
#include <mutex>
#include <stdlib.h>

#include <string.h>

#include "common.h"
#include <atomic>
#include "h264_kernels.h"

#include "h264_lf_line.h"

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
    uint8_t *pix = base + ed.offset + d * ys;
    const int cls = (chroma ? 1 : 0) + (intra ? 2 : 0);
    if (vert_edge && !(reinterpret_cast<uintptr_t>(pix) & 3)) {
        /* a vertical edge's line is 8 contiguous bytes p3 .. q3: two dwords in, the dwords that changed out (the sample-wise
         * form below costs up to 8 byte loads and 6 byte stores per lane) */
        uint32_t *w = reinterpret_cast<uint32_t *>(pix - 4);
        const uint32_t a = w[0], b = w[1];
        LfLine v = { (int)(a & 255), (int)((a >> 8) & 255), (int)((a >> 16) & 255), (int)(a >> 24),
                     (int)(b & 255), (int)((b >> 8) & 255), (int)((b >> 16) & 255), (int)(b >> 24) };
        const int m = lf_line(v, cls, ed.alpha, ed.beta, tc0);
        if (m & 7)
            w[0] = (uint32_t)v.p3 | (uint32_t)v.p2 << 8 | (uint32_t)v.p1 << 16 | (uint32_t)v.p0 << 24;
        if (m & 56)
            w[1] = (uint32_t)v.q0 | (uint32_t)v.q1 << 8 | (uint32_t)v.q2 << 16 | (uint32_t)v.q3 << 24;
        return;
    }
    lf_apply(pix, xs, cls, ed.alpha, ed.beta, tc0);
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
                                                           const FFHipH264Edge *edges, int *progress, int *fail)
{
    /* blockIdx.y = frame: frames are independent, each has its own counters (mb_h progress words; one spare) */
    luma += (size_t)blockIdx.y * frame_pitch;
    edges += (size_t)blockIdx.y * mb_w * mb_h * 8;
    progress += (size_t)blockIdx.y * (mb_h + 1);
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
                            __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
 * k_h264_deblock_band — the same decoder-order wavefront, built around the time of ONE macroblock.
 *
 * A picture is a chain of ~mb_w + 2 mb_h dependent macroblocks and a macroblock is a chain of 8 filters (every edge reads
 * samples the previous one wrote): the picture's time is (mb_w + 2 mb_h) x the time one wave needs for one macroblock.  Measured
 * on the frame kernel above (tools/db_exp.py): 5.5 us per macroblock, of which 4 us are the filters themselves (~2000 issued
 * instructions: byte loads / stores around every edge, per-lane control flow, edge records fetched through LDS), 1.2 us the
 * tile bookkeeping and only the rest the hand-off.  Hence:
 *   - a pass keeps its line in REGISTERS across its edges: lane = row for the vertical edges (the row's samples arrive as
 *     dwords), lane = column for the horizontal ones; the tile is touched once per pass, no barrier between edges;
 *   - the macroblock's edge records arrive through SCALAR loads one macroblock ahead: alpha / beta / kind are SGPRs, the
 *     skip and intra decisions are scalar branches, the filters themselves are branch-free (selects, v_sad_u32, v_med3);
 *   - two tiles alternate: the left context of a macroblock is simply the previous tile's last dword column (no copy), and
 *     the store list of a step is FIXED per lane (its rows' first NDW-1 dwords + the previous macroblock's last dword, final
 *     now that this macroblock's left-edge filter has run): one store instruction for the rows, one for the context rows.
 * Rows inside a BAND of W consecutive macroblock rows (one workgroup, wave w = row W b + w; W = 4: one wave per SIMD, the waves
 * are latency-bound and must not share an issue port) hand off through LDS: wave w publishes the bottom CTX rows of each
 * macroblock into a ring of R slots (first NDW-1 dwords at step x, the last dword a step later) and bumps its LDS progress
 * word; wave w+1 spins on it (MB x may start when x + 2 macroblocks of the row above are done), and finishes the CTX-1
 * rows above itself: every picture byte is written once, by the wave that gives it its final value.  A producer never runs
 * more than R macroblocks ahead (it checks the consumer's progress before reusing a slot).  Only the last row of a band talks
 * to the next band through memory, with the frame kernel's protocol (write-through stores, acknowledged, then a device-scope
 * counter).
 * CHROMA: one 4:2:0 chroma plane, 8x8 samples per macroblock, edges at 0 and 4 (the even luma edges, filter_mb_dir
 * h264_loopfilter.c:644-700).  Dword-aligned planes only (the byte path stays on k_h264_deblock_frame).
 */
#define DB_R 16      /* ring slots per row boundary */

__device__ __forceinline__ int db_sad(int a, int b)
{
    int d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

/* one sample line across one edge, in registers, branch-free; v[0..7] = p3 p2 p1 p0 q0 q1 q2 q3.  Same arithmetic as lf_line
 * (h264dsp_template.c:104-330): `if (tc0) p1 += clip(..., -tc0, tc0)` is the unconditional form because the clip range is empty
 * when tc0 == 0. */
template <bool CHROMA>
__device__ __forceinline__ void db_normal(int (&v)[8], int alpha, int beta, int tc0, int en = 1)
{
    const int p2 = v[1], p1 = v[2], p0 = v[3], q0 = v[4], q1 = v[5], q2 = v[6];
    /* `&`, not `&&`: no short-circuit control flow — the lanes of an edge take every path anyway */
    int c = en & (int)(db_sad(p0, q0) < alpha) & (int)(db_sad(p1, p0) < beta) & (int)(db_sad(q1, q0) < beta);
    if (CHROMA) {
        c &= (int)(tc0 > 0);
        const int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc0, tc0);
        v[3] = c ? clip3(p0 + delta, 0, 255) : p0;
        v[4] = c ? clip3(q0 - delta, 0, 255) : q0;
        return;
    }
    c &= (int)(tc0 >= 0);
    const int ap = db_sad(p2, p0) < beta, aq = db_sad(q2, q0) < beta;
    const int avg = (p0 + q0 + 1) >> 1;
    const int dp = clip3(((p2 + avg) >> 1) - p1, -tc0, tc0), dq = clip3(((q2 + avg) >> 1) - q1, -tc0, tc0);
    const int tc = tc0 + ap + aq;
    const int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    v[2] = (c & ap) ? p1 + dp : p1;
    v[5] = (c & aq) ? q1 + dq : q1;
    v[3] = c ? clip3(p0 + delta, 0, 255) : p0;
    v[4] = c ? clip3(q0 - delta, 0, 255) : q0;
}

template <bool CHROMA>
__device__ __forceinline__ void db_intra(int (&v)[8], int alpha, int beta, int en = 1)
{
    const int p3 = v[0], p2 = v[1], p1 = v[2], p0 = v[3], q0 = v[4], q1 = v[5], q2 = v[6], q3 = v[7];
    const int d0 = db_sad(p0, q0);
    const int c = en & (int)(d0 < alpha) & (int)(db_sad(p1, p0) < beta) & (int)(db_sad(q1, q0) < beta);
    const int wp0 = (2 * p1 + p0 + q1 + 2) >> 2, wq0 = (2 * q1 + q0 + p1 + 2) >> 2; /* the weak forms */
    if (CHROMA) {
        v[3] = c ? wp0 : p0;
        v[4] = c ? wq0 : q0;
        return;
    }
    const int strong = d0 < ((alpha >> 2) + 2);
    const int sp = c & strong & (int)(db_sad(p2, p0) < beta), sq = c & strong & (int)(db_sad(q2, q0) < beta);
    const int s4 = p0 + q0;
    /* every candidate value is computed first and made opaque: with the arithmetic visible behind the selects the compiler sinks it
     * into divergent branches (exec-mask bookkeeping around three-instruction blocks) instead of emitting v_cndmask */
    int sp0 = (p2 + 2 * p1 + 2 * s4 + q1 + 4) >> 3, sp1 = (p2 + p1 + s4 + 2) >> 2, sp2 = (2 * p3 + 3 * p2 + p1 + s4 + 4) >> 3;
    int sq0 = (p1 + 2 * s4 + 2 * q1 + q2 + 4) >> 3, sq1 = (s4 + q1 + q2 + 2) >> 2, sq2 = (2 * q3 + 3 * q2 + q1 + s4 + 4) >> 3;
    int w0 = c ? wp0 : p0, w1 = c ? wq0 : q0;
    asm("" : "+v"(sp0), "+v"(sp1), "+v"(sp2), "+v"(sq0), "+v"(sq1), "+v"(sq2), "+v"(w0), "+v"(w1));
    v[3] = sp ? sp0 : w0;
    v[2] = sp ? sp1 : p1;
    v[1] = sp ? sp2 : p2;
    v[4] = sq ? sq0 : w1;
    v[5] = sq ? sq1 : q1;
    v[6] = sq ? sq2 : q2;
}

typedef uint32_t db_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t db_u8 __attribute__((ext_vector_type(8)));
typedef const db_u4 __attribute__((address_space(4))) *db_cc4;
typedef const db_u8 __attribute__((address_space(4))) *db_cc8;

template <bool CHROMA, int DB_W> /* DB_W: waves (macroblock rows) per band */
__global__ __launch_bounds__(64 * DB_W) void k_h264_deblock_band(uint8_t *plane, size_t frame_pitch, ptrdiff_t stride, int mb_w, int mb_h,
                                                                 const FFHipH264Edge *edges, int *gprog, int nbands, int *fail, int fault)
{
    constexpr int MB = CHROMA ? 8 : 16;          /* samples per macroblock side */
    constexpr int CTX = CHROMA ? 2 : 4;          /* context rows above a macroblock */
    constexpr int NDW = MB / 4;                  /* dwords per macroblock row */
    constexpr int NE = CHROMA ? 4 : 8;           /* edge records per macroblock */
    constexpr int TPP = MB + 4;                  /* tile pitch (an odd number of dwords: rows fall into different banks) */
    constexpr int TR = MB + CTX;                 /* tile rows: context rows first */
    constexpr int TSZ = TR * TPP;
    __shared__ __align__(16) uint8_t tiles[DB_W][2 * TSZ];
    __shared__ uint32_t ring[DB_W][DB_R][CTX * NDW];
    __shared__ int lprog[DB_W + 1];

    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int band = blockIdx.x, my = band * DB_W + w;
    plane += (size_t)blockIdx.y * frame_pitch;
    edges += (size_t)blockIdx.y * mb_w * mb_h * NE;
    gprog += (size_t)blockIdx.y * nbands;
    if (lane == 0)
        lprog[w] = 0;
    if (threadIdx.x == 0)
        lprog[DB_W] = 0;
    __syncthreads();
    if (my >= mb_h)
        return;
    uint8_t *tbase = tiles[w];
    uint8_t *rowbase = plane + (ptrdiff_t)my * MB * stride;
    const bool has_below = my + 1 < mb_h;
    const bool to_lds = has_below && w + 1 < DB_W;      /* the row below is this workgroup's */
    const bool to_mem = has_below && w + 1 == DB_W;     /* ... the next workgroup's */
    const bool from_lds = my > 0 && w > 0;              /* else (my > 0): from the band above, through memory */
    const int rlast = to_lds ? MB - CTX : MB - 1;       /* last own row this wave gives its final value */

    /* ---- per-lane constants.  (pr, ps): this lane's row / dword slot of the MB x NDW grid (lanes 0 .. MB NDW - 1) ---- */
    const int pr = lane / NDW, ps = lane % NDW;
    const bool in_mb = lane < MB * NDW;
    const int lown = (pr + CTX) * TPP + 4 * ps;                         /* tile offset of the lane's own dword */
    const ptrdiff_t gown = (ptrdiff_t)pr * stride + 4 * ps;            /* ... its picture offset inside the macroblock */
    /* store list, own rows: slots 0 .. NDW-2 = this macroblock's dwords, slot NDW-1 = the previous macroblock's last dword */
    const bool sprev = ps == NDW - 1;
    const int lst = (pr + CTX) * TPP + (sprev ? MB - 4 : 4 * ps);
    const ptrdiff_t gst = (ptrdiff_t)pr * stride + (sprev ? -4 : 4 * ps);
    const bool st_row = in_mb && pr <= rlast;
    /* store list, context rows -(CTX-1) .. -1: all NDW dwords of this macroblock (lanes 0 .. (CTX-1) NDW - 1) */
    const int cr = lane / NDW + 1;                                      /* tile row 1 .. CTX-1 */
    const bool st_ctx = my > 0 && lane < (CTX - 1) * NDW;
    const int lctx = cr * TPP + 4 * ps;
    const ptrdiff_t gctx = (ptrdiff_t)(cr - CTX) * stride + 4 * ps;
    /* ring (bottom CTX rows = tile rows MB .. MB+CTX-1): lanes 0 .. CTX NDW - 1 */
    const bool in_ring = lane < CTX * NDW;
    const int lring = (MB + lane / NDW) * TPP + (sprev ? MB - 4 : 4 * ps);
    const int tcsh = 8 * (CHROMA ? (lane & 7) >> 1 : (lane & 15) >> 2); /* this line's tc0 byte of an edge record */

    /* The next macroblock's own samples are fetched a step ahead; the wait for them is a vmcnt(0), i.e. it also drains
     * whatever stores are in flight.  So the picture stores of a macroblock are issued at the START of the next step (its tile
     * is still there: two tiles alternate), together with the fetch: by the next wait, a whole macroblock later, they are
     * long acknowledged, and no round trip to L2 sits on the picture's critical path. */
    uint32_t own = 0;
    auto fetch = [&](int mx) {
        if (in_mb)
            own = *reinterpret_cast<const uint32_t *>(rowbase + mx * MB + gown);
    };
    /* picture stores of macroblock m (tile `tm`) and of the last dword column of macroblock m - 1 (tile `tp`): everything they
     * have made final (see the header) */
    auto flush = [&](int m, const uint8_t *tm, const uint8_t *tp, bool row_end) {
        if (fault & 2)
            return;
        uint8_t *mb = rowbase + m * MB;
        if (st_row && !(sprev && m == 0)) {
            const uint32_t v = *reinterpret_cast<const uint32_t *>((sprev ? tp : tm) + lst);
            uint32_t *g = reinterpret_cast<uint32_t *>(mb + gst);
            if (to_mem)
                __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                *g = v;
        }
        if (st_ctx)
            *reinterpret_cast<uint32_t *>(mb + gctx) = *reinterpret_cast<const uint32_t *>(tm + lctx);
        if (row_end && st_row && sprev) { /* the row's last dword column: no macroblock to its right will touch it */
            const uint32_t v = *reinterpret_cast<const uint32_t *>(tm + lst);
            uint32_t *g = reinterpret_cast<uint32_t *>(mb + gst + MB);
            if (to_mem)
                __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                *g = v;
        }
    };
    /* edge records of a macroblock: NE x 3 dwords through the scalar cache */
    const uint32_t *erow = reinterpret_cast<const uint32_t *>(edges + (size_t)my * mb_w * NE);
    uint32_t en[3 * NE];
    auto fetch_edges = [&](int mx) { /* issued at the top of the macroblock's own step: the wait for the row above hides it */
        const uint32_t *p = erow + (size_t)mx * 3 * NE;
        if (CHROMA) {
            const db_u8 a = *(db_cc8)p;
            const db_u4 b = *(db_cc4)(p + 8);
            en[0] = a.s0; en[1] = a.s1; en[2] = a.s2; en[3] = a.s3; en[4] = a.s4; en[5] = a.s5; en[6] = a.s6; en[7] = a.s7;
            en[8] = b.x; en[9] = b.y; en[10] = b.z; en[11] = b.w;
        } else {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const db_u8 a = *(db_cc8)(p + 8 * q);
                en[8 * q] = a.s0; en[8 * q + 1] = a.s1; en[8 * q + 2] = a.s2; en[8 * q + 3] = a.s3;
                en[8 * q + 4] = a.s4; en[8 * q + 5] = a.s5; en[8 * q + 6] = a.s6; en[8 * q + 7] = a.s7;
            }
        }
    };
    int known = 0, kbelow = 0;
    fetch(0);
    for (int mx = 0; mx < mb_w; mx++) {
        const uint8_t *mb = rowbase + mx * MB;
        uint8_t *cur = tbase + (mx & 1) * TSZ, *prev = tbase + ((mx & 1) ^ 1) * TSZ;
        const bool last = mx + 1 == mb_w;
        fetch_edges(mx);
        const uint32_t (&ec)[3 * NE] = en;
        const uint32_t mine = own;   /* the wait for the fetch: everything issued a step ago is complete now */
        if (to_mem && mx >= 3 && !(fault & 1)) {
            /* through memory to the next band: the stores issued at the top of the previous step (macroblock mx - 2's rows,
             * mx - 3's last dword column) are acknowledged: macroblocks < mx - 2 are complete in memory */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            if (lane == 0)
                __hip_atomic_store(&gprog[band], mx - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mx > 0)
            flush(mx - 1, prev, cur, false); /* before the tile of macroblock mx - 2 (`cur`) is overwritten */
        if (in_mb)
            *reinterpret_cast<uint32_t *>(cur + lown) = mine;
        if (!last)
            fetch(mx + 1);
        wave_lds_sync();
        /* ---- vertical edges, left to right: lane = row; samples -4 .. MB-1 of the row ---- */
        if (lane < MB && !(fault & 4)) {
            uint8_t *trow = cur + (lane + CTX) * TPP, *tprev = prev + (lane + CTX) * TPP + MB - 4;
            int x[MB + 4];
#pragma unroll
            for (int d = 0; d < NDW + 1; d++) {
                const uint32_t v = d ? *reinterpret_cast<const uint32_t *>(trow + 4 * d - 4) : *reinterpret_cast<const uint32_t *>(tprev);
                x[4 * d] = v & 255; x[4 * d + 1] = (v >> 8) & 255; x[4 * d + 2] = (v >> 16) & 255; x[4 * d + 3] = v >> 24;
            }
#pragma unroll
            for (int k = 0; k < NDW; k++) {
                const uint32_t rec = ec[3 * k + 1], tcw = ec[3 * k + 2];
                const int alpha = (rec >> 8) & 255, beta = (rec >> 16) & 255;
                if (alpha && beta && !(k == 0 && mx == 0)) { /* scalar */
                    int v[8] = { x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3], x[4 * k + 4], x[4 * k + 5], x[4 * k + 6], x[4 * k + 7] };
                    if ((rec & 255) >= 4)
                        db_intra<CHROMA>(v, alpha, beta);
                    else
                        db_normal<CHROMA>(v, alpha, beta, (int)(int8_t)(tcw >> tcsh));
#pragma unroll
                    for (int i = 1; i < 7; i++)
                        x[4 * k + i] = v[i];
                }
            }
#pragma unroll
            for (int d = 0; d < NDW + 1; d++) {
                const uint32_t v = (uint32_t)x[4 * d] | ((uint32_t)x[4 * d + 1] << 8) | ((uint32_t)x[4 * d + 2] << 16) | ((uint32_t)x[4 * d + 3] << 24);
                if (d)
                    *reinterpret_cast<uint32_t *>(trow + 4 * d - 4) = v;
                else
                    *reinterpret_cast<uint32_t *>(tprev) = v;
            }
        }
        /* ---- context rows: the row above must have finished macroblock mx + 1.  Only the horizontal edges read (and rewrite)
         *      them, so the wait sits behind the vertical pass, which it overlaps ---- */
        if (my > 0) {
            /* LDS progress counts finished steps (the ring slot's last dword arrives a step late); the memory progress counts
             * macroblocks complete in memory */
            const int want = (fault & 8) ? 0 : from_lds ? min(mx + 2, mb_w) : mx + 1;
            int spins = 0;
            while (known < want) {
                known = from_lds ? __hip_atomic_load(&lprog[w - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                 : __hip_atomic_load(&gprog[band - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (known >= want)
                    break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 24)) { /* never in a correct run; do not hang the device */
                    if (lane == 0)
                        __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (in_ring) {
                uint32_t v;
                if (from_lds)
                    v = ring[w - 1][mx % DB_R][lane];
                else /* written write-through by the band above: device-scope (L1-bypassing) loads */
                    v = __hip_atomic_load(reinterpret_cast<const uint32_t *>(mb + (ptrdiff_t)(lane / NDW - CTX) * stride + 4 * ps),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *reinterpret_cast<uint32_t *>(cur + (lane / NDW) * TPP + 4 * ps) = v;
            }
        }
        wave_lds_sync();
        /* ---- horizontal edges, top to bottom: lane = column; y[i] = row i - 4 (chroma: rows -2 .. 7, y[0], y[1] unused) ---- */
        if (lane < MB && !(fault & 4)) {
            uint8_t *tcol = cur + lane;
            int y[MB + 4];
#pragma unroll
            for (int r = 0; r < MB + 4; r++)
                y[r] = r >= 4 - CTX ? tcol[(r - (4 - CTX)) * TPP] : 0;
#pragma unroll
            for (int k = 0; k < NDW; k++) {
                const uint32_t rec = ec[3 * (NDW + k) + 1], tcw = ec[3 * (NDW + k) + 2];
                const int alpha = (rec >> 8) & 255, beta = (rec >> 16) & 255;
                if (alpha && beta && !(k == 0 && my == 0)) {
                    int v[8] = { y[4 * k], y[4 * k + 1], y[4 * k + 2], y[4 * k + 3], y[4 * k + 4], y[4 * k + 5], y[4 * k + 6], y[4 * k + 7] };
                    if ((rec & 255) >= 4)
                        db_intra<CHROMA>(v, alpha, beta);
                    else
                        db_normal<CHROMA>(v, alpha, beta, (int)(int8_t)(tcw >> tcsh));
#pragma unroll
                    for (int i = 1; i < 7; i++)
                        y[4 * k + i] = v[i];
                }
            }
            /* rows a horizontal filter can have changed: all but the first context row and the last row of the block */
#pragma unroll
            for (int r = 5 - CTX; r < MB + 3; r++)
                tcol[(r - (4 - CTX)) * TPP] = (uint8_t)y[r];
        }
        wave_lds_sync();
        /* ---- the row below, inside the band: bottom CTX rows into the ring ---- */
        if (to_lds) {
            if (mx >= DB_R) { /* slot reuse: the consumer must be done with macroblock mx - DB_R */
                int spins = 0;
                while (kbelow < mx - DB_R + 1) {
                    kbelow = __hip_atomic_load(&lprog[w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (kbelow >= mx - DB_R + 1)
                        break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) {
                        if (lane == 0)
                            __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        return;
                    }
                }
            }
            if (in_ring) {
                if (!sprev)
                    ring[w][mx % DB_R][lane] = *reinterpret_cast<const uint32_t *>(cur + lring);
                else if (mx > 0)
                    ring[w][(mx - 1) % DB_R][lane] = *reinterpret_cast<const uint32_t *>(prev + lring);
                if (last && sprev)
                    ring[w][mx % DB_R][lane] = *reinterpret_cast<const uint32_t *>(cur + lring);
            }
        }
        /* ---- publish (fault & 1: test hook — rows never publish, so the row below must time out and report) ---- */
        if (fault & 1)
            continue;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) /* rows that hand off through memory publish here too: the row above paces its ring by this word */
            __hip_atomic_store(&lprog[w], mx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    /* the last macroblock's stores, its row-end dword column included */
    wave_lds_sync();
    flush(mb_w - 1, tbase + ((mb_w - 1) & 1) * TSZ, tbase + (((mb_w - 1) & 1) ^ 1) * TSZ, true);
    if (to_mem && !(fault & 1)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0)
            __hip_atomic_store(&gprog[band], mb_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/*
 * k_h264_deblock_skew — the decoder-order wavefront with the data movement built first (round 3).
 *
 * The band kernel above spends a wave (16-20 useful lanes) per macroblock row and moves the picture as per-lane dwords; its
 * load/store skeleton alone streamed at 7 % of HBM (profiles/r02_deblock_parts.txt).  Here ONE wave owns Q = 4 consecutive
 * macroblock rows (8 for a chroma plane) and walks them SKEWED: lanes 16 q .. 16 q + 15 are row q of the band, and at wave step
 * s row q filters macroblock x = s - 2 q — exactly the lag the order demands (macroblock (x, y) needs (x + 1, y - 1) finished,
 * libavcodec/h264_loopfilter.c:716 with the raster walk of h264_slice.c's loop_filter()).  So
 *   - all 64 lanes filter in every pass (lane = row of its macroblock for the vertical edges, lane = column for the horizontal);
 *   - three of four row hand-offs are free: the rows of a band share one LDS strip (Q x 16 rows + 4 context rows, a ring of 8
 *     macroblock slots wide, 144-byte pitch + 16 bytes of skew per row group: the row passes' 16-byte accesses and the column
 *     passes' byte accesses both spread over the banks), a row's top context simply IS the bottom of the row group above;
 *   - the picture moves as 16-byte rows: a lane loads its macroblock row a step ahead (one global_load_dwordx4 per lane and
 *     step) and stores, a step behind, the row of the PREVIOUS macroblock's columns shifted up by four rows — the 16 x 16 region
 *     (x - 1, rows 16 y - 4 .. 16 y + 11) is exactly what macroblock (x, y) has just made final: every picture byte is
 *     written once, as part of a 16-byte row, by the row group that gives it its final value;
 *   - only a band's last row talks to the next band through memory: its bottom four rows go out as write-through dwords, the
 *     counter follows a step later (the stores are a whole step old and acknowledged by then: no round trip waits on the
 *     critical path), and the consumer reads them with device-scope loads — the band kernel's protocol, once per Q rows.
 * A picture's bands sit on ONE XCD (block L -> XCD L & 7: picture f on XCD f & 7), so the hand-off traffic of a picture stays in
 * one L2 and eight pictures run side by side on the eight XCDs.
 * Needs 16-byte (chroma: 8-byte) aligned planes / strides / pitches and 16-byte aligned edge records; anything else takes the
 * band kernel (4-byte aligned) or the row kernel.
 */
/* ---- the skewed-rows kernel's edge filter ----------------------------------------------------------
 * A wave that is alone on its SIMD issues ONE instruction every four cycles, scalar or vector, and a step of the wavefront is eight
 * DEPENDENT edges: the filter is written for the fewest instructions, not for the fewest operations.
 *   - every comparison of h264dsp_template.c:104-330 is a sign: |a - b| < t  <=>  v_sad_u32(a, b, -t) < 0, and a conjunction is the
 *     sign of a maximum (v_max3_i32) — no compare / s_and chains, no exec-mask control flow;
 *   - alpha == 0 or beta == 0 (a bS = 0 edge) disables itself: |a - b| - 0 is never negative;
 *   - clips are v_med3_i32 (lanes whose range is empty, tc0 < 0, are deselected anyway);
 *   - the bS = 4 filter runs behind one wave-uniform branch and overrides the lanes it owns from the ORIGINAL samples.
 * v[0..7] = p3 p2 p1 p0 q0 q1 q2 q3 of one line; rec = {bS, alpha, beta, -} bytes, tcw = the edge's four tc0 bytes. */
__device__ __forceinline__ int db_sad3(int a, int b, int c)
{
    int d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int db_med3(int a, int lo, int hi)
{
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(lo), "v"(hi));
    return d;
}
__device__ __forceinline__ int db_clip255(int a)
{
    int d;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(d) : "v"(a), "s"(255));
    return d;
}
__device__ __forceinline__ int db_max3(int a, int b, int c)
{
    int d;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

/* sh = bit depth - 8: alpha, beta and tc0 arrive in 8-bit units and are scaled as h264dsp_template.c:104-330 scales them (alpha, beta
 * << sh; luma tc0 * (1 << sh); chroma ((tc0 - 1) << sh) + 1); maxv = 2^depth - 1 */
template <bool CHROMA>
__device__ __forceinline__ void db_edge(int (&v)[8], uint32_t rec, uint32_t tcw, int tcsh, bool skip, int sh = 0, int maxv = 255)
{
    const int p3 = v[0], p2 = v[1], p1 = v[2], p0 = v[3], q0 = v[4], q1 = v[5], q2 = v[6], q3 = v[7];
    const int alpha = (int)((rec >> 8) & 255) << sh, negb = -((int)((rec >> 16) & 255) << sh);
    const int nega = skip ? 0 : -alpha;                       /* a picture edge: never filtered */
    const int tc8 = __builtin_amdgcn_sbfe(tcw, tcsh, 8);
    const int tc0 = CHROMA ? (tc8 - 1) * (1 << sh) + 1 : tc8 * (1 << sh);
    const bool is4 = (rec & 255) >= 4;
    /* m < 0: |p0 - q0| < alpha && |p1 - p0| < beta && |q1 - q0| < beta */
    const int m = db_max3(db_sad3(p0, q0, nega), db_sad3(p1, p0, negb), db_sad3(q1, q0, negb));
    const int mn = max(m, CHROMA ? -tc0 : ~tc0);              /* ... && tc0 > 0 (chroma) / tc0 >= 0 (luma): the bS < 4 filter's lanes */
    {   /* the bS < 4 filter, every lane (four macroblocks of different rows share an instruction: they are rarely all bS = 0, and a
         * wave-uniform skip costs register copies on both paths) */
        const int x4 = ((q0 - p0) << 2) + (p1 - q1) + 4;
        if (CHROMA) {
            const int delta = (mn >> 31) & db_med3(x4 >> 3, -tc0, tc0);
            v[3] = db_med3(p0 + delta, 0, maxv);
            v[4] = db_med3(q0 - delta, 0, maxv);
        } else {
            const int dap = db_sad3(p2, p0, negb), daq = db_sad3(q2, q0, negb);     /* < 0: |p2 - p0| < beta */
            const int avg = (p0 + q0 + 1) >> 1, ntc0 = -tc0;
            const int dp = db_med3(((p2 + avg) >> 1) - p1, ntc0, tc0), dq = db_med3(((q2 + avg) >> 1) - q1, ntc0, tc0);
            const int tc = tc0 + (int)((uint32_t)dap >> 31) + (int)((uint32_t)daq >> 31);
            const int delta = (mn >> 31) & db_med3(x4 >> 3, -tc, tc);
            v[2] = p1 + (dp & (max(mn, dap) >> 31));
            v[5] = q1 + (dq & (max(mn, daq) >> 31));
            v[3] = db_med3(p0 + delta, 0, maxv);
            v[4] = db_med3(q0 - delta, 0, maxv);
        }
    }
    const int mi = is4 ? m : 0;                               /* < 0: a bS = 4 line that passes the alpha / beta test */
    if (__builtin_amdgcn_ballot_w64(mi < 0)) {
        const int wp0 = (2 * p1 + p0 + q1 + 2) >> 2, wq0 = (2 * q1 + q0 + p1 + 2) >> 2;   /* the weak forms */
        if (CHROMA) {
            v[3] = mi < 0 ? wp0 : v[3];
            v[4] = mi < 0 ? wq0 : v[4];
        } else {
            const int ds = db_sad3(p0, q0, -((alpha >> 2) + 2));                        /* < 0: the strong filter */
            const int msp = db_max3(mi, ds, db_sad3(p2, p0, negb)), msq = db_max3(mi, ds, db_sad3(q2, q0, negb));
            const int s4 = p0 + q0, ep = p1 + s4, eq = q1 + s4;
            int sp0 = (2 * ep + p2 + q1 + 4) >> 3, sp1 = (p2 + ep + 2) >> 2, sp2 = (2 * (p3 + p2) + p2 + ep + 4) >> 3;
            int sq0 = (2 * eq + q2 + p1 + 4) >> 3, sq1 = (q2 + eq + 2) >> 2, sq2 = (2 * (q3 + q2) + q2 + eq + 4) >> 3;
            int w0 = mi < 0 ? wp0 : v[3], w1 = mi < 0 ? wq0 : v[4], w2 = mi < 0 ? p1 : v[2], w3 = mi < 0 ? q1 : v[5];
            /* every candidate first, opaque: with the arithmetic visible behind the selects the compiler sinks it into divergent
             * branches (exec-mask bookkeeping around three-instruction blocks) instead of emitting v_cndmask */
            asm("" : "+v"(sp0), "+v"(sp1), "+v"(sp2), "+v"(sq0), "+v"(sq1), "+v"(sq2), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
            v[3] = msp < 0 ? sp0 : w0;
            v[2] = msp < 0 ? sp1 : w2;
            v[1] = msp < 0 ? sp2 : p2;
            v[4] = msq < 0 ? sq0 : w1;
            v[5] = msq < 0 ? sq1 : w3;
            v[6] = msq < 0 ? sq2 : q2;
        }
    }
}

#ifndef DB_SKEW
#define DB_SKEW 1
#endif
/* spin on an LDS counter of another wave of the workgroup; false (and the launch's fail flag) after 2^22 polls — never in a correct run */
__device__ __forceinline__ bool db_wait_lds(const int *ctr, int want, int *fail)
{
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) {
            if ((threadIdx.x & 63) == 0)
                __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}
/* the pictures of a launch that do not sit at a constant pitch (round 4: the picture objects of a batched flush, each with its own
 * planes and its own edge records): picture f's plane and records by table */
struct FFHipDbPtrs { uint8_t *plane[FFHIP_DB_PTRS]; const FFHipH264Edge *edges[FFHIP_DB_PTRS]; };
template <bool CHROMA, typename PIX = uint8_t>
__global__ __launch_bounds__(256) void k_h264_deblock_skew(uint8_t *plane, size_t frame_pitch, ptrdiff_t stride, int mb_w, int mb_h,
                                                         const FFHipH264Edge *edges, int *gprog, int nbands, int bwaves, int nframes,
                                                         int *fail, int fault, int xrot, int bd, FFHipDbPtrs PT, int use_ptrs)
{
    constexpr int MB = CHROMA ? 8 : 16;          /* samples per macroblock side = lanes per row group */
    constexpr int PS = (int)sizeof(PIX), MBB = MB * PS; /* bytes per sample (uint16_t above 8 bits), per macroblock row */
    const int sh = PS == 1 ? 0 : bd - 8, maxv = PS == 1 ? 255 : (1 << bd) - 1;
    constexpr int NDW = MB / 4;                  /* dwords per macroblock row = edges per direction */
    constexpr int NE = 2 * NDW;                  /* edge records per macroblock */
    constexpr int Q = 64 / MB;                   /* macroblock rows per wave */
    constexpr int SL = 8;                        /* ring slots (macroblocks) per row */
    constexpr int SK = DB_SKEW;                  /* macroblocks a row group trails the one above */
    constexpr int PITCH = SL * MBB + 16;         /* 8 bits: 144 / 80 bytes = 36 / 20 dwords, 16 (8) rows spread over all banks */
    /* A workgroup is W = 1 .. 4 waves on the SIMDs of one CU working on W CONSECUTIVE bands ("super-band") in ONE LDS strip: wave w's
     * top context is the bottom of wave w - 1's last row group, in place, exactly as between the row groups of a wave — the
     * hand-off between the waves of a workgroup is an LDS counter (a few hundred ns), only the workgroup's first / last wave talk
     * through memory (a write-through store, its acknowledgement, a counter, a poll and a load: several microseconds and six
     * steps of lag per hand-off, which is what a lone picture's time was made of). */
    constexpr int ETAB_BYTES = Q * (3 * NE + 4) * 4;
    extern __shared__ __align__(16) uint8_t db_lds[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), W = (int)(blockDim.x >> 6);
    const int strip_bytes = ((4 + W * Q * MB) * PITCH + (W * Q + 1) * 16 + 15) & ~15;
    uint8_t *const tile = db_lds + wv * Q * (MB * PITCH + 16);   /* this wave's rows: local row r, group q -> r * PITCH + (q + 1) * 16 */
    uint32_t (*const etab)[3 * NE + 4] = reinterpret_cast<uint32_t (*)[3 * NE + 4]>(db_lds + strip_bytes + wv * ETAB_BYTES);
    /* vdone[w] / hdone[w]: steps of wave w whose vertical pass / whole step are complete, counted over all of its super-bands */
    int *const vdone = reinterpret_cast<int *>(db_lds + strip_bytes + W * ETAB_BYTES), *const hdone = vdone + W;
    if (threadIdx.x < 2 * (unsigned)W)
        vdone[threadIdx.x] = 0;
    __syncthreads();

    /* block -> (picture, first super-band): the bands of a picture on one XCD.  A picture gets bwaves / W workgroups; workgroup j
     * takes super-bands j, j + bwaves / W, ...  (Super-band b's predecessor belongs to a workgroup dispatched no later, or to an
     * earlier pass of the same one: no wave waits on the unborn.) */
    /* xrot: the launcher's running count — a lone picture per launch (the picture pipeline: one launch per plane, many pictures in
     * flight on their own streams) would otherwise always land on XCD 0 */
    const int L = blockIdx.x, xcd = (L - xrot) & 7, nwg = bwaves / W; /* bwaves is a multiple of the waves per workgroup */
    const int f = xcd + 8 * ((L >> 3) / nwg), sb0 = (L >> 3) % nwg;
    if (f >= nframes)
        return;
    if (use_ptrs) { /* (read once, here: the table is indexed at run time) */
        plane = PT.plane[f];
        edges = PT.edges[f];
    } else {
        plane += (size_t)f * frame_pitch;
        edges += (size_t)f * mb_w * mb_h * NE;
    }
    gprog += (size_t)f * nbands;

    const int lane = threadIdx.x & 63, q = lane / MB, l = lane % MB;
    const int nsteps = mb_w + 2 + SK * (Q - 1); /* a row runs two steps past its last macroblock: the columns of x - 2 leave at step x */
    int base = 0;                                /* this wave's step count before the current super-band */
    for (int sb = sb0; sb * W < nbands; sb += nwg, base += nsteps) {
    const int band = sb * W + wv;
    if (band < nbands) {
    const bool from_mem = wv == 0 && band > 0, from_lds = wv > 0;
    const int y = band * Q + q;
    const bool row_ok = y < mb_h;
    const int qb = min(Q - 1, mb_h - 1 - band * Q);      /* the band's last row inside the picture (wave-uniform) */
    const bool next_band = band + 1 < nbands;
    const int tcsh = 8 * (CHROMA ? l >> 1 : l >> 2);     /* this line's tc0 byte of an edge record (row l / column l) */
    /* LDS: local row r' = picture row - (band's first row) + 4; rows of group q carry skew (q + 1) * 16, the 4 context rows 0 */
    uint8_t *const ownrow = tile + (MB * q + l + 4) * PITCH + (q + 1) * 16;       /* V pass: this lane's macroblock row */
    uint8_t *const strow = tile + (MB * q + l) * PITCH + (q + (l >= 4 ? 1 : 0)) * 16; /* store pass: the row four above it */
    const uint8_t *const grow = plane + ((ptrdiff_t)y * MB + l) * stride;          /* picture row of ownrow */
    uint8_t *const gst = plane + ((ptrdiff_t)y * MB + l - 4) * stride;             /* picture row of strow */
    const bool to_mem = next_band && wv == W - 1, to_lds = next_band && wv < W - 1;
    const bool st_ok = row_ok && (y > 0 || l >= 4), bot_ok = row_ok && q == qb && !to_lds; /* the wave below stores them with its rows */
    /* bottom rows of the band's last row: 4 rows x NDW groups of 4 samples = MB lanes (row group qb) */
    const int br = MB - 4 + l / NDW, bq = l % NDW;
    uint8_t *const botl = tile + (MB * qb + br + 4) * PITCH + (qb + 1) * 16 + 4 * PS * bq;
    uint8_t *const botg = plane + ((ptrdiff_t)(band * Q + qb) * MB + br) * stride + 4 * PS * bq;
    /* top context of the band (rows -4 .. -1 of its first row): the same MB lanes of row group 0 */
    const int cr = l / NDW, cd = l % NDW;
    uint8_t *const ctxl = tile + cr * PITCH + 4 * PS * cd;
    const uint8_t *const ctxg = plane + ((ptrdiff_t)band * Q * MB - 4 + cr) * stride + 4 * PS * cd;
    const uint32_t *const erow = reinterpret_cast<const uint32_t *>(edges + (size_t)(row_ok ? y : 0) * mb_w * NE);

    typedef uint32_t rowv_n __attribute__((ext_vector_type(MBB / 4)));
    typedef rowv_n __attribute__((aligned(MBB < 16 ? MBB : 16))) rowv; /* a row of 16-bit luma is two 16-byte pieces */
    typedef uint32_t quadv_n __attribute__((ext_vector_type(PS)));          /* four samples */
    typedef quadv_n __attribute__((aligned(4))) quadv;
    rowv own = {};
    db_u4 eown = { 0, 0, 0, 0 };
    auto fetch = [&](int xn) { /* macroblock xn's row l and (lanes l < 3 NDW / 2) 16 bytes of its edge records */
        if (row_ok && xn >= 0 && xn < mb_w) {
            own = *reinterpret_cast<const rowv *>(grow + xn * MBB);
            if (l < 3 * NE / 4)
                eown = *reinterpret_cast<const db_u4 *>(erow + (size_t)xn * 3 * NE + 4 * l);
        }
    };
    int known = 0;
    bool have_top = false;
    uint32_t topv[PS] = {};
    auto load_top = [&](int x0) {
#pragma unroll
        for (int i = 0; i < PS; i++)
            topv[i] = __hip_atomic_load(reinterpret_cast<const uint32_t *>(ctxg + x0 * MBB) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    fetch(0 - SK * q);
    for (int s = 0; s < nsteps; s++) {
        const int x = s - SK * q;
        const bool act = row_ok && x >= 0 && x < mb_w;
        /* ---- everything issued a step ago is complete: the loads of this step's macroblocks and the stores of the previous
         *      step's results.  The band's bottom rows of macroblocks < xb - 2 are therefore in memory: publish ---- */
        __builtin_amdgcn_s_waitcnt(0);
        const int xb = s - SK * qb; /* the stores issued a step ago (at xb - 1) held the columns of macroblock xb - 3 */
        if (to_mem && xb >= 3 && xb <= mb_w + 2 && !(fault & 1)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0)
                __hip_atomic_store(&gprog[band], xb - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        /* ---- the edge records of this step's macroblocks: staged by the lanes that loaded them, read back (broadcast) by every
         *      lane of the row group; LDS operations of one wave execute in order, nothing to wait for in between ---- */
        if (act && l < 3 * NE / 4)
            *reinterpret_cast<db_u4 *>(&etab[q][4 * l]) = eown;
        const rowv cur = own;                        /* this step's macroblock row: straight from the registers it was loaded into */
        wave_lds_sync();
        uint32_t erec[NE], etcw[NE];
#pragma unroll
        for (int k = 0; k < NE; k++) {
            erec[k] = etab[q][3 * k + 1];
            etcw[k] = etab[q][3 * k + 2];
        }
        uint8_t *const pl = ownrow + ((x - 1) & (SL - 1)) * MBB + MBB - 4 * PS, *const pm = ownrow + (x & (SL - 1)) * MBB;
        const quadv lw = *reinterpret_cast<const quadv *>(pl);   /* the left macroblock's last four samples of this row */
        fetch(x + 1);
        /* ---- picture stores of the previous step's results: macroblock x - 1 was filtered a step ago, which made the columns of
         *      macroblock x - 2 final (rows shifted up by four); behind the row's last macroblock its own columns are final too ---- */
        {
            const int m = x - 2;
            if (!(fault & 2) && m >= 0 && m < mb_w) {
                if (st_ok)
                    *reinterpret_cast<rowv *>(gst + m * MBB) = *reinterpret_cast<const rowv *>(strow + (m & (SL - 1)) * MBB);
                if (bot_ok) /* ... and the bottom four rows of the band's last row: write-through, the next band reads them */
#pragma unroll
                    for (int i = 0; i < PS; i++)
                        __hip_atomic_store(reinterpret_cast<uint32_t *>(botg + m * MBB) + i, reinterpret_cast<const uint32_t *>(botl + (m & (SL - 1)) * MBB)[i],
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        wave_lds_sync();
        /* ---- the ring is SL macroblocks long: the last row group is about to overwrite the column of macroblock x - SL, whose bottom
         *      rows the wave below reads until its step x - SL + 2 (two steps behind its top edge: the store pass) ---- */
        if (to_lds && !(fault & 8) && !db_wait_lds(&hdone[wv + 1], base + s - SK * qb - SL + 3, fail))
            return;
        /* ---- vertical edges, left to right: lane = row; samples -4 .. MB-1 of the row, in registers throughout ---- */
        if (act) {
            int v0[MB + 4];
            constexpr uint32_t SM = PS == 1 ? 0xFFu : 0xFFFFu; /* sample i of a run of dwords: dword i PS / 4, bit 8 (i PS % 4) */
#pragma unroll
            for (int i = 0; i < 4; i++)
                v0[i] = (int)((lw[(i * PS) >> 2] >> (8 * ((i * PS) & 3))) & SM);
#pragma unroll
            for (int i = 0; i < MB; i++)
                v0[4 + i] = (int)((cur[(i * PS) >> 2] >> (8 * ((i * PS) & 3))) & SM);
            if (!(fault & 4)) {
#pragma unroll
                for (int k = 0; k < NDW; k++) {
                    int v[8] = { v0[4 * k], v0[4 * k + 1], v0[4 * k + 2], v0[4 * k + 3], v0[4 * k + 4], v0[4 * k + 5], v0[4 * k + 6], v0[4 * k + 7] };
                    db_edge<CHROMA>(v, erec[k], etcw[k], tcsh, k == 0 && x == 0, sh, maxv);
#pragma unroll
                    for (int i = 1; i < 7; i++)
                        v0[4 * k + i] = v[i];
                }
            }
            quadv lo = {};
#pragma unroll
            for (int i = 0; i < 4; i++)
                lo[(i * PS) >> 2] |= (uint32_t)v0[i] << (8 * ((i * PS) & 3));
            *reinterpret_cast<quadv *>(pl) = lo;
            rowv mw = {};
#pragma unroll
            for (int i = 0; i < MB; i++)
                mw[(i * PS) >> 2] |= (uint32_t)v0[4 + i] << (8 * ((i * PS) & 3));
            *reinterpret_cast<rowv *>(pm) = mw;
        }
        if (to_lds) { /* this step's vertical pass is in LDS (one wave's LDS operations execute in order: the counter follows the rows) */
            wave_lds_sync();
            if (lane == 0 && !(fault & 1)) /* fault & 1: the test hook — no hand-off is published, the waiting wave must time out and report */
                __hip_atomic_store(&vdone[wv], base + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        /* ---- the band's top context.  From the wave above, in place: its last row group must have filtered the left edge of
         *      macroblock s + 1 (its step s + 1 + SK (Q - 1)) ---- */
        if (from_lds && s < mb_w && !(fault & 8) && !db_wait_lds(&vdone[wv - 1], base + s + 2 + SK * (Q - 1), fail))
            return;
        /* ---- from the workgroup above, through memory: it must have its bottom rows of macroblock s in memory (count >= s + 1).  Only
         *      the horizontal edges of row group 0 read (and rewrite) them, so the wait sits behind the vertical pass ---- */
        if (from_mem && s < mb_w) {
            if (!have_top) {
                const int want = (fault & 8) ? 0 : s + 1;
                int spins = 0;
                while (known < want) {
                    known = __hip_atomic_load(&gprog[band - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (known >= want)
                        break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) { /* never in a correct run; do not hang the device */
                        if (lane == 0)
                            __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        return;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (lane < MB)
                    load_top(s);
            }
            if (lane < MB) {
#pragma unroll
                for (int i = 0; i < PS; i++)
                    reinterpret_cast<uint32_t *>(ctxl + (s & (SL - 1)) * MBB)[i] = topv[i];
            }
        }
        wave_lds_sync();
        /* the next macroblock's context, if the band above has already published it: its latency hides behind the H pass */
        have_top = false;
        if (from_mem && s + 1 < mb_w && known >= s + 2) {
            if (lane < MB)
                load_top(s + 1);
            have_top = true;
        }
        /* ---- horizontal edges, top to bottom: lane = column; yv[i] = row i - 4 of the macroblock ---- */
        if (act && !(fault & 4)) {
            uint8_t *tcol = tile + MB * q * PITCH + q * 16 + (x & (SL - 1)) * MBB + l * PS;   /* row -4 of group q (skew of group q - 1) */
            int yv[MB + 4];
#pragma unroll
            for (int r = 0; r < MB + 4; r++)
                yv[r] = (CHROMA && r < 2) ? 0 : (int)*reinterpret_cast<const PIX *>(tcol + r * PITCH + (r >= 4 ? 16 : 0));
#pragma unroll
            for (int k = 0; k < NDW; k++) {
                int v[8] = { yv[4 * k], yv[4 * k + 1], yv[4 * k + 2], yv[4 * k + 3], yv[4 * k + 4], yv[4 * k + 5], yv[4 * k + 6], yv[4 * k + 7] };
                db_edge<CHROMA>(v, erec[NDW + k], etcw[NDW + k], tcsh, k == 0 && y == 0, sh, maxv);
#pragma unroll
                for (int i = 1; i < 7; i++)
                    yv[4 * k + i] = v[i];
            }
            /* rows a horizontal filter can have changed: luma p2 .. q2 of every edge, chroma p0 / q0 */
#pragma unroll
            for (int r = 1; r < MB + 3; r++)
                if (!CHROMA || (r & 3) == 3 || (r & 3) == 0)
                    *reinterpret_cast<PIX *>(tcol + r * PITCH + (r >= 4 ? 16 : 0)) = (PIX)yv[r];
        }
        wave_lds_sync();
        if (from_lds && lane == 0)
            __hip_atomic_store(&hdone[wv], base + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    /* the last step's stores (the flush of the band's last row) are out: publish the whole row */
    if (to_mem && !(fault & 1)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0)
            __hip_atomic_store(&gprog[band], mb_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wave_lds_sync();
    } /* band < nbands */
    __syncthreads(); /* the strip is reused: every wave is through with this super-band */
    } /* super-bands of this workgroup */
}

/* the progress counters come from the per-device pool (progress_pool.hip): a slot per launch, zeroed in stream order */
static int deblock_frames(bool chroma, uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                          const FFHipH264Edge *edges, hipStream_t stream, int bd = 8, uint8_t *const *planes = nullptr,
                          const FFHipH264Edge *const *edge_tabs = nullptr)
{
    if (mb_w <= 0 || mb_h <= 0 || nframes <= 0)
        return 0;
    if (planes) { /* picture f's plane and edge records by table: the skewed-rows kernel only; the checks below see the OR of all */
        uintptr_t al = 0, ale = 0;
        for (int f = 0; f < nframes; f++) {
            if (!planes[f] || !edge_tabs || !edge_tabs[f])
                return FFHIP_EINVAL;
            al |= (uintptr_t)planes[f];
            ale |= (uintptr_t)edge_tabs[f];
        }
        plane = reinterpret_cast<uint8_t *>(al);
        edges = reinterpret_cast<const FFHipH264Edge *>(ale);
        frame_pitch = 0;
    }
    const bool aligned = !(((uintptr_t)plane | (size_t)stride | frame_pitch) & 3);
    if (chroma && !aligned) {
        ffhip_set_error("ffhip_h264_deblock_frame_chroma: plane, stride and frame pitch must be 4-byte aligned");
        return FFHIP_EINVAL;
    }
    const char *eo = FFHIP_KNOB("FFHIP_DEBLOCK_OLD"); /* 1: the per-row-workgroup kernel, 2: the band kernel (measurement / cross-check) */
    const char *ef = FFHIP_KNOB("FFHIP_DEBLOCK_FAULT"); /* test hook: lost hand-offs -> timeout -> FFHIP_EIO at the next check */
    const int fault = ef ? atoi(ef) : 0; /* 1: the test hook; 2 no stores, 4 no filters, 8 no waiting: timing experiments (wrong output) */
    const int old = eo ? atoi(eo) : 0;
    /* the skewed-rows kernel moves 16-byte (chroma: 8-byte) rows and reads the edge records as 16-byte vectors */
    const size_t amask = chroma && bd == 8 ? 7 : 15;
    const bool skew = (!old || bd > 8) && !(((uintptr_t)plane | (size_t)stride | frame_pitch) & amask) && !((uintptr_t)edges & 15);
    if (bd > 8 && !skew) { /* the 4-byte aligned and byte paths below are 8-bit kernels */
        ffhip_set_error("ffhip_h264_deblock_frame (bit depth %d): plane, stride, frame pitch and edge records must be 16-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    const bool band = !skew && aligned && !(old == 1 && !chroma);
    /* band kernel, rows per band.  A lone picture is latency-bound: 4 = one wave per SIMD, the waves of a band must not share an
     * issue port.  A batch that fills the chip anyway is throughput-bound: 16 keeps 15 of 16 hand-offs in LDS.
     * FFHIP_DEBLOCK_BAND = 4 / 8 / 16 overrides. */
    const char *ew = FFHIP_KNOB("FFHIP_DEBLOCK_BAND");
    const int bw = skew ? (chroma ? 8 : 4) : ew && atoi(ew) == 16 ? 16 : ew && atoi(ew) == 8 ? 8 : ew && atoi(ew) == 4 ? 4 :
                   (long long)nframes * mb_h > 2048 ? 16 : 4;
    const int nbands = cdiv(mb_h, bw);
    const int per_frame = (band || skew) ? nbands : mb_h + 1;
    if (per_frame > FFHIP_PROGRESS_SLOT_INTS) {
        ffhip_set_error("ffhip_h264_deblock_frame: %d macroblock rows exceed the supported %d", mb_h, FFHIP_PROGRESS_SLOT_INTS - 1);
        return FFHIP_EINVAL;
    }
    const int ne = chroma ? 4 : 8;
    if (planes && !skew) {
        ffhip_set_error("ffhip_h264_deblock: a batch of separate pictures needs 16-byte aligned planes, strides and edge records");
        return FFHIP_EINVAL;
    }
    int per_launch = FFHIP_PROGRESS_SLOT_INTS / per_frame; /* frames whose counters fit one pool slot */
    if (planes && per_launch > FFHIP_DB_PTRS)
        per_launch = FFHIP_DB_PTRS;
    for (int f0 = 0; f0 < nframes; f0 += per_launch) {
        const int nf = nframes - f0 < per_launch ? nframes - f0 : per_launch;
        FFHipProgressSlot ps;
        const int r = ffhip_progress_acquire(nf * per_frame, stream, &ps);
        if (r < 0)
            return r;
        int *const prog = ps.prog, *const fail = ps.fail;
        uint8_t *pl = planes ? nullptr : plane + (size_t)f0 * frame_pitch;
        const FFHipH264Edge *ed = planes ? nullptr : edges + (size_t)f0 * mb_w * mb_h * ne;
        FFHipDbPtrs PT;
        memset(&PT, 0, sizeof(PT));
        if (planes)
            for (int f = 0; f < nf; f++) {
                PT.plane[f] = planes[f0 + f];
                PT.edges[f] = edge_tabs[f0 + f];
            }
        if (skew) {
            /* waves per picture: one per band while the chip has SIMDs to spare (a lone picture is latency-bound), else what an
             * XCD's 128 SIMDs leave each of its pictures, but never fewer than a quarter of the bands (the wavefront's width) */
            const char *eb = FFHIP_KNOB("FFHIP_DEBLOCK_WAVES"), *ewp = FFHIP_KNOB("FFHIP_DEBLOCK_WPB");
            int wpb = ewp && atoi(ewp) >= 1 && atoi(ewp) <= 4 ? atoi(ewp) : 4; /* cooperating waves per workgroup (bands per super-band) */
            if (bd > 8 && !chroma && wpb > 3)
                wpb = 3; /* 16-bit luma: a strip row is 272 bytes, 3 waves' strip is what 64 KB of LDS hold */
            const int per_xcd = cdiv(nf, 8);
            int bwaves = eb && atoi(eb) > 0 ? atoi(eb) : 128 / per_xcd;
            if (bwaves < cdiv(nbands, 4)) bwaves = cdiv(nbands, 4);
            if (bwaves > nbands) bwaves = nbands;
            bwaves = cdiv(bwaves, wpb) * wpb; /* whole workgroups (waves beyond the last band idle) */
            const dim3 g(8 * (bwaves / wpb) * per_xcd), t(64 * wpb);
            const int mbs = chroma ? 8 : 16, qq = 64 / mbs, nee = chroma ? 4 : 8, psz = bd > 8 ? 2 : 1;
            const unsigned lds = (((unsigned)((4 + wpb * qq * mbs) * (8 * mbs * psz + 16) + (wpb * qq + 1) * 16 + 15)) & ~15u) +
                                 (unsigned)wpb * (unsigned)qq * (3 * nee + 4) * 4 + 2u * wpb * 4;
            static std::atomic<unsigned> launches{0};
            const int xrot = nf < 8 ? (int)(launches.fetch_add((unsigned)nf, std::memory_order_relaxed) & 7) : 0; /* where the batch's first picture goes */
#define DBS_LAUNCH(CH, T) hipLaunchKernelGGL((k_h264_deblock_skew<CH, T>), g, t, lds, stream, pl, frame_pitch, stride, mb_w, mb_h, ed, prog, nbands, bwaves, \
                                             nf, fail, fault, xrot, bd, PT, planes ? 1 : 0)
            if (bd > 8) { if (chroma) DBS_LAUNCH(true, uint16_t); else DBS_LAUNCH(false, uint16_t); }
            else        { if (chroma) DBS_LAUNCH(true, uint8_t); else DBS_LAUNCH(false, uint8_t); }
#undef DBS_LAUNCH
        } else if (!band)
            hipLaunchKernelGGL(k_h264_deblock_frame, dim3(mb_h, nf), dim3(64), 0, stream, pl, frame_pitch, stride, mb_w, mb_h, ed, prog, fail);
#define DB_LAUNCH(CH, W) hipLaunchKernelGGL((k_h264_deblock_band<CH, W>), dim3(nbands, nf), dim3(64 * W), 0, stream, pl, frame_pitch, stride, \
                                            mb_w, mb_h, ed, prog, nbands, fail, fault)
        else if (chroma) { if (bw == 16) DB_LAUNCH(true, 16); else if (bw == 8) DB_LAUNCH(true, 8); else DB_LAUNCH(true, 4); }
        else             { if (bw == 16) DB_LAUNCH(false, 16); else if (bw == 8) DB_LAUNCH(false, 8); else DB_LAUNCH(false, 4); }
#undef DB_LAUNCH
        const hipError_t e = hipGetLastError();
        const int r2 = ffhip_progress_release(&ps, stream, e == hipSuccess);
        if (e != hipSuccess) {
            ffhip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e), __FILE__, __LINE__);
            return FFHIP_EIO;
        }
        if (r2 < 0)
            return r2;
    }
    return 0;
}

int ffhip_launch_h264_deblock_frames(uint8_t *luma, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                     const FFHipH264Edge *edges, hipStream_t stream)
{
    return deblock_frames(false, luma, frame_pitch, nframes, stride, mb_w, mb_h, edges, stream);
}

int ffhip_launch_h264_deblock_frame(uint8_t *luma, ptrdiff_t stride, int mb_w, int mb_h, const FFHipH264Edge *edges,
                                    hipStream_t stream)
{
    return deblock_frames(false, luma, 0, 1, stride, mb_w, mb_h, edges, stream);
}

int ffhip_launch_h264_deblock_frames_chroma(uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                            const FFHipH264Edge *edges, hipStream_t stream)
{
    return deblock_frames(true, plane, frame_pitch, nframes, stride, mb_w, mb_h, edges, stream);
}

/* nframes pictures that do not sit at a constant pitch: planes[f] and its edge records edges[f] (host arrays of device pointers) */
int ffhip_launch_h264_deblock_pictures_bd(int bd, int chroma, uint8_t *const *planes, const FFHipH264Edge *const *edges, int nframes, ptrdiff_t stride,
                                          int mb_w, int mb_h, hipStream_t stream)
{
    return deblock_frames(chroma != 0, nullptr, 0, nframes, stride, mb_w, mb_h, nullptr, stream, bd, planes, edges);
}

/* 9 .. 14 bits: uint16_t samples (stride and frame pitch in bytes), the skewed-rows kernel only */
int ffhip_launch_h264_deblock_frames_bd(int bd, int chroma, uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                        const FFHipH264Edge *edges, hipStream_t stream)
{
    return deblock_frames(chroma != 0, plane, frame_pitch, nframes, stride, mb_w, mb_h, edges, stream, bd);
}
