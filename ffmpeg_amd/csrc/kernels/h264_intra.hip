/*
 * h264_intra.hip — the intra reconstruction wavefront of the H.264 picture pipeline.
 *
 * Intra prediction reads the reconstructed (not yet deblocked) samples of the macroblocks to the left, above-left, above and
 * above-right, and inside a macroblock chains through the residual add from block to block (hl_decode_mb(),
 * libavcodec/h264_mb_template.c:151-262; hl_decode_mb_predict_luma, libavcodec/h264_mb.c:612-735).  A picture's intra macroblocks
 * are therefore one dependency graph, not a batch: per-level launches would cost ~10^4 launches per 4K I-picture.  Here ONE launch
 * walks it: one wave per macroblock row steps through the row's intra macroblocks left to right (the inter macroblocks of the
 * picture are complete when this launch starts — their prediction and residual stages ran before it); macroblock (x, y) starts
 * once row y - 1 has finished every macroblock up to x + 1.  A row publishes "all macroblocks left of X are done" where X is
 * its next intra macroblock, so a P-picture's few intra macroblocks cost a few hand-offs, and an I-picture is the full
 * mb_w + 2 mb_h chain.  All three planes of a macroblock are reconstructed in the same step on an LDS tile (h264_intra_mb.h:
 * the phases of one macroblock, shared with the CPU emulation that pins them against the oracle).
 *
 * Hand-off between rows: the protocol of k_h264_deblock_frame (h264_deblock.hip) — everything that crosses rows moves with
 * device-scope relaxed loads and stores (the 8 XCDs' L2s are not coherent with each other), stores are acknowledged
 * (s_waitcnt 0) before the row's counter moves, and the counter's value is awaited before the neighbour loads are issued.
 */
#include <stddef.h>
#include <stdlib.h>

#include "common.h"
#include "h264_intra_mb.h"
#include "h264_kernels.h"

static_assert(sizeof(FFHipH264IntraMB) == 108, "FFHipH264IntraMB is a 108-byte record");

namespace {
__device__ __forceinline__ void imb_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct ImbWave {
    int lane;
    template <class F>
    __device__ __forceinline__ void run(F body)
    {
        body(lane);
        imb_wave_sync();
    }
};

/* four samples: a dword at 8 bits, two above */
template <typename PIX> struct ImbQuad { typedef uint32_t T; };
template <> struct ImbQuad<uint16_t> { typedef uint64_t T; };
template <typename Q>
__device__ __forceinline__ Q ld_dev(const uint8_t *p)
{
    return __hip_atomic_load(reinterpret_cast<const Q *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename Q>
__device__ __forceinline__ void st_dev(uint8_t *p, Q v)
{
    __hip_atomic_store(reinterpret_cast<Q *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* the top neighbours of macroblock (mx, my) out of memory, as the lanes hold them: luma columns -4 .. 27 in lanes 0..7, Cb / Cr columns
 * -4 .. 7 in lanes 24..29; what lies outside the picture reads as 0 */
template <typename Q, int PS>
__device__ __forceinline__ Q imb_top_from_mem(const uint8_t *py, const uint8_t *pcb, const uint8_t *pcr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int my,
                                              int mx, int lane, int parts)
{
    const bool has_l = mx > 0, has_r = mx + 1 < mb_w;
    Q v = 0;
    if (lane < 8 ? !(parts & 1) : !(parts & 2)) {
    } else if (lane < 8) {
        const int c = 4 * lane - 4;
        if ((c >= 0 || has_l) && (c < 16 || has_r))
            v = ld_dev<Q>(py + ((ptrdiff_t)my * 16 - 1) * sy + (mx * 16 + c) * PS);
    } else if (lane >= 24 && lane < 30) {
        const int p = (lane - 24) / 3, c = 4 * ((lane - 24) % 3) - 4;
        if (c >= 0 || has_l)
            v = ld_dev<Q>((p ? pcr : pcb) + ((ptrdiff_t)my * 8 - 1) * sc + (mx * 8 + c) * PS);
    }
    return v;
}
} // namespace

/* dword positions inside FFHipH264IntraMB the prefetch reads out of the lanes that hold them */
static_assert(offsetof(FFHipH264IntraMB, type) == 4 && offsetof(FFHipH264IntraMB, flags) == 40 && offsetof(FFHipH264IntraMB, coef) == 68 &&
                  offsetof(FFHipH264IntraMB, blocks) == 72,
              "record layout");
#define IMB_REC_DW ((int)(sizeof(FFHipH264IntraMB) / 4))

/* PIX = uint8_t: strides / plane pointers in bytes, runs of int16 coefficients (at most 16 x 16 luma + 8 x 16 chroma = 384 int16 = 3 dwords
 * per lane).  PIX = uint16_t (9..14 bits): int32 coefficients, the sixteen luma DCs ahead of them: (16 + 384) int32 = 7 dwords per lane.
 *
 * Round 3: W = blockDim.x / 64 consecutive macroblock rows per workgroup (the recipe of k_h264_deblock_skew).  Inside the workgroup
 * a row hands its macroblocks' BOTTOM LINES to the row below through an LDS line buffer (`lines`: one luma line and two chroma lines of
 * the picture's width per row boundary) and an LDS counter — no store acknowledgement, no poll, no load round trip; the line is
 * pre-filled with what memory holds (the inter macroblocks of a P-picture are complete before this launch) unless the whole row is
 * intra.  A macroblock whose left neighbour was the previous record of the row takes its left column from its own tile.  Only the
 * rows at a workgroup boundary use memory: the producer publishes a macroblock one step LATE (at the top of the next step, when the
 * write-through stores issued a whole step earlier have long been acknowledged), the consumer reads the counter and the next record's
 * top neighbours one step AHEAD — both stay off the step's critical path, at the price of a few macroblocks of lag per boundary. */
/* Round 4: a launch carries up to FFHIP_INTRA_PICS pictures of one geometry (blockIdx.y: the picture; each its own planes, records
 * and progress counters) — the wavefront of ONE 1080p picture keeps 68 waves busy on a chip that holds 8,192, and its latency
 * (mb_w + 2 mb_h dependent steps) does not shrink; what a decoder with N pictures in hand (frame threads, an all-intra stream) wants is
 * N wavefronts side by side in one launch.  Workgroups are dispatched x-fastest, so a row's upper neighbour of the same picture is
 * always dispatched before it, whatever the other pictures do. */
/* (two workgroups per CU where it costs no spill: 8 bits fits 256 VGPRs; above, the kernel keeps its 286) */
template <typename PIX>
__global__ __launch_bounds__(256, 2) void k_h264_intra_frame(FFHipIntraPics S, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h, int *progress_all,
                                                          int *fail, int maxv, int luma_only)
{
    /* (read once: the set is indexed at run time, and fields used in place would be re-read from the kernel arguments in the loops) */
    uint8_t *const py = S.pic[blockIdx.y].y, *const pcb = S.pic[blockIdx.y].cb, *const pcr = S.pic[blockIdx.y].cr;
    const FFHipH264IntraMB *const recs = S.pic[blockIdx.y].recs;
    const int32_t *const row_start = S.pic[blockIdx.y].row_start;
    const int16_t *const coefs = S.pic[blockIdx.y].coefs;
    typedef typename ImbQuad<PIX>::T Q;
    typedef typename ImbCoef<PIX>::T CF;
    constexpr int PS = (int)sizeof(PIX), NDW = PS == 1 ? 3 : 7, IMB_RUN_MAX = NDW * 128 /* int16 */, WMAX = 4;
    __shared__ __align__(16) ImbTileT<PIX> Ts[WMAX];
    /* the macroblock being reconstructed and the next one: its record and coefficient run are fetched while this one is
     * worked on — read from global memory inside the block loop they were 16 dependent round trips per macroblock (measured:
     * 17.6 us per macroblock of an I-picture) */
    __shared__ __align__(16) FFHipH264IntraMB Rbs[WMAX][2];
    __shared__ __align__(16) int16_t Cbs[WMAX][2][IMB_RUN_MAX];
    __shared__ int ldone[WMAX];
    __shared__ uint32_t p4tab[IMB_TABS]; /* imb_tab(): the prediction rules as tables */
    extern __shared__ __align__(16) uint8_t imb_lines[];
    const int W = (int)(blockDim.x >> 6), wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    /* twice the workgroups a picture's rows need: the second set reconstructs the chroma planes, a wavefront of its own (intra
     * prediction never crosses planes) with its own counters — a step of the luma chain is a quarter shorter without them */
    const int nwg = (mb_h + W - 1) / W;
    /* luma_only (4:4:4, hl_decode_mb_444: h264_mb_template.c:256): every "picture" of the launch is ONE plane of a picture, reconstructed
     * by the luma rules from records that carry that plane's blocks; its cb / cr pointers are never used */
    const bool split = !luma_only && (int)gridDim.x > nwg, second = split && (int)blockIdx.x >= nwg;
    const int parts = luma_only ? 1 : split ? (second ? 2 : 1) : 3;
    const bool do_y = parts & 1, do_c = parts & 2;
    const int my = ((int)blockIdx.x - (second ? nwg : 0)) * W + wv;
    int *const progress = progress_all + ((size_t)blockIdx.y * (split ? 2 : 1) + (second ? 1 : 0)) * (size_t)(mb_h + 1);
    ImbTileT<PIX> &T = Ts[wv];
    FFHipH264IntraMB(&Rb)[2] = Rbs[wv];
    int16_t(&Cb)[2][IMB_RUN_MAX] = Cbs[wv];
    /* a row boundary inside the workgroup: 4 mb_w luma quads, then 2 mb_w quads of Cb and of Cr */
    const int lyq = mb_w * 4, lcq = mb_w * 2, line_q = lyq + 2 * lcq;
    Q *const mine = reinterpret_cast<Q *>(imb_lines) + (size_t)wv * line_q;
    const Q *const above = reinterpret_cast<const Q *>(imb_lines) + (size_t)(wv > 0 ? wv - 1 : 0) * line_q;
    const bool to_lds = wv + 1 < W && my + 1 < mb_h, from_lds = wv > 0, to_mem = !to_lds && my + 1 < mb_h, from_mem = wv == 0 && my > 0;
    if (lane == 0)
        __hip_atomic_store(&ldone[wv], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int i = (int)threadIdx.x; i < IMB_TABS; i += (int)blockDim.x)
        p4tab[i] = imb_tab(i);
    if (lane < 16)
        T.zero[lane] = 0;
    __syncthreads();
    if (my >= mb_h)
        return;
    int k = __builtin_amdgcn_readfirstlane(row_start[my]);
    const int kend = __builtin_amdgcn_readfirstlane(row_start[my + 1]);
    /* nothing of this row is pending left of its first intra macroblock */
    const int first = __builtin_amdgcn_readfirstlane(k < kend ? (int)recs[k].mb_x : mb_w);
    if (to_mem && lane == 0)
        __hip_atomic_store(&progress[my], first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (to_lds) {
        if (kend - k < mb_w) { /* some macroblocks are inter: their bottom lines are in memory already */
            const uint8_t *ry = py + ((ptrdiff_t)my * 16 + 15) * sy;
            for (int q = lane; q < lyq && do_y; q += 64)
                mine[q] = *reinterpret_cast<const Q *>(ry + (size_t)q * 4 * PS);
            for (int p = 0; p < 2 && do_c; p++) {
                const uint8_t *rc = (p ? pcr : pcb) + ((ptrdiff_t)my * 8 + 7) * sc;
                for (int q = lane; q < lcq; q += 64)
                    mine[lyq + p * lcq + q] = *reinterpret_cast<const Q *>(rc + (size_t)q * 4 * PS);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
            __hip_atomic_store(&ldone[wv], first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (k >= kend)
        return;
    /* dword `lane` of record idx; past the row's last record: of that one again, past the record's last dword: that one again (an
     * unconditional load: a default value written under a mask makes the compiler wait for every outstanding memory operation first) */
    auto fetch_rec = [&](int idx) __attribute__((always_inline)) {
        return reinterpret_cast<const uint32_t *>(recs + (idx < kend ? idx : kend - 1))[lane < IMB_REC_DW ? lane : IMB_REC_DW - 1];
    };
    uint32_t cw[NDW];
    auto fetch_run = [&](uint32_t rec_dw) __attribute__((always_inline)) { /* the run of the record whose dwords the lanes hold: NDW dwords per lane */
        const uint32_t type = __builtin_amdgcn_readlane(rec_dw, 1) & 0xFFu, blocks = __builtin_amdgcn_readlane(rec_dw, 18);
        const int flags = (int)(__builtin_amdgcn_readlane(rec_dw, 10) & 0xFFu);
        const int at = (int)__builtin_amdgcn_readlane(rec_dw, 17), ndw = imb_run_len((int)type, blocks, PS, flags) >> 1;
        const uint32_t *g = reinterpret_cast<const uint32_t *>(coefs + at);
#pragma unroll
        for (int j = 0; j < NDW; j++)
            cw[j] = lane + 64 * j < ndw ? g[lane + 64 * j] : 0u;
    };
    auto park = [&](int slot, uint32_t rec_dw) __attribute__((always_inline)) {
        if (lane < IMB_REC_DW)
            reinterpret_cast<uint32_t *>(&Rb[slot])[lane] = rec_dw;
#pragma unroll
        for (int j = 0; j < NDW; j++)
            reinterpret_cast<uint32_t *>(Cb[slot])[lane + 64 * j] = cw[j];
    };
    auto top_from_mem = [&](int mx) __attribute__((always_inline)) -> Q { return imb_top_from_mem<Q, PS>(py, pcb, pcr, sy, sc, mb_w, my, mx, lane, parts); };
    {
        const uint32_t r0 = fetch_rec(k);
        fetch_run(r0);
        park(0, r0);
    }
    imb_wave_sync();
    int known = 0;       /* last value seen of the counter of the row above */
    int ahead = 0;       /* from_mem: that counter, read a step ahead (lane 0) */
    int prev_mx = -2;    /* the macroblock this wave reconstructed last: its right columns are still in the tile */
    int mx = first;      /* this record's mb_x, out of the record prefetched a step earlier: a scalar (read from the parked record it is a
                          * vector value, and the waits below loops under execution masks) */
    Q pf = 0;            /* from_mem: the next record's top neighbours, read a step ahead */
    bool have_pf = false;
    /* Two records and one run ahead.  At the top of step k the record of macroblock k + 2 leaves; at the end of the step — after the
     * reconstruction, BEFORE the macroblock's stores — the run of k + 1 (in registers since the end of step k - 1) is parked in LDS and
     * the run of k + 2 leaves.  Nothing a step waits for is younger than a store: vector memory operations complete in order, and a
     * record looked at in the middle of a step used to wait for the previous macroblock's stores to be acknowledged. */
    uint32_t nrec = fetch_rec(k + 1);
    fetch_run(nrec);
    ImbWave X{ lane };
    for (int cur = 0; k < kend; k++, cur ^= 1) {
        const FFHipH264IntraMB &R = Rb[cur];
        if (to_mem) {
            /* the previous macroblock's write-through stores left a whole step ago: acknowledged, the counter moves */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            if (lane == 0)
                __hip_atomic_store(&progress[my], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t nrec2 = fetch_rec(k + 2);
        /* ---- the row above has finished macroblock mx + 1 ---- */
        const int want = min(mx + 2, mb_w);
        if (from_lds) {
            int spins = 0;
            while (known < want) {
                known = __hip_atomic_load(&ldone[wv - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
        } else if (from_mem && !have_pf) {
            known = max(known, __builtin_amdgcn_readfirstlane(ahead));
            int spins = 0;
            while (known < want) {
                known = __hip_atomic_load(&progress[my - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (known >= want)
                    break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 24)) {
                    if (lane == 0)
                        __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); /* the neighbour loads are issued after the counter was seen */
        }
        /* ---- neighbours into the tile, one quad per lane; what lies outside the picture reads as 0 ---- */
        uint8_t *ymb = py + (ptrdiff_t)my * 16 * sy + mx * 16 * PS;
        /* (Cr as an offset from Cb: a pointer picked per lane out of two loses its address space — flat instead of global accesses, and
         * a flat access also counts as an LDS operation the next LDS wait stalls for) */
        uint8_t *const cmb0 = pcb + (ptrdiff_t)my * 8 * sc + mx * 8 * PS;
        const ptrdiff_t cr_off = pcr - pcb;
        const bool has_l = mx > 0, has_r = mx + 1 < mb_w, left_here = prev_mx == mx - 1;
        Q nb = 0;
        if (from_lds) {
            if (lane < 8) { /* the row above: columns -4 .. 27 */
                const int c = 4 * lane - 4;
                if ((c >= 0 || has_l) && (c < 16 || has_r) && do_y)
                    nb = above[mx * 4 + lane - 1];
            } else if (lane >= 24 && lane < 30 && do_c) {
                const int p = (lane - 24) / 3, q = (lane - 24) % 3 - 1;
                if (q >= 0 || has_l)
                    nb = above[lyq + p * lcq + mx * 2 + q];
            }
        } else if (from_mem) {
            nb = have_pf ? pf : top_from_mem(mx);
        }
        if (lane >= 8 && lane < 24) { /* the column to the left */
            if (!do_y)
                ;
            else if (left_here)
                nb = *reinterpret_cast<const Q *>(&T.y[imb_yi(lane - 8, 12)]);
            else if (has_l)
                nb = ld_dev<Q>(ymb + (ptrdiff_t)(lane - 8) * sy - 4 * PS);
        } else if (lane >= 30 && lane < 46 && do_c) {
            if (left_here)
                nb = *reinterpret_cast<const Q *>(&T.c[(lane - 30) >> 3][imb_ci((lane - 30) & 7, 4)]);
            else if (has_l)
                nb = ld_dev<Q>(cmb0 + (((lane - 30) >> 3) ? cr_off : 0) + (ptrdiff_t)((lane - 30) & 7) * sc - 4 * PS);
        }
        if (lane < 8) {
            *reinterpret_cast<Q *>(&T.y[imb_yi(-1, 4 * lane - 4)]) = nb;
        } else if (lane < 24) { /* + zeros right of the macroblock (a top-right block that does not exist) */
            const int r = lane - 8;
            *reinterpret_cast<Q *>(&T.y[imb_yi(r, -4)]) = nb;
            *reinterpret_cast<Q *>(&T.y[imb_yi(r, 16)]) = 0u;
            *reinterpret_cast<Q *>(&T.y[imb_yi(r, 20)]) = 0u;
        } else if (lane < 30) {
            *reinterpret_cast<Q *>(&T.c[(lane - 24) / 3][imb_ci(-1, 4 * ((lane - 24) % 3) - 4)]) = nb;
        } else if (lane < 46) {
            *reinterpret_cast<Q *>(&T.c[(lane - 30) >> 3][imb_ci((lane - 30) & 7, -4)]) = nb;
        }
        imb_wave_sync();
        /* the next record's mb_x (read out of the lanes whether or not there is a next record: a load left pending on one path makes the
         * compiler wait for ALL memory operations — the previous macroblock's stores included — at the top of the loop) */
        const int next_x = (int)(__builtin_amdgcn_readlane(nrec, 0) & 0xFFFFu);
        const int next = k + 1 < kend ? next_x : mb_w;
        have_pf = false;
        if (from_mem && k + 1 < kend) { /* a step ahead: the counter, and the next record's top neighbours when they are known to be there */
            if (known >= min(next + 2, mb_w)) {
                pf = top_from_mem(next);
                have_pf = true;
            } else if (lane == 0) { /* consumed at the top of the next step, not here */
                ahead = __hip_atomic_load(&progress[my - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        imb_reconstruct<PIX>(X, T, R, reinterpret_cast<const CF *>(Cb[cur]), p4tab, maxv, parts);
        /* the next macroblock's record and run into LDS, the run after it leaves (after the row's last record: that record's again —
         * every step issues and consumes the same loads, so nothing is left pending on a path of its own) */
        park(cur ^ 1, nrec);
        fetch_run(nrec2);
        /* ---- the macroblock leaves the tile: 64 + 32 quads of samples; write-through where another workgroup reads them ---- */
        {
            uint8_t *dy = ymb + (ptrdiff_t)(lane >> 2) * sy + 4 * (lane & 3) * PS;
            const Q vy = *reinterpret_cast<const Q *>(&T.y[imb_yi(lane >> 2, 4 * (lane & 3))]);
            if (!do_y)
                ;
            else if (to_mem)
                st_dev<Q>(dy, vy);
            else
                *reinterpret_cast<Q *>(dy) = vy;
            if (to_lds && (lane >> 2) == 15 && do_y)
                mine[mx * 4 + (lane & 3)] = vy;
            if (lane < 32 && do_c) {
                const int p = lane >> 4, r = (lane >> 1) & 7, c = 4 * (lane & 1);
                uint8_t *dc = cmb0 + (p ? cr_off : 0) + (ptrdiff_t)r * sc + c * PS;
                const Q vc = *reinterpret_cast<const Q *>(&T.c[p][imb_ci(r, c)]);
                if (to_mem)
                    st_dev<Q>(dc, vc);
                else
                    *reinterpret_cast<Q *>(dc) = vc;
                if (to_lds && r == 7)
                    mine[lyq + p * lcq + mx * 2 + (lane & 1)] = vc;
            }
        }
        if (to_lds) { /* LDS operations of a wave execute in order: the line is in place when the counter moves */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0)
                __hip_atomic_store(&ldone[wv], next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        prev_mx = mx;
        mx = next;
        nrec = nrec2;
        imb_wave_sync(); /* the other record / run and the tile are rewritten by the next step */
    }
    if (to_mem) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0)
            __hip_atomic_store(&progress[my], mb_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int ffhip_launch_h264_intra_frame(uint8_t *y, uint8_t *cb, uint8_t *cr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                                  const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs, hipStream_t stream)
{
    return ffhip_launch_h264_intra_frame_bd(8, y, cb, cr, sy, sc, mb_w, mb_h, recs, row_start, coefs, stream);
}

int ffhip_launch_h264_intra_frame_bd(int bd, uint8_t *y, uint8_t *cb, uint8_t *cr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                                     const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs, hipStream_t stream)
{
    FFHipIntraPic one = { y, cb, cr, recs, row_start, coefs };
    return ffhip_launch_h264_intra_frames_bd(bd, 1, &one, sy, sc, mb_w, mb_h, stream);
}

int ffhip_launch_h264_intra_frames_bd(int bd, int npics, const FFHipIntraPic *pics, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                                      hipStream_t stream, int luma_only)
{
    if (mb_w <= 0 || mb_h <= 0 || npics <= 0)
        return 0;
    if (bd != 8 && bd != 9 && bd != 10 && bd != 12 && bd != 14) {
        ffhip_set_error("ffhip_h264_intra_frame: bit depth %d (8, 9, 10, 12 and 14 are the depths H.264 defines)", bd);
        return FFHIP_EINVAL;
    }
    const unsigned amask = bd > 8 ? 7u : 3u; /* four samples per access */
    if (!pics) {
        ffhip_set_error("ffhip_h264_intra_frame: null argument");
        return FFHIP_EINVAL;
    }
    for (int i = 0; i < npics; i++) {
        const FFHipIntraPic &P = pics[i];
        if (!P.y || !P.cb || !P.cr || !P.recs || !P.row_start || !P.coefs) {
            ffhip_set_error("ffhip_h264_intra_frame: null argument (picture %d)", i);
            return FFHIP_EINVAL;
        }
        if (((uintptr_t)P.y | (uintptr_t)P.cb | (uintptr_t)P.cr | (size_t)sy | (size_t)sc) & amask) {
            ffhip_set_error("ffhip_h264_intra_frame: planes and strides must be %u-byte aligned", amask + 1);
            return FFHIP_EINVAL;
        }
    }
    /* rows per workgroup: as many (up to 4) as the line buffers between them fit beside the static per-row tiles in 64 KB of LDS */
    const int ps_ = bd > 8 ? 2 : 1;
    const size_t fixed = bd > 8 ? sizeof(ImbTileT<uint16_t>) * 4 + sizeof(FFHipH264IntraMB) * 8 + 4 * 2 * 7 * 256 + 64 + IMB_TABS * 4
                                : sizeof(ImbTileT<uint8_t>) * 4 + sizeof(FFHipH264IntraMB) * 8 + 4 * 2 * 3 * 256 + 64 + IMB_TABS * 4;
    const size_t line = (size_t)mb_w * 32 * ps_;
    int W = 4;
#ifdef FFHIP_MEASURE
    if (const char *e = getenv("FFHIP_INTRA_WPB"))
        W = atoi(e) >= 1 && atoi(e) <= 4 ? atoi(e) : 4;
#endif
    while (W > 1 && fixed + (size_t)(W - 1) * line > 64 * 1024)
        W--;
    W = W < mb_h ? W : mb_h;
    const size_t lds = (size_t)(W - 1) * line;
    const int nwg = (mb_h + W - 1) / W;
    if (mb_h + 1 > FFHIP_PROGRESS_SLOT_INTS) {
        ffhip_set_error("ffhip_h264_intra_frame: %d macroblock rows exceed the progress pool", mb_h);
        return FFHIP_EINVAL;
    }
    /* the chroma planes as a wavefront of their own while the pictures of a launch leave the chip room for twice the workgroups
     * (two per CU by registers) */
    bool split = !luma_only && (size_t)npics * nwg * 2 <= 576;
#ifdef FFHIP_MEASURE
    if (const char *e = getenv("FFHIP_INTRA_SPLIT"))
        split = !luma_only && atoi(e) != 0;
#endif
    if ((mb_h + 1) * 2 > FFHIP_PROGRESS_SLOT_INTS)
        split = false;
    int per = FFHIP_PROGRESS_SLOT_INTS / ((mb_h + 1) * (split ? 2 : 1));
    per = per > FFHIP_INTRA_PICS ? FFHIP_INTRA_PICS : per;
    for (int p0 = 0; p0 < npics; p0 += per) {
        const int n = npics - p0 < per ? npics - p0 : per;
        FFHipIntraPics S;
        S.n = n;
        for (int i = 0; i < FFHIP_INTRA_PICS; i++)
            S.pic[i] = pics[p0 + (i < n ? i : 0)];
        FFHipProgressSlot ps;
        const int r = ffhip_progress_acquire((mb_h + 1) * (split ? 2 : 1) * n, stream, &ps);
        if (r < 0)
            return r;
        int *const prog = ps.prog, *const fail = ps.fail;
        if (bd > 8)
            hipLaunchKernelGGL(k_h264_intra_frame<uint16_t>, dim3(split ? 2 * nwg : nwg, n), dim3(64 * W), lds, stream, S, sy, sc, mb_w, mb_h, prog, fail, (1 << bd) - 1, luma_only);
        else
            hipLaunchKernelGGL(k_h264_intra_frame<uint8_t>, dim3(split ? 2 * nwg : nwg, n), dim3(64 * W), lds, stream, S, sy, sc, mb_w, mb_h, prog, fail, 255, luma_only);
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
