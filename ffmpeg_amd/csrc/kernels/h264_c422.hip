/*
 * h264_c422.hip — the two frame-order kernels of the H.264 picture layer for the CHROMA planes of a 4:2:2 picture (round 4).
 *
 * At chroma_format_idc 2 a macroblock's chroma is 8 x 16 (libavcodec/h264_mb_template.c:41-262 with block_h = 16): the luma plane goes
 * through the luma kernels as it stands (the intra wavefront's luma-only form, k_h264_deblock_skew), the inter stages are the function
 * tables' (chroma MC of height 16, weights, idct_add8_422 expanded on the host) — what is left are the two dependency chains:
 *
 *   k_h264_intra_c422    intra prediction + residual of the chroma planes: pred8x16 (h264pred_template.c:567-817), chroma422_dc_dequant_idct,
 *                        the eight blocks of idct_add8_422 (h264idct_template.c:230-252,295-321) — the phase body is imb_c422_reconstruct()
 *                        of h264_intra_mb.h, shared with the CPU emulation.  Chroma prediction reads the left, upper-left and upper
 *                        neighbours: macroblock (x, y) starts when row y - 1 has finished macroblock x.
 *   k_h264_deblock_c422  the in-loop filter of one 8 x 16-macroblock plane in decoder order: per macroblock the vertical edges x = 0, 4
 *                        (h_loop_filter_chroma422: 16 lines, tc0 per 4 lines) then the horizontal edges y = 0, 4, 8, 12 (v_loop_filter_chroma);
 *                        macroblock (x, y) starts when row y - 1 has finished x + 1 (its right neighbour's first edge rewrites column 7).
 *
 * Both are the plain form of their 4:2:0 counterparts — one wave per macroblock row, a counter per row in the progress pool, everything that
 * crosses rows moved with device-scope loads and stores behind agent-scope fences — without those kernels' batching, LDS hand-offs and
 * prefetches: 4:2:2 is the contribution format, correctness first.  A launch takes the pictures (chroma planes) of a batch side by side.
 */
#include <stddef.h>

#include "common.h"
#include "h264_intra_mb.h"
#include "h264_kernels.h"

static_assert(sizeof(FFHipH264IntraC422) == 32, "FFHipH264IntraC422 is a 32-byte record");

namespace {
__device__ __forceinline__ void c4_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
struct C4Wave {
    int lane;
    template <class F>
    __device__ __forceinline__ void run(F body)
    {
        body(lane);
        c4_wave_sync();
    }
};
template <typename PIX> struct C4Quad { typedef uint32_t T; };
template <> struct C4Quad<uint16_t> { typedef uint64_t T; };
template <typename Q>
__device__ __forceinline__ Q c4_ld(const uint8_t *p)
{
    return __hip_atomic_load(reinterpret_cast<const Q *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename Q>
__device__ __forceinline__ void c4_st(uint8_t *p, Q v)
{
    __hip_atomic_store(reinterpret_cast<Q *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* waits until the row above has published `want`; false after a timeout (never in a correct run) */
__device__ __forceinline__ bool c4_wait(const int *counter, int want, int *fail, int lane)
{
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 24)) {
            if (lane == 0)
                __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}
/* every store of the wave is out and visible before the counter moves */
__device__ __forceinline__ void c4_publish(int *counter, int value, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0)
        __hip_atomic_store(counter, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
} // namespace

/* up to FFHIP_C422_PICS pictures of one geometry per launch (blockIdx.y = the picture: its planes, records and its slice of the counters) */
struct FFHipC422IntraSet { FFHipH264C422Pic pic[FFHIP_C422_PICS]; };
struct FFHipC422PlaneSet { uint8_t *plane[2 * FFHIP_C422_PICS]; const FFHipH264Edge *edges[2 * FFHIP_C422_PICS]; };

template <typename PIX>
__global__ __launch_bounds__(64) void k_h264_intra_c422(FFHipC422IntraSet S, ptrdiff_t sc, int mb_w, int mb_h, int *progress, int *fail, int maxv)
{
    uint8_t *const pcb = S.pic[blockIdx.y].cb, *const pcr = S.pic[blockIdx.y].cr;
    const FFHipH264IntraC422 *const recs = S.pic[blockIdx.y].recs;
    const int32_t *const row_start = S.pic[blockIdx.y].row_start;
    const int16_t *const coefs = S.pic[blockIdx.y].coefs;
    progress += (size_t)blockIdx.y * (size_t)mb_h;
    typedef typename C4Quad<PIX>::T Q;
    typedef typename ImbCoef<PIX>::T CF;
    constexpr int PS = (int)sizeof(PIX);
    __shared__ __align__(16) ImbTileC422<PIX> T;
    __shared__ __align__(16) FFHipH264IntraC422 Rs;
    const int my = (int)blockIdx.x, lane = (int)threadIdx.x;
    int k = __builtin_amdgcn_readfirstlane(row_start[my]);
    const int kend = __builtin_amdgcn_readfirstlane(row_start[my + 1]);
    if (lane < 16)
        T.zero[lane] = 0;
    /* nothing of this row is pending left of its first intra macroblock (the inter macroblocks were complete before the launch) */
    c4_publish(&progress[my], k < kend ? (int)recs[k].mb_x : mb_w, lane);
    C4Wave X{ lane };
    int prev_mx = -2;
    for (; k < kend; k++) {
        if (lane < 8)
            reinterpret_cast<uint32_t *>(&Rs)[lane] = reinterpret_cast<const uint32_t *>(recs + k)[lane];
        c4_wave_sync();
        const int mx = __builtin_amdgcn_readfirstlane((int)Rs.mb_x);
        const bool has_l = mx > 0, has_t = my > 0;
        if (has_t && !c4_wait(&progress[my - 1], min(mx + 1, mb_w), fail, lane))
            return;
        uint8_t *const base[2] = { pcb + (ptrdiff_t)my * 16 * sc + (ptrdiff_t)mx * 8 * PS, pcr + (ptrdiff_t)my * 16 * sc + (ptrdiff_t)mx * 8 * PS };
        /* neighbours, one quad per lane: lanes 0..5 the row above (columns -4 .. 7 of both planes), lanes 8..39 the column to the left —
         * out of this wave's own tile when the macroblock to the left was its previous one; what lies outside the picture reads as 0 */
        Q nb = 0;
        if (lane < 6) {
            const int p = lane / 3, c = 4 * (lane % 3) - 4;
            if (has_t && (c >= 0 || has_l))
                nb = c4_ld<Q>(base[p] - sc + c * PS);
        } else if (lane >= 8 && lane < 40) {
            const int p = (lane - 8) >> 4, r = (lane - 8) & 15;
            if (prev_mx == mx - 1)
                nb = *reinterpret_cast<const Q *>(&T.c[p][imb_ci(r, 4)]);
            else if (has_l)
                nb = c4_ld<Q>(base[p] + (ptrdiff_t)r * sc - 4 * PS);
        }
        c4_wave_sync();
        if (lane < 6)
            *reinterpret_cast<Q *>(&T.c[lane / 3][imb_ci(-1, 4 * (lane % 3) - 4)]) = nb;
        else if (lane >= 8 && lane < 40)
            *reinterpret_cast<Q *>(&T.c[(lane - 8) >> 4][imb_ci((lane - 8) & 15, -4)]) = nb;
        c4_wave_sync();
        imb_c422_reconstruct<PIX>(X, T, Rs, reinterpret_cast<const CF *>(coefs + Rs.coef), maxv);
        {
            const int p = lane >> 5, r = (lane >> 1) & 15, c = 4 * (lane & 1);
            c4_st<Q>(base[p] + (ptrdiff_t)r * sc + c * PS, *reinterpret_cast<const Q *>(&T.c[p][imb_ci(r, c)]));
        }
        c4_publish(&progress[my], k + 1 < kend ? (int)recs[k + 1].mb_x : mb_w, lane);
        prev_mx = mx;
    }
}

/* ---- the in-loop filter ---- */
#define C4_TP 16 /* tile pitch in samples: columns -4 .. 7 at [c + 4] */
template <typename PIX>
__device__ __forceinline__ void c4_edge(PIX *pix, int xs, bool intra, int alpha, int beta, int tc0, int bd)
{
    /* h264_loop_filter_chroma / _chroma_intra (h264dsp_template.c:228-322) on one line */
    const int p0 = pix[-xs], p1 = pix[-2 * xs], q0 = pix[0], q1 = pix[xs], maxv = (1 << bd) - 1;
    alpha <<= bd - 8;
    beta <<= bd - 8;
    if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta))
        return;
    if (intra) {
        pix[-xs] = (PIX)((2 * p1 + p0 + q1 + 2) >> 2);
        pix[0] = (PIX)((2 * q1 + q0 + p1 + 2) >> 2);
        return;
    }
    const int tc = (int)(((unsigned)(tc0 - 1)) << (bd - 8)) + 1;
    if (tc <= 0)
        return;
    const int delta = min(max(((q0 - p0) * 4 + (p1 - q1) + 4) >> 3, -tc), tc);
    pix[-xs] = (PIX)min(max(p0 + delta, 0), maxv);
    pix[0] = (PIX)min(max(q0 - delta, 0), maxv);
}

template <typename PIX>
__global__ __launch_bounds__(64) void k_h264_deblock_c422(FFHipC422PlaneSet S, ptrdiff_t stride, int mb_w, int mb_h, int *progress, int *fail, int bd)
{
    uint8_t *const plane = S.plane[blockIdx.y];
    const FFHipH264Edge *const edges = S.edges[blockIdx.y];
    progress += (size_t)blockIdx.y * (size_t)mb_h;
    typedef typename C4Quad<PIX>::T Q;
    constexpr int PS = (int)sizeof(PIX);
    /* tile[r + 2][c + 4]: rows -2 .. 15, columns -4 .. 7 of the macroblock */
    __shared__ __align__(16) PIX tile[18 * C4_TP];
    __shared__ __align__(16) FFHipH264Edge ed[6];
    const int my = (int)blockIdx.x, lane = (int)threadIdx.x;
    uint8_t *const rowbase = plane + (ptrdiff_t)my * 16 * stride;
    for (int mx = 0; mx < mb_w; mx++) {
        uint8_t *const mb = rowbase + (ptrdiff_t)mx * 8 * PS;
        if (lane < 18)
            reinterpret_cast<uint32_t *>(ed)[lane] = reinterpret_cast<const uint32_t *>(edges + (size_t)(my * mb_w + mx) * 6)[lane];
        /* the row above has finished macroblock mx + 1: its right neighbour's first vertical edge rewrites column 7 of the rows we read */
        if (my > 0 && !c4_wait(&progress[my - 1], min(mx + 2, mb_w), fail, lane))
            return;
        /* the macroblock's own 16 x 8 samples (lanes 0..31), the two rows above it over columns -4 .. 7 (lanes 32..37); the four columns to
         * the left stay in the tile from the previous macroblock (this wave filtered it: its latest values are nowhere else yet in order) */
        if (lane < 32) {
            const int r = lane >> 1, c = 4 * (lane & 1);
            *reinterpret_cast<Q *>(&tile[(r + 2) * C4_TP + c + 4]) = c4_ld<Q>(mb + (ptrdiff_t)r * stride + c * PS);
        } else if (lane < 38 && my > 0) {
            const int r = (lane - 32) / 3 - 2, c = 4 * ((lane - 32) % 3) - 4;
            if (c >= 0 || mx > 0)
                *reinterpret_cast<Q *>(&tile[(r + 2) * C4_TP + c + 4]) = c4_ld<Q>(mb + (ptrdiff_t)r * stride + c * PS);
        }
        c4_wave_sync();
        /* vertical edges x = 0, 4: lane = line */
        for (int k = 0; k < 2; k++) {
            const FFHipH264Edge e = ed[k];
            if (lane < 16 && e.alpha && e.beta && !(k == 0 && mx == 0))
                c4_edge<PIX>(&tile[(lane + 2) * C4_TP + 4 + 4 * k], 1, e.kind >= 4, e.alpha, e.beta, e.tc0[lane >> 2], bd);
            c4_wave_sync();
        }
        /* horizontal edges y = 0, 4, 8, 12: lane = column */
        for (int k = 0; k < 4; k++) {
            const FFHipH264Edge e = ed[2 + k];
            if (lane < 8 && e.alpha && e.beta && !(k == 0 && my == 0))
                c4_edge<PIX>(&tile[(4 * k + 2) * C4_TP + 4 + lane], C4_TP, e.kind >= 4, e.alpha, e.beta, e.tc0[lane >> 1], bd);
            c4_wave_sync();
        }
        /* what this macroblock may have changed goes back: its own samples, row -1 above it (p0 of the top edge), and the quad of
         * columns -4 .. -1 of rows 0 .. 15 (p0 of the left edge is column -1) */
        if (lane < 32) {
            const int r = lane >> 1, c = 4 * (lane & 1);
            c4_st<Q>(mb + (ptrdiff_t)r * stride + c * PS, *reinterpret_cast<const Q *>(&tile[(r + 2) * C4_TP + c + 4]));
        } else if (lane < 34) {
            if (my > 0) {
                const int c = 4 * (lane - 32);
                c4_st<Q>(mb - stride + c * PS, *reinterpret_cast<const Q *>(&tile[1 * C4_TP + c + 4]));
            }
        } else if (lane >= 40 && lane < 56 && mx > 0) {
            const int r = lane - 40;
            c4_st<Q>(mb + (ptrdiff_t)r * stride - 4 * PS, *reinterpret_cast<const Q *>(&tile[(r + 2) * C4_TP]));
        }
        c4_publish(&progress[my], mx + 1, lane);
        /* the next macroblock's left context: columns 4 .. 7 of rows -2 .. 15 become its columns -4 .. -1 */
        Q carry = 0;
        if (lane < 18)
            carry = *reinterpret_cast<const Q *>(&tile[lane * C4_TP + 8]);
        c4_wave_sync();
        if (lane < 18)
            *reinterpret_cast<Q *>(&tile[lane * C4_TP]) = carry;
        c4_wave_sync();
    }
}

static bool c422_bd_ok(int bd) { return bd == 8 || bd == 9 || bd == 10 || bd == 12 || bd == 14; }

int ffhip_launch_h264_intra_c422_pics(int bd, int npics, const FFHipH264C422Pic *pics, ptrdiff_t sc, int mb_w, int mb_h, hipStream_t stream)
{
    if (mb_w <= 0 || mb_h <= 0 || npics <= 0)
        return 0;
    const unsigned amask = bd > 8 ? 7u : 3u;
    if (!c422_bd_ok(bd) || !pics || ((size_t)sc & amask)) {
        ffhip_set_error("ffhip_h264_intra_c422: bad argument (depths 8 / 9 / 10 / 12 / 14; planes and stride %u-byte aligned)", amask + 1);
        return FFHIP_EINVAL;
    }
    for (int i = 0; i < npics; i++)
        if (!pics[i].cb || !pics[i].cr || !pics[i].recs || !pics[i].row_start || !pics[i].coefs || (((uintptr_t)pics[i].cb | (uintptr_t)pics[i].cr) & amask)) {
            ffhip_set_error("ffhip_h264_intra_c422: null or misaligned argument (picture %d)", i);
            return FFHIP_EINVAL;
        }
    if (mb_h > FFHIP_PROGRESS_SLOT_INTS) {
        ffhip_set_error("ffhip_h264_intra_c422: %d macroblock rows exceed the progress pool", mb_h);
        return FFHIP_EINVAL;
    }
    int per = FFHIP_PROGRESS_SLOT_INTS / mb_h;
    per = per > FFHIP_C422_PICS ? FFHIP_C422_PICS : per;
    for (int p0 = 0; p0 < npics; p0 += per) {
        const int n = npics - p0 < per ? npics - p0 : per;
        FFHipC422IntraSet S;
        for (int i = 0; i < FFHIP_C422_PICS; i++)
            S.pic[i] = pics[p0 + (i < n ? i : 0)];
        FFHipProgressSlot ps;
        const int r = ffhip_progress_acquire(mb_h * n, stream, &ps);
        if (r < 0)
            return r;
        if (bd > 8)
            hipLaunchKernelGGL(k_h264_intra_c422<uint16_t>, dim3(mb_h, n), dim3(64), 0, stream, S, sc, mb_w, mb_h, ps.prog, ps.fail, (1 << bd) - 1);
        else
            hipLaunchKernelGGL(k_h264_intra_c422<uint8_t>, dim3(mb_h, n), dim3(64), 0, stream, S, sc, mb_w, mb_h, ps.prog, ps.fail, 255);
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

int ffhip_launch_h264_intra_c422(int bd, uint8_t *cb, uint8_t *cr, ptrdiff_t sc, int mb_w, int mb_h, const FFHipH264IntraC422 *recs,
                                 const int32_t *row_start, const int16_t *coefs, hipStream_t stream)
{
    const FFHipH264C422Pic one = { cb, cr, recs, row_start, coefs };
    return ffhip_launch_h264_intra_c422_pics(bd, 1, &one, sc, mb_w, mb_h, stream);
}

/* nplanes chroma planes of one geometry and stride (Cb and Cr of one picture, or of the pictures of a batch), each with its edge records */
int ffhip_launch_h264_deblock_c422_planes(int bd, int nplanes, uint8_t *const *planes, const FFHipH264Edge *const *edges, ptrdiff_t stride, int mb_w,
                                          int mb_h, hipStream_t stream)
{
    if (mb_w <= 0 || mb_h <= 0 || nplanes <= 0)
        return 0;
    const unsigned amask = bd > 8 ? 7u : 3u;
    if (!c422_bd_ok(bd) || !planes || !edges || ((size_t)stride & amask)) {
        ffhip_set_error("ffhip_h264_deblock_c422: bad argument (depths 8 / 9 / 10 / 12 / 14; planes and stride %u-byte aligned)", amask + 1);
        return FFHIP_EINVAL;
    }
    for (int i = 0; i < nplanes; i++)
        if (!planes[i] || !edges[i] || ((uintptr_t)planes[i] & amask) || ((uintptr_t)edges[i] & 3)) {
            ffhip_set_error("ffhip_h264_deblock_c422: null or misaligned argument (plane %d)", i);
            return FFHIP_EINVAL;
        }
    if (mb_h > FFHIP_PROGRESS_SLOT_INTS) {
        ffhip_set_error("ffhip_h264_deblock_c422: %d macroblock rows exceed the progress pool", mb_h);
        return FFHIP_EINVAL;
    }
    int per = FFHIP_PROGRESS_SLOT_INTS / mb_h;
    per = per > 2 * FFHIP_C422_PICS ? 2 * FFHIP_C422_PICS : per;
    for (int p0 = 0; p0 < nplanes; p0 += per) {
        const int n = nplanes - p0 < per ? nplanes - p0 : per;
        FFHipC422PlaneSet S;
        for (int i = 0; i < 2 * FFHIP_C422_PICS; i++) {
            S.plane[i] = planes[p0 + (i < n ? i : 0)];
            S.edges[i] = edges[p0 + (i < n ? i : 0)];
        }
        FFHipProgressSlot ps;
        const int r = ffhip_progress_acquire(mb_h * n, stream, &ps);
        if (r < 0)
            return r;
        if (bd > 8)
            hipLaunchKernelGGL(k_h264_deblock_c422<uint16_t>, dim3(mb_h, n), dim3(64), 0, stream, S, stride, mb_w, mb_h, ps.prog, ps.fail, bd);
        else
            hipLaunchKernelGGL(k_h264_deblock_c422<uint8_t>, dim3(mb_h, n), dim3(64), 0, stream, S, stride, mb_w, mb_h, ps.prog, ps.fail, bd);
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

int ffhip_launch_h264_deblock_c422(int bd, uint8_t *plane, ptrdiff_t stride, int mb_w, int mb_h, const FFHipH264Edge *edges, hipStream_t stream)
{
    return ffhip_launch_h264_deblock_c422_planes(bd, 1, &plane, &edges, stride, mb_w, mb_h, stream);
}
