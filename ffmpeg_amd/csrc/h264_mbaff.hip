/*
 * h264_mbaff.hip — MBAFF frames in the H.264 picture layer (round 6): the two DEPENDENCY CHAINS of a frame whose
 * macroblock pairs mix frame and field macroblocks (mb_adaptive_frame_field_flag; libavcodec/h264_mb_template.c:61-78,
 * h264_loopfilter.c:494-560,716-760, h264_slice.c:2480-2505), 4:2:0, 8 - 14 bits.
 *
 * How an MBAFF frame is taken apart.  A field macroblock of a pair is the field-picture case per macroblock: its lines are every
 * second line of the frame from the pair's first (top field macroblock) or second (bottom) line on, twice the line size apart — which is
 * how hl_decode_mb() itself addresses it (mb_linesize = 2 * linesize, block_offset[48..], `dest_y -= linesize * 15` for the bottom
 * macroblock: h264_mb_template.c:61-78) and how the references' fields arrive (ff_h264_fill_mbaff_ref_list, h264_refs.c: entries
 * 16 + 2 i + parity with doubled line sizes).  So the INTER half of the picture needs no new kernel: the recorder
 * (integration/avcodec_h264_picture_hip.c) keeps three ordinary picture objects over the same planes — one for the frame macroblocks
 * (line size S), one per field parity (line size 2 S, half the rows, the bottom one starting a line further down) — and their
 * prediction, weight and residual lists run through the kernels every picture runs through.
 * What does not decompose are the two chains whose order crosses macroblocks:
 *
 *   intra reconstruction   k_h264_mbaff_intra: one wave per macroblock-PAIR row walks the row's intra macroblocks in decoding order (top,
 *                          then bottom macroblock of a pair); a macroblock is reconstructed on the tile of h264_intra_mb.h — the SAME phase
 *                          bodies as every other picture's (imb_reconstruct) — filled and written back at the macroblock's own line
 *                          step: "the row above" of a field macroblock is the line two frame lines up, of a bottom frame macroblock the
 *                          top macroblock's last line — memory adjacency at the macroblock's step is exactly the neighbour derivation of
 *                          the standard's 6.4.12.2, and what hl_decode_mb()'s predictors read.  Pair (x, p) starts when row p - 1 has
 *                          finished pair x + 1.
 *   the in-loop filter     k_h264_mbaff_deblock: ff_h264_filter_mb()'s dsp calls of a macroblock — up to ten per luma macroblock in an
 *                          MBAFF frame: the left edge in two halves with their own bS / qp (filter_mb_mbaff_edgev), the top edge of a frame
 *                          macroblock under a field pair once per field at twice the line size, the _mbaff members that cover 8 lines —
 *                          are RECORDED AS CALLS (pointer, line size, alpha, beta, tc0 in the order issued) and executed in that order, one
 *                          wave per pair row and plane, on an LDS tile of the pair and what its calls reach around it: pair x of row p
 *                          after pair x + 1 of row p - 1.  Which edge is filtered how is the reference's decision, call by call.
 *
 * Both are the plain form (h264_c422.hip's protocol: a counter per row in the progress pool, device-scope loads and stores behind
 * agent fences); interlaced material is correctness first.  The CPU tier executes the same lists in oracle/emul_h264_mbaff.cpp.
 */
#include <stddef.h>
#include <string.h>

#include <vector>

#include "kernels/common.h"
#include "kernels/h264_intra_mb.h"
#include "kernels/h264_kernels.h"
#include "kernels/h264_lf_line.h"
#include "kernels/progress_pool.h"

struct FFHipH264Mbaff {
    int mb_w, mb_h;                              /* the frame's macroblocks; mb_h even */
    int bd = 8;                                  /* 8, or 9 / 10 / 12 / 14: uint16_t samples, int32_t coefficients */
    int device;
    std::vector<FFHipH264IntraMB> recs;          /* intra macroblocks in decoding order: pair rows, pairs left to right, top then bottom */
    std::vector<uint32_t> geo;                   /* per record: mb_x | mb_y (frame row) << 12 | field << 24 */
    std::vector<int16_t> coefs;                  /* their packed coefficient runs */
    std::vector<int32_t> intra_row;              /* mb_h / 2 + 1 starts into recs */
    std::vector<FFHipH264Edge> calls[3];         /* per plane: the loop-filter calls in the order issued (pad = flags: 1 doubled line size, 2 _mbaff member) */
    std::vector<int32_t> pair_end[3];            /* per plane and pair (row-major): one past its last call */
    int last_pair[3];                            /* the pair whose calls are being appended (calls must arrive pair by pair, in order) */
    void *dev = nullptr;                         /* the lists as the last flush uploaded them */
    size_t dev_sz = 0;
    hipEvent_t done = nullptr;                   /* behind the last flush's launches: `dev` is theirs until it has passed */
    bool pending = false;
    int last_status = 0;
};

/* the launches of the last flush have read their lists: `dev` may be written (or freed) again */
static void mbaff_settle(FFHipH264Mbaff *m)
{
    if (m->pending && m->done)
        (void)hipEventSynchronize(m->done);
    m->pending = false;
}

extern "C" int ffhip_h264_mbaff_create(FFHipH264Mbaff **m, int mb_w, int mb_h)
{
    return ffhip_h264_mbaff_create_fmt(m, mb_w, mb_h, 8);
}

extern "C" int ffhip_h264_mbaff_create_fmt(FFHipH264Mbaff **m, int mb_w, int mb_h, int bit_depth)
{
    if (!m || mb_w <= 0 || mb_h <= 0 || (mb_h & 1) || mb_w > 4095 || mb_h > 4095)
        return FFHIP_EINVAL;
    if (bit_depth != 8 && bit_depth != 9 && bit_depth != 10 && bit_depth != 12 && bit_depth != 14) {
        ffhip_set_error("ffhip_h264_mbaff_create_fmt: bit_depth %d (8, 9, 10, 12, 14)", bit_depth);
        return FFHIP_EINVAL;
    }
    if (mb_h / 2 > FFHIP_PROGRESS_SLOT_INTS / 3) {
        ffhip_set_error("ffhip_h264_mbaff: %d macroblock pair rows exceed the progress pool", mb_h / 2);
        return FFHIP_EINVAL;
    }
    FFHipH264Mbaff *p = new (std::nothrow) FFHipH264Mbaff();
    if (!p)
        return FFHIP_ENOMEM;
    p->mb_w = mb_w;
    p->mb_h = mb_h;
    p->bd = bit_depth;
    if (hipGetDevice(&p->device) != hipSuccess) {
        (void)hipGetLastError();
        p->device = -1; /* recording needs no device; flush does */
    }
    ffhip_h264_mbaff_begin(p); /* (the per-pair tables exist from here on: lists() / flush() of an object nothing was recorded into are empty, not wild) */
    *m = p;
    return 0;
}

extern "C" void ffhip_h264_mbaff_free(FFHipH264Mbaff **m)
{
    if (!m || !*m)
        return;
    {
        FFHipDeviceGuard dg((*m)->device);
        mbaff_settle(*m);
        if ((*m)->done)
            (void)hipEventDestroy((*m)->done);
        if ((*m)->dev)
            (void)hipFree((*m)->dev);
    }
    delete *m;
    *m = nullptr;
}

extern "C" void ffhip_h264_mbaff_begin(FFHipH264Mbaff *m)
{
    if (!m)
        return;
    m->recs.clear();
    m->geo.clear();
    m->coefs.clear();
    for (int pl = 0; pl < 3; pl++) {
        m->calls[pl].clear();
        m->pair_end[pl].assign((size_t)m->mb_w * (m->mb_h / 2), 0);
        m->last_pair[pl] = -1;
    }
    m->last_status = 0;
}

extern "C" int ffhip_h264_mbaff_intra_mb(FFHipH264Mbaff *m, const FFHipH264IntraMB *desc, int field, const uint8_t *non_zero_count_cache, int16_t *mb,
                                         const int16_t *mb_luma_dc, const uint8_t *pcm)
{
    if (!m || !desc || !non_zero_count_cache || !mb)
        return FFHIP_EINVAL;
    if (desc->mb_x < 0 || desc->mb_x >= m->mb_w || desc->mb_y < 0 || desc->mb_y >= m->mb_h) {
        ffhip_set_error("ffhip_h264_mbaff_intra_mb: macroblock (%d, %d) outside %d x %d", desc->mb_x, desc->mb_y, m->mb_w, m->mb_h);
        return FFHIP_EINVAL;
    }
    /* decoding order: (pair row, mb_x, bottom) ascending */
    if (!m->geo.empty()) {
        const uint32_t g = m->geo.back();
        const int px = (int)(g & 0xFFF), py = (int)((g >> 12) & 0xFFF);
        const long prev = ((long)(py >> 1) * m->mb_w + px) * 2 + (py & 1), cur = ((long)(desc->mb_y >> 1) * m->mb_w + desc->mb_x) * 2 + (desc->mb_y & 1);
        if (cur <= prev) {
            ffhip_set_error("ffhip_h264_mbaff_intra_mb: macroblock (%d, %d) out of decoding order", desc->mb_x, desc->mb_y);
            return FFHIP_EINVAL;
        }
    }
    FFHipH264IntraMB r = *desc;
    const size_t at = m->coefs.size();
    m->coefs.resize(at + 832); /* (a run: at most 391 int16 at 8 bits, 2 x (384 + 16) above, from a 16-byte boundary) */
    int32_t n = (int32_t)at;
    const int rc = ffhip_h264_intra_pack_hbd(m->bd, &r, non_zero_count_cache, mb, mb_luma_dc, pcm, m->coefs.data(), &n, (int32_t)m->coefs.size());
    if (rc < 0) {
        m->coefs.resize(at);
        return rc;
    }
    m->coefs.resize((size_t)n);
    m->recs.push_back(r);
    m->geo.push_back((uint32_t)desc->mb_x | (uint32_t)desc->mb_y << 12 | (uint32_t)(field ? 1 : 0) << 24);
    return 0;
}

extern "C" int ffhip_h264_mbaff_filter_call(FFHipH264Mbaff *m, int plane, int mb_x, int mb_y, const FFHipH264Edge *call)
{
    if (!m || !call || plane < 0 || plane > 2 || mb_x < 0 || mb_x >= m->mb_w || mb_y < 0 || mb_y >= m->mb_h)
        return FFHIP_EINVAL;
    const int pair = (mb_y >> 1) * m->mb_w + mb_x;
    if (pair < m->last_pair[plane]) {
        ffhip_set_error("ffhip_h264_mbaff_filter_call: pair (%d, %d) after pair %d: calls arrive in decoding order", mb_x, mb_y >> 1, m->last_pair[plane]);
        return FFHIP_EINVAL;
    }
    if ((call->kind & ~7) || (call->pad & ~3) || (call->offset & 3)) {
        ffhip_set_error("ffhip_h264_mbaff_filter_call: bad record (kind %d, flags %d, offset %d)", call->kind, call->pad, call->offset);
        return FFHIP_EINVAL;
    }
    /* pairs without calls between the last one and this one end where the last one ended */
    const int32_t here = (int32_t)m->calls[plane].size();
    for (int q = m->last_pair[plane] + 1; q < pair; q++)
        m->pair_end[plane][(size_t)q] = here;
    m->calls[plane].push_back(*call);
    m->pair_end[plane][(size_t)pair] = here + 1;
    m->last_pair[plane] = pair;
    return 0;
}

/* pair_end[] of the pairs behind the last one that had calls, intra_row[] */
static void mbaff_finish(FFHipH264Mbaff *m)
{
    const int npairs = m->mb_w * (m->mb_h / 2);
    for (int pl = 0; pl < 3; pl++) {
        const int32_t n = (int32_t)m->calls[pl].size();
        for (int q = m->last_pair[pl] + 1; q < npairs; q++)
            m->pair_end[pl][(size_t)q] = n;
        m->last_pair[pl] = npairs - 1;
    }
    m->intra_row.assign((size_t)m->mb_h / 2 + 1, 0);
    size_t k = 0;
    for (int p = 0; p <= m->mb_h / 2; p++) {
        while (k < m->geo.size() && (int)((m->geo[k] >> 12) & 0xFFF) >> 1 < p)
            k++;
        m->intra_row[(size_t)p] = (int32_t)k;
    }
}

extern "C" int ffhip_h264_mbaff_lists(FFHipH264Mbaff *m, FFHipH264MbaffLists *out)
{
    if (!m || !out)
        return FFHIP_EINVAL;
    mbaff_finish(m);
    memset(out, 0, sizeof(*out));
    out->mb_w = m->mb_w;
    out->mb_h = m->mb_h;
    out->bit_depth = m->bd;
    out->recs = m->recs.data();
    out->geo = m->geo.data();
    out->coefs = m->coefs.data();
    out->intra_row = m->intra_row.data();
    out->nrecs = (int32_t)m->recs.size();
    out->ncoefs = (int32_t)m->coefs.size();
    for (int pl = 0; pl < 3; pl++) {
        out->calls[pl] = m->calls[pl].data();
        out->pair_end[pl] = m->pair_end[pl].data();
        out->ncalls[pl] = (int32_t)m->calls[pl].size();
    }
    return 0;
}

/* ================================================================================================== */
/* kernels */

namespace {
__device__ __forceinline__ void mb_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
struct MbWave {
    int lane;
    template <class F>
    __device__ __forceinline__ void run(F body)
    {
        body(lane);
        mb_wave_sync();
    }
};
__device__ __forceinline__ uint32_t mb_ld(const uint8_t *p)
{
    return __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mb_st(uint8_t *p, uint32_t v)
{
    __hip_atomic_store(reinterpret_cast<uint32_t *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* four samples: a dword at 8 bits, two above */
template <typename PIX> struct MbQuad { typedef uint32_t T; };
template <> struct MbQuad<uint16_t> { typedef uint64_t T; };
template <typename Q>
__device__ __forceinline__ Q mb_ldq(const uint8_t *p)
{
    return __hip_atomic_load(reinterpret_cast<const Q *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename Q>
__device__ __forceinline__ void mb_stq(uint8_t *p, Q v)
{
    __hip_atomic_store(reinterpret_cast<Q *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool mb_wait(const int *counter, int want, int *fail, int lane)
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
/* every store of the wave is out and visible (to its own later loads and to the other rows) */
__device__ __forceinline__ void mb_drain()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void mb_publish(int *counter, int value, int lane)
{
    mb_drain();
    if (lane == 0)
        __hip_atomic_store(counter, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
} // namespace

/* one wave per pair row; progress[p] = "every pair left of this one is reconstructed".  PIX = uint8_t, or uint16_t above 8 bits (strides and
 * plane pointers in bytes, int32 coefficients: h264_intra_mb.h ImbCoef) */
template <typename PIX>
__global__ __launch_bounds__(64) void k_h264_mbaff_intra(uint8_t *py, uint8_t *pcb, uint8_t *pcr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                                                         const FFHipH264IntraMB *recs, const uint32_t *geo, const int32_t *row_start,
                                                         const int16_t *coefs, int *progress, int *fail, int maxv)
{
    typedef typename MbQuad<PIX>::T Q;
    typedef typename ImbCoef<PIX>::T CF;
    constexpr int PS = (int)sizeof(PIX);
    __shared__ __align__(16) ImbTileT<PIX> T;
    __shared__ __align__(16) FFHipH264IntraMB R;
    __shared__ uint32_t p4tab[IMB_TABS];
    const int p = (int)blockIdx.x, lane = (int)threadIdx.x;
    for (int i = lane; i < IMB_TABS; i += 64)
        p4tab[i] = imb_tab(i);
    if (lane < 16)
        T.zero[lane] = 0;
    mb_wave_sync();
    int k = __builtin_amdgcn_readfirstlane(row_start[p]);
    const int kend = __builtin_amdgcn_readfirstlane(row_start[p + 1]);
    mb_publish(&progress[p], k < kend ? (int)(geo[k] & 0xFFF) : mb_w, lane);
    MbWave X{ lane };
    for (; k < kend; k++) {
        if (lane < (int)(sizeof(FFHipH264IntraMB) / 4))
            reinterpret_cast<uint32_t *>(&R)[lane] = reinterpret_cast<const uint32_t *>(recs + k)[lane];
        const uint32_t g = geo[k];
        const int mx = (int)(g & 0xFFF), my = (int)((g >> 12) & 0xFFF), field = (int)(g >> 24) & 1;
        mb_wave_sync();
        if (p > 0 && !mb_wait(&progress[p - 1], min(mx + 2, mb_w), fail, lane))
            return;
        /* the macroblock's first frame line and its line step: a field macroblock of pair p starts on the pair's line 0 / 1 */
        const int step = field ? 2 : 1;
        const int line0 = field ? 32 * p + (my & 1) : 16 * my;
        const ptrdiff_t ysy = sy * step, csc = sc * step;
        uint8_t *ymb = py + (ptrdiff_t)line0 * sy + mx * 16 * PS;
        const int cline0 = field ? 16 * p + (my & 1) : 8 * my;
        uint8_t *cmb[2] = { pcb + (ptrdiff_t)cline0 * sc + mx * 8 * PS, pcr + (ptrdiff_t)cline0 * sc + mx * 8 * PS };
        const bool has_l = mx > 0, has_t = line0 - step >= 0, has_r = mx + 1 < mb_w;
        /* the tile's neighbours, a quad per lane (what lies outside the picture reads as 0): lanes 0..7 the luma row above over columns
         * -4 .. 27, 8..23 the luma column to the left, 24..29 the chroma rows above (columns -4 .. 7), 30..45 the chroma columns to the left */
        {
            Q v = 0;
            if (lane < 8) {
                const int c = 4 * lane - 4;
                if (has_t && (c >= 0 || has_l) && (c < 16 || has_r))
                    v = mb_ldq<Q>(ymb - ysy + c * PS);
                *reinterpret_cast<Q *>(&T.y[imb_yi(-1, c)]) = v;
            } else if (lane < 24) {
                const int r = lane - 8;
                if (has_l)
                    v = mb_ldq<Q>(ymb + (ptrdiff_t)r * ysy - 4 * PS);
                *reinterpret_cast<Q *>(&T.y[imb_yi(r, -4)]) = v;
                *reinterpret_cast<Q *>(&T.y[imb_yi(r, 16)]) = 0;
                *reinterpret_cast<Q *>(&T.y[imb_yi(r, 20)]) = 0;
            } else if (lane < 30) {
                const int pl = (lane - 24) / 3, c = 4 * ((lane - 24) % 3) - 4;
                if (has_t && (c >= 0 || has_l))
                    v = mb_ldq<Q>(cmb[pl] - csc + c * PS);
                *reinterpret_cast<Q *>(&T.c[pl][imb_ci(-1, c)]) = v;
            } else if (lane < 46) {
                const int pl = (lane - 30) >> 3, r = (lane - 30) & 7;
                if (has_l)
                    v = mb_ldq<Q>(cmb[pl] + (ptrdiff_t)r * csc - 4 * PS);
                *reinterpret_cast<Q *>(&T.c[pl][imb_ci(r, -4)]) = v;
            }
        }
        mb_wave_sync();
        imb_reconstruct<PIX>(X, T, R, reinterpret_cast<const CF *>(coefs + R.coef), p4tab, maxv, 3);
        /* the macroblock back into the picture: 64 luma quads, 32 chroma quads */
        mb_stq<Q>(ymb + (ptrdiff_t)(lane >> 2) * ysy + 4 * (lane & 3) * PS, *reinterpret_cast<const Q *>(&T.y[imb_yi(lane >> 2, 4 * (lane & 3))]));
        if (lane < 32) {
            const int pl = lane >> 4, r = (lane >> 1) & 7, c = 4 * (lane & 1);
            mb_stq<Q>(cmb[pl] + (ptrdiff_t)r * csc + c * PS, *reinterpret_cast<const Q *>(&T.c[pl][imb_ci(r, c)]));
        }
        /* the next record: the other macroblock of this pair (the pair is not finished), or a pair further right */
        const int nx = k + 1 < kend ? (int)(geo[k + 1] & 0xFFF) : mb_w;
        if (nx != mx)
            mb_publish(&progress[p], nx, lane);
        else
            mb_drain();
    }
}

/* The recorded loop-filter calls of all three planes: blockIdx.x = the pair row, blockIdx.y = the plane; one wave each.
 *
 * A pair's calls touch its own 32 lines (chroma 16), up to 8 lines (chroma 4) of the pair above — a frame macroblock's top edge under a
 * field pair is filtered once per field at twice the line size, three samples deep each (h264_loopfilter.c:520-545) — and 4 columns of
 * the pair to the left: 40 x 20 bytes.  The wave fetches that tile ONCE (dwords at device scope), runs the pair's calls on it in LDS in
 * the order recorded — a call costs an LDS round trip, not a memory one — and writes back the dwords that changed.  flush() places every call
 * in its pair's tile on the host (one division per call there, none here).  1920 x 1088, ~110 000 calls per frame: 9.3 ms with a memory
 * round trip per call and a launch per plane, 3.7 ms on the tile, 1.9 ms with a lane per column in the v_ members and the calls placed by
 * the host (profiles/r06_h264_mbaff_v0 / _v1 / r06_h264_mbaff_kernel_stats.csv).  No other wave
 * touches the tile meanwhile: row p - 1 is at pair x + 2 or beyond (columns >= 16 x + 28), row p + 1 at pair x - 2 or before. */
/* a call as the kernel takes it (12 bytes, made by flush() from the recorded FFHipH264Edge and the plane's line size): where in the pair's
 * tile, which member, its thresholds */
struct MbaffDevCall {
    uint16_t toff;       /* byte offset of pix in the tile */
    uint8_t kf;          /* bits 0..2 FFHIP_H264_LF_* kind, bit 3 twice the line size, bit 4 the _mbaff member */
    uint8_t alpha, beta;
    uint8_t pad[3];
    int8_t tc0[4];
};
static_assert(sizeof(MbaffDevCall) == 12, "MbaffDevCall is three dwords");

struct MbaffLfArgs {
    uint8_t *plane[3];
    ptrdiff_t stride[3];
    const MbaffDevCall *calls[3];
    const int32_t *pair_end[3];
    int mb_w, prow;
};

/* the tile of a pair on a plane: W x H samples of the pair, AB lines above, 4 columns to the left */
#define MBAFF_TILE(cpl, W, H, AB, TP) const int W = (cpl) ? 8 : 16, H = (cpl) ? 16 : 32, AB = (cpl) ? 4 : 8, TP = W + 4

__global__ __launch_bounds__(64) void k_h264_mbaff_deblock(MbaffLfArgs A, int *progress_all, int *fail)
{
    __shared__ __align__(16) uint8_t tile[40 * 20];
    __shared__ __align__(16) uint32_t C[64 * 3];
    const int p = (int)blockIdx.x, pl = (int)blockIdx.y, lane = (int)threadIdx.x;
    MBAFF_TILE(pl != 0, W, H, AB, TP);
    const int TD = TP / 4, TL = H + AB, ND = TL * TD; /* dwords per tile line, lines, dwords */
    uint8_t *const plane = A.plane[pl];
    const ptrdiff_t stride = A.stride[pl];
    const MbaffDevCall *const calls = A.calls[pl];
    const int32_t *const pair_end = A.pair_end[pl];
    const int mb_w = A.mb_w;
    int *const progress = progress_all + pl * A.prow;
    if (!calls) /* nothing recorded on this plane */
        return;
    int at = p > 0 ? pair_end[(size_t)p * mb_w - 1] : 0;
    for (int x = 0; x < mb_w; x++) {
        const int end = pair_end[(size_t)p * mb_w + x];
        if (at >= end) { /* a pair without calls: nothing of it is pending */
            if (lane == 0)
                __hip_atomic_store(&progress[p], x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        /* the pair's first calls are on their way while the row above is awaited */
        int n = min(64, end - at);
        uint32_t c0 = 0, c1 = 0, c2 = 0;
        if (lane < n) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(calls + at + lane);
            c0 = src[0]; c1 = src[1]; c2 = src[2];
        }
        if (p > 0 && !mb_wait(&progress[p - 1], min(x + 2, mb_w), fail, lane))
            return;
        const int line0 = H * p - AB, col0 = W * x - 4;
        uint32_t orig[4];
        bool have[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int d = lane + 64 * k, tl = d / TD, tc = d - tl * TD;
            have[k] = d < ND && line0 + tl >= 0 && col0 + 4 * tc >= 0;
            orig[k] = 0;
            if (have[k])
                orig[k] = mb_ld(plane + (ptrdiff_t)(line0 + tl) * stride + col0 + 4 * tc);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (lane + 64 * k < ND)
                reinterpret_cast<uint32_t *>(tile)[lane + 64 * k] = orig[k];
        for (;;) {
            if (lane < n) {
                C[3 * lane] = c0; C[3 * lane + 1] = c1; C[3 * lane + 2] = c2;
            }
            mb_wave_sync();
            for (int i = 0; i < n; i++) {
                const uint32_t w0 = C[3 * i], w1 = C[3 * i + 1];
                const int toff = (int)(w0 & 0xFFFF), kf = (int)(w0 >> 16) & 255, alpha = (int)(w0 >> 24), beta = (int)(w1 & 255);
                const bool chroma = kf & 2, intra = kf & 4, hfilt = kf & 1; /* h_ members filter across a VERTICAL edge: a line = a row */
                const int half = (kf >> 4) & 1;                             /* the _mbaff members: half the lines, half as many per tc0 entry */
                const int step = (kf & 8) ? 2 * TP : TP;                    /* tile bytes from a line of the call to the next */
                const int cls = (chroma ? 1 : 0) + (intra ? 2 : 0);
                const int8_t *tc0 = reinterpret_cast<const int8_t *>(&C[3 * i + 2]);
                if (hfilt) {
                    /* lines = rows: luma 16 (tc0 per 4), chroma 8 (per 2); _mbaff: 8 (per 2), 4 (per 1) (h264dsp_template.c:127-133,262-272) */
                    const int nlines = (chroma ? 8 : 16) >> half, per = (chroma ? 2 : 4) >> half;
                    if (lane < nlines) {
                        uint8_t *l = tile + toff + lane * step - 4;
                        const uint32_t a = *reinterpret_cast<const uint32_t *>(l), b = *reinterpret_cast<const uint32_t *>(l + 4);
                        LfLine v = { (int)(a & 255), (int)((a >> 8) & 255), (int)((a >> 16) & 255), (int)(a >> 24),
                                     (int)(b & 255), (int)((b >> 8) & 255), (int)((b >> 16) & 255), (int)(b >> 24) };
                        const int m = lf_line(v, cls, alpha, beta, intra ? 0 : tc0[lane / per]);
                        if (m & 7)
                            *reinterpret_cast<uint32_t *>(l) = (uint32_t)v.p3 | (uint32_t)v.p2 << 8 | (uint32_t)v.p1 << 16 | (uint32_t)v.p0 << 24;
                        if (m & 56)
                            *reinterpret_cast<uint32_t *>(l + 4) = (uint32_t)v.q0 | (uint32_t)v.q1 << 8 | (uint32_t)v.q2 << 16 | (uint32_t)v.q3 << 24;
                    }
                } else {
                    /* lines = columns, a lane each: luma 16 (tc0 per 4; rows -4 .. 3: the strong filter's p3 / q3), chroma 8 (per 2; rows -2 .. 1) */
                    const int ncols = chroma ? 8 : 16, per = chroma ? 2 : 4;
                    if (lane < ncols) {
                        uint8_t *c = tile + toff + lane;
                        LfLine v;
                        v.p1 = c[-2 * step]; v.p0 = c[-step]; v.q0 = c[0]; v.q1 = c[step];
                        v.p3 = v.p2 = v.q2 = v.q3 = 0;
                        if (!chroma) {
                            v.p3 = c[-4 * step]; v.p2 = c[-3 * step]; v.q2 = c[2 * step]; v.q3 = c[3 * step];
                        }
                        const int m = lf_line(v, cls, alpha, beta, intra ? 0 : tc0[lane / per]);
                        if (m & 1)  c[-3 * step] = (uint8_t)v.p2;
                        if (m & 2)  c[-2 * step] = (uint8_t)v.p1;
                        if (m & 4)  c[-step] = (uint8_t)v.p0;
                        if (m & 8)  c[0] = (uint8_t)v.q0;
                        if (m & 16) c[step] = (uint8_t)v.q1;
                        if (m & 32) c[2 * step] = (uint8_t)v.q2;
                    }
                }
                mb_wave_sync(); /* the next call reads what this one wrote, through other lanes */
            }
            at += n;
            if (at >= end)
                break;
            n = min(64, end - at);
            if (lane < n) {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(calls + at + lane);
                c0 = src[0]; c1 = src[1]; c2 = src[2];
            }
        }
        /* the dwords that changed, back into the picture */
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int d = lane + 64 * k, tl = d / TD, tc = d - tl * TD;
            if (have[k]) {
                const uint32_t cur = reinterpret_cast<const uint32_t *>(tile)[d];
                if (cur != orig[k])
                    mb_st(plane + (ptrdiff_t)(line0 + tl) * stride + col0 + 4 * tc, cur);
            }
        }
        mb_publish(&progress[p], x + 1, lane);
    }
}

/* The same above 8 bits: uint16_t samples (a dword = two of them), alpha / beta / tc0 scaled to the depth (h264dsp_template.c:104-330 with
 * BIT_DEPTH > 8), the filters' clips at (1 << depth) - 1 */
__global__ __launch_bounds__(64) void k_h264_mbaff_deblock_hbd(MbaffLfArgs A, int *progress_all, int *fail, int bd)
{
    __shared__ __align__(16) uint16_t tile[40 * 20];
    __shared__ __align__(16) uint32_t C[64 * 3];
    const int p = (int)blockIdx.x, pl = (int)blockIdx.y, lane = (int)threadIdx.x;
    MBAFF_TILE(pl != 0, W, H, AB, TP);
    const int TD = TP / 2, TL = H + AB, ND = TL * TD; /* dwords per tile line (two samples each), lines, dwords (400 / 120) */
    const int sh = bd - 8, maxv = (1 << bd) - 1;
    uint8_t *const plane = A.plane[pl];
    const ptrdiff_t stride = A.stride[pl];
    const MbaffDevCall *const calls = A.calls[pl];
    const int32_t *const pair_end = A.pair_end[pl];
    const int mb_w = A.mb_w;
    int *const progress = progress_all + pl * A.prow;
    if (!calls)
        return;
    int at = p > 0 ? pair_end[(size_t)p * mb_w - 1] : 0;
    for (int x = 0; x < mb_w; x++) {
        const int end = pair_end[(size_t)p * mb_w + x];
        if (at >= end) {
            if (lane == 0)
                __hip_atomic_store(&progress[p], x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        int n = min(64, end - at);
        uint32_t c0 = 0, c1 = 0, c2 = 0;
        if (lane < n) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(calls + at + lane);
            c0 = src[0]; c1 = src[1]; c2 = src[2];
        }
        if (p > 0 && !mb_wait(&progress[p - 1], min(x + 2, mb_w), fail, lane))
            return;
        const int line0 = H * p - AB, col0 = W * x - 4;
        uint32_t orig[7];
        bool have[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int d = lane + 64 * k, tl = d / TD, tc = d - tl * TD;
            have[k] = d < ND && line0 + tl >= 0 && col0 + 2 * tc >= 0;
            orig[k] = 0;
            if (have[k])
                orig[k] = mb_ld(plane + (ptrdiff_t)(line0 + tl) * stride + 2 * (col0 + 2 * tc));
        }
#pragma unroll
        for (int k = 0; k < 7; k++)
            if (lane + 64 * k < ND)
                reinterpret_cast<uint32_t *>(tile)[lane + 64 * k] = orig[k];
        for (;;) {
            if (lane < n) {
                C[3 * lane] = c0; C[3 * lane + 1] = c1; C[3 * lane + 2] = c2;
            }
            mb_wave_sync();
            for (int i = 0; i < n; i++) {
                const uint32_t w0 = C[3 * i], w1 = C[3 * i + 1];
                const int toff = (int)(w0 & 0xFFFF), kf = (int)(w0 >> 16) & 255, alpha = (int)(w0 >> 24) << sh, beta = (int)(w1 & 255) << sh;
                const bool chroma = kf & 2, intra = kf & 4, hfilt = kf & 1;
                const int half = (kf >> 4) & 1;
                const int step = (kf & 8) ? 2 * TP : TP; /* samples from a line of the call to the next */
                const int cls = (chroma ? 1 : 0) + (intra ? 2 : 0);
                const int8_t *tc0 = reinterpret_cast<const int8_t *>(&C[3 * i + 2]);
                const int nl = hfilt ? (chroma ? 8 : 16) >> half : (chroma ? 8 : 16);
                const int per = hfilt ? (chroma ? 2 : 4) >> half : (chroma ? 2 : 4);
                if (lane < nl) {
                    /* h_: the line = row `lane`, its samples one apart; v_: the line = column `lane`, its samples a tile line (or two) apart */
                    uint16_t *c = hfilt ? tile + toff + lane * step : tile + toff + lane;
                    const int xs = hfilt ? 1 : step;
                    const int t0 = intra ? 0 : (int)tc0[lane / per];
                    const int tcs = chroma ? (int)(((uint32_t)(t0 - 1) << sh) + 1u) : t0 * (1 << sh);
                    LfLine v;
                    v.p1 = c[-2 * xs]; v.p0 = c[-xs]; v.q0 = c[0]; v.q1 = c[xs];
                    v.p3 = v.p2 = v.q2 = v.q3 = 0;
                    if (!chroma) {
                        v.p3 = c[-4 * xs]; v.p2 = c[-3 * xs]; v.q2 = c[2 * xs]; v.q3 = c[3 * xs];
                    }
                    const int m = lf_line(v, cls, alpha, beta, tcs, maxv);
                    if (m & 1)  c[-3 * xs] = (uint16_t)v.p2;
                    if (m & 2)  c[-2 * xs] = (uint16_t)v.p1;
                    if (m & 4)  c[-xs] = (uint16_t)v.p0;
                    if (m & 8)  c[0] = (uint16_t)v.q0;
                    if (m & 16) c[xs] = (uint16_t)v.q1;
                    if (m & 32) c[2 * xs] = (uint16_t)v.q2;
                }
                mb_wave_sync();
            }
            at += n;
            if (at >= end)
                break;
            n = min(64, end - at);
            if (lane < n) {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(calls + at + lane);
                c0 = src[0]; c1 = src[1]; c2 = src[2];
            }
        }
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int d = lane + 64 * k, tl = d / TD, tc = d - tl * TD;
            if (have[k]) {
                const uint32_t cur = reinterpret_cast<const uint32_t *>(tile)[d];
                if (cur != orig[k])
                    mb_st(plane + (ptrdiff_t)(line0 + tl) * stride + 2 * (col0 + 2 * tc), cur);
            }
        }
        mb_publish(&progress[p], x + 1, lane);
    }
}

/* FFHipH264Edge -> MbaffDevCall for plane pl at line size `stride`: the call of pair (x, p) placed in that pair's tile.  false: the call
 * reaches outside it (not a call ff_h264_filter_mb() issues for a macroblock of that pair). */
static bool mbaff_place_call(const FFHipH264Edge &e, int pl, int stride, int ps /* bytes per sample */, int x, int p, MbaffDevCall *out)
{
    MBAFF_TILE(pl != 0, W, H, AB, TP);
    const int TL = H + AB;
    const int kind = e.kind & 7, chroma = (kind & 2) != 0, hfilt = kind & 1, half = (e.pad & FFHIP_H264_LF_CALL_MBAFF) != 0;
    const int stl = (e.pad & FFHIP_H264_LF_CALL_FIELD) ? 2 : 1;
    if (e.offset < 0 || chroma != (pl != 0))
        return false;
    const int line = e.offset / stride, colb = e.offset - line * stride;
    if (colb % (4 * ps))
        return false;
    const int tl = line - (H * p - AB), tc = colb / ps - (W * x - 4); /* tile line, tile column (samples) */
    if (tc < 4 || (tc & 3))
        return false;
    if (hfilt) {
        const int nlines = (chroma ? 8 : 16) >> half;
        if (tl < 0 || tl + (nlines - 1) * stl >= TL || tc >= TP)
            return false;
    } else {
        const int up = chroma ? 2 : 4, down = chroma ? 1 : 3, ncols = chroma ? 8 : 16;
        if (half || tl - up * stl < 0 || tl + down * stl >= TL || tc + ncols > TP)
            return false;
    }
    out->toff = (uint16_t)(tl * TP + tc);
    out->kf = (uint8_t)(kind | (stl == 2 ? 8 : 0) | (half ? 16 : 0));
    out->alpha = e.alpha;
    out->beta = e.beta;
    out->pad[0] = out->pad[1] = out->pad[2] = 0;
    memcpy(out->tc0, e.tc0, 4);
    return true;
}

extern "C" int ffhip_h264_mbaff_flush(FFHipH264Mbaff *m, uint8_t *const dst[3], const int stride[3], void *stream_)
{
    if (!m || !dst || !stride || !dst[0] || !dst[1] || !dst[2])
        return FFHIP_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    for (int pl = 0; pl < 3; pl++)
        if (((uintptr_t)dst[pl] | (size_t)stride[pl]) & (m->bd > 8 ? 7 : 3) || stride[pl] <= 0) {
            ffhip_set_error("ffhip_h264_mbaff_flush: planes and line sizes must be aligned to four samples and positive");
            return FFHIP_EINVAL;
        }
    if (stride[1] != stride[2]) {
        ffhip_set_error("ffhip_h264_mbaff_flush: Cb and Cr share a line size");
        return FFHIP_EINVAL;
    }
    FFHipDeviceGuard dg(m->device);
    mbaff_finish(m);
    const int prow = m->mb_h / 2, npairs = m->mb_w * prow;
    /* the calls as the kernel takes them: placed in their pair's tile at this line size */
    std::vector<MbaffDevCall> dcalls[3];
    for (int pl = 0; pl < 3; pl++) {
        dcalls[pl].resize(m->calls[pl].size());
        size_t at = 0;
        for (int q = 0; q < npairs; q++)
            for (; at < (size_t)m->pair_end[pl][(size_t)q]; at++)
                if (!mbaff_place_call(m->calls[pl][at], pl, stride[pl], m->bd > 8 ? 2 : 1, q % m->mb_w, q / m->mb_w, &dcalls[pl][at])) {
                    ffhip_set_error("ffhip_h264_mbaff_flush: plane %d call %zu (offset %d, kind %d, flags %d) lies outside macroblock pair (%d, %d)", pl, at,
                                    m->calls[pl][at].offset, m->calls[pl][at].kind, m->calls[pl][at].pad, q % m->mb_w, q / m->mb_w);
                    return m->last_status = FFHIP_EINVAL;
                }
    }
    /* one device blob: records, geo, coefs, intra_row, then per plane calls + pair_end */
    size_t off[12], total = 0;
    auto place = [&](int i, size_t bytes) { off[i] = total; total += (bytes + 255) & ~(size_t)255; };
    place(0, m->recs.size() * sizeof(FFHipH264IntraMB));
    place(1, m->geo.size() * 4);
    place(2, m->coefs.size() * 2 + 1024); /* (the last run is read in whole dwords) */
    place(3, m->intra_row.size() * 4);
    for (int pl = 0; pl < 3; pl++) {
        place(4 + 2 * pl, m->calls[pl].size() * sizeof(FFHipH264Edge));
        place(5 + 2 * pl, (size_t)npairs * 4);
    }
    if (total > m->dev_sz) {
        mbaff_settle(m);
        if (m->dev)
            (void)hipFree(m->dev);
        m->dev = nullptr;
        m->dev_sz = 0;
        if (hipMalloc(&m->dev, total) != hipSuccess) {
            (void)hipGetLastError();
            ffhip_set_error("ffhip_h264_mbaff_flush: hipMalloc(%zu) failed", total);
            return FFHIP_ENOMEM;
        }
        m->dev_sz = total;
    }
    mbaff_settle(m); /* a flush still in flight reads the lists this one is about to overwrite */
    if (!m->done && hipEventCreateWithFlags(&m->done, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        m->done = nullptr;
        ffhip_set_error("ffhip_h264_mbaff_flush: hipEventCreate failed");
        return m->last_status = FFHIP_EIO;
    }
    uint8_t *b = (uint8_t *)m->dev;
    /* (blocking copies: the lists are small, and the host vectors may be cleared by the next begin() as soon as this call returns) */
    auto up = [&](int i, const void *src, size_t bytes) -> bool { return !bytes || hipMemcpy(b + off[i], src, bytes, hipMemcpyHostToDevice) == hipSuccess; };
    bool ok = up(0, m->recs.data(), m->recs.size() * sizeof(FFHipH264IntraMB)) && up(1, m->geo.data(), m->geo.size() * 4) &&
              up(2, m->coefs.data(), m->coefs.size() * 2) && up(3, m->intra_row.data(), m->intra_row.size() * 4);
    for (int pl = 0; pl < 3 && ok; pl++)
        ok = up(4 + 2 * pl, dcalls[pl].data(), dcalls[pl].size() * sizeof(MbaffDevCall)) && up(5 + 2 * pl, m->pair_end[pl].data(), (size_t)npairs * 4);
    if (!ok) {
        (void)hipGetLastError();
        ffhip_set_error("ffhip_h264_mbaff_flush: uploading the lists failed");
        return m->last_status = FFHIP_EIO;
    }
    /* whatever was launched reads `dev` until this event has passed */
    auto leave = [&](int rc) {
        if (hipEventRecord(m->done, stream) == hipSuccess) {
            m->pending = true;
        } else {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(stream);
        }
        return m->last_status = rc;
    };
    int rc = 0;
    if (!m->recs.empty()) {
        FFHipProgressSlot ps;
        rc = ffhip_progress_acquire(prow, stream, &ps);
        if (rc < 0)
            return leave(rc);
        if (m->bd == 8)
            hipLaunchKernelGGL(k_h264_mbaff_intra<uint8_t>, dim3(prow), dim3(64), 0, stream, dst[0], dst[1], dst[2], (ptrdiff_t)stride[0], (ptrdiff_t)stride[1],
                               m->mb_w, m->mb_h, reinterpret_cast<const FFHipH264IntraMB *>(b + off[0]), reinterpret_cast<const uint32_t *>(b + off[1]),
                               reinterpret_cast<const int32_t *>(b + off[3]), reinterpret_cast<const int16_t *>(b + off[2]), ps.prog, ps.fail, 255);
        else
            hipLaunchKernelGGL(k_h264_mbaff_intra<uint16_t>, dim3(prow), dim3(64), 0, stream, dst[0], dst[1], dst[2], (ptrdiff_t)stride[0],
                               (ptrdiff_t)stride[1], m->mb_w, m->mb_h, reinterpret_cast<const FFHipH264IntraMB *>(b + off[0]),
                               reinterpret_cast<const uint32_t *>(b + off[1]), reinterpret_cast<const int32_t *>(b + off[3]),
                               reinterpret_cast<const int16_t *>(b + off[2]), ps.prog, ps.fail, (1 << m->bd) - 1);
        const hipError_t e = hipGetLastError();
        const int r2 = ffhip_progress_release(&ps, stream, e == hipSuccess);
        if (e != hipSuccess) {
            ffhip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e), __FILE__, __LINE__);
            return leave(FFHIP_EIO);
        }
        if (r2 < 0)
            return leave(r2);
    }
    if (!m->calls[0].empty() || !m->calls[1].empty() || !m->calls[2].empty()) {
        MbaffLfArgs A;
        for (int pl = 0; pl < 3; pl++) {
            A.plane[pl] = dst[pl];
            A.stride[pl] = stride[pl];
            A.calls[pl] = m->calls[pl].empty() ? nullptr : reinterpret_cast<const MbaffDevCall *>(b + off[4 + 2 * pl]);
            A.pair_end[pl] = reinterpret_cast<const int32_t *>(b + off[5 + 2 * pl]);
        }
        A.mb_w = m->mb_w;
        A.prow = prow;
        FFHipProgressSlot ps;
        rc = ffhip_progress_acquire(3 * prow, stream, &ps);
        if (rc < 0)
            return leave(rc);
        if (m->bd == 8)
            hipLaunchKernelGGL(k_h264_mbaff_deblock, dim3(prow, 3), dim3(64), 0, stream, A, ps.prog, ps.fail);
        else
            hipLaunchKernelGGL(k_h264_mbaff_deblock_hbd, dim3(prow, 3), dim3(64), 0, stream, A, ps.prog, ps.fail, m->bd);
        const hipError_t e = hipGetLastError();
        const int r2 = ffhip_progress_release(&ps, stream, e == hipSuccess);
        if (e != hipSuccess) {
            ffhip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e), __FILE__, __LINE__);
            return leave(FFHIP_EIO);
        }
        if (r2 < 0)
            return leave(r2);
    }
    return leave(0);
}
