/*
 * vp9_lf.hip — VP9 loop filter, 8 / 10 / 12 bits, batched: loop_filter() of libavcodec/vp9dsp_template.c:1780-1889,
 * the body of loop_filter_8[wd][dir], loop_filter_16[dir] and loop_filter_mix2[wd1][wd2][dir] (:1891-1966).
 * A record is one 8-sample segment of an edge; 8 lanes per segment, one per sample line; every read of a line happens before
 * any write of it.  The flat filters are evaluated as windows: radius 3 over p3..q3 (radius 7 over p7..q7) around the sample with
 * the ends repeated and the centre counted twice, as running sums.  Segments of one launch must not share samples (all edges of
 * one direction that are at least 16 apart, or checkasm-style tiles): VP9 orders overlapping edges inside a superblock.
 */
#include "common.h"
#include <type_traits>

#include "h264_kernels.h"

static_assert(sizeof(FFHipVp9Edge) == 12, "FFHipVp9Edge is a 12-byte record");
static_assert(sizeof(FFHipVp9LfSb) == 1280, "FFHipVp9LfSb is 320 dwords");

__device__ __forceinline__ int vl_abs(int v) { return v < 0 ? -v : v; }

/* PIX = uint8_t (bd 8) / uint16_t; E, I, H arrive in 8-bit units and are scaled by << (bd - 8), the flatness threshold is
 * 1 << (bd - 8), the filter value clips to bd - 1 signed bits (vp9dsp_template.c:1784-1788,1866-1878); stride and offsets in bytes */
/* one sample line across an edge: px[0..15] = p7 .. p0, q0 .. q7 (only 4..11 are read below 16 wide); E, I, H already scaled to the
 * depth, F = 1 << (bd - 8), fmax = 2^(bd-1) - 1; put(k, v) stores sample k (0..15) */
template <class Put>
__device__ __forceinline__ void vp9_lf_line(const int (&px)[16], int wd, int E, int I, int H, int F, int fmax, int maxv, Put put)
{
    auto clipf = [&](int v) { return min(max(v, -fmax - 1), fmax); };
    auto clipp = [&](int v) { return min(max(v, 0), maxv); };
    const int p3 = px[4], p2 = px[5], p1 = px[6], p0 = px[7], q0 = px[8], q1 = px[9], q2 = px[10], q3 = px[11];
    if (!(vl_abs(p3 - p2) <= I && vl_abs(p2 - p1) <= I && vl_abs(p1 - p0) <= I && vl_abs(q1 - q0) <= I && vl_abs(q2 - q1) <= I &&
          vl_abs(q3 - q2) <= I && vl_abs(p0 - q0) * 2 + (vl_abs(p1 - q1) >> 1) <= E))
        return;
    bool flat_in = wd >= 8, flat_out = wd >= 16;
#pragma unroll
    for (int k = 1; k <= 3; k++)
        flat_in = flat_in && vl_abs(px[7 - k] - p0) <= F && vl_abs(px[8 + k] - q0) <= F;
#pragma unroll
    for (int k = 4; k <= 7; k++)
        flat_out = flat_out && vl_abs(px[7 - k] - p0) <= F && vl_abs(px[8 + k] - q0) <= F;
    if (flat_out && flat_in) {
        /* window sums of radius 7 with clamped ends: s(c+1) = s(c) + px[min(c+8, 15)] - px[max(c-7, 0)] */
        int s = 8 * px[0];
#pragma unroll
        for (int t = 1; t <= 8; t++)
            s += px[t]; /* window of c = 1: indices -6..8 -> seven times px[0] (one of them is index 0 itself) + px[1..8] */
        s -= px[0];
#pragma unroll
        for (int c = 1; c <= 14; c++) {
            put(c, (s + px[c] + 8) >> 4);
            s += px[c + 8 > 15 ? 15 : c + 8] - px[c - 7 < 0 ? 0 : c - 7];
        }
    } else if (flat_in) {
        /* radius 3 over px[4..11] */
        int s = 3 * px[4] + px[5] + px[6] + px[7] + px[8]; /* window of c = 5: indices 2..8 clamped to 4..11 */
#pragma unroll
        for (int c = 5; c <= 10; c++) {
            put(c, (s + px[c] + 4) >> 3);
            s += px[c + 4 > 11 ? 11 : c + 4] - px[c - 3 < 4 ? 4 : c - 3];
        }
    } else {
        const bool hev = vl_abs(p1 - p0) > H || vl_abs(q1 - q0) > H;
        int f = clipf(3 * (q0 - p0) + (hev ? clipf(p1 - q1) : 0));
        const int f1 = min(f + 4, fmax) >> 3, f2 = min(f + 3, fmax) >> 3;
        put(7, clipp(p0 + f2));
        put(8, clipp(q0 - f1));
        if (!hev) {
            f = (f1 + 1) >> 1;
            put(6, clipp(p1 + f));
            put(9, clipp(q1 - f));
        }
    }
}

/* PIX = uint8_t (bd 8) / uint16_t; E, I, H arrive in 8-bit units and are scaled by << (bd - 8), the flatness threshold is
 * 1 << (bd - 8), the filter value clips to bd - 1 signed bits (vp9dsp_template.c:1784-1788,1866-1878); stride and offsets in bytes.
 * One sample line per lane, 8 lanes per record.  A column edge's line (dir 0) is contiguous: when it is dword aligned it is read
 * as 2 or 4 dwords (8-byte words at 16 bits) and the words that changed are written back. */
template <typename PIX>
__device__ __forceinline__ void vp9_lf_lines(uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, int e, int line, int bd)
{
    constexpr int PS = (int)sizeof(PIX), SPW = PS == 1 ? 4 : 2; /* samples per dword */
    if (e >= n)
        return;
    const FFHipVp9Edge ed = edges[e];
    const int wd = ed.wd_idx == 0 ? 4 : ed.wd_idx == 1 ? 8 : 16;
    const ptrdiff_t st = stride / (ptrdiff_t)sizeof(PIX), along = ed.dir ? 1 : st, across = ed.dir ? st : 1;
    PIX *pix = reinterpret_cast<PIX *>(base + ed.offset) + line * along;
    const int sh = bd - 8;
    int px[16]; /* p7 .. p0, q0 .. q7 */
    const bool wide = !ed.dir && !(reinterpret_cast<uintptr_t>(pix) & 3);
    uint32_t *w32 = reinterpret_cast<uint32_t *>(pix - 8);
    if (wide) {
#pragma unroll
        for (int q = 0; q < 16 / SPW; q++) {
            const bool need = wd >= 16 || (q >= 4 / SPW && q < 12 / SPW);
            const uint32_t v = need ? w32[q] : 0;
            if constexpr (PS == 1) {
                px[4 * q] = v & 255; px[4 * q + 1] = (v >> 8) & 255; px[4 * q + 2] = (v >> 16) & 255; px[4 * q + 3] = v >> 24;
            } else {
                px[2 * q] = v & 0xFFFF; px[2 * q + 1] = v >> 16;
            }
        }
    } else {
        /* a row edge's line (or an unaligned column edge's): sample by sample, stores where the filter decides them — the 8 lanes
         * of a record touch 8 adjacent bytes per row, which coalesce */
#pragma unroll
        for (int k = 0; k < 16; k++)
            px[k] = (wd >= 16 || (k >= 4 && k < 12)) ? pix[(k - 8) * across] : 0;
        vp9_lf_line(px, wd, ed.E << sh, ed.I << sh, ed.H << sh, 1 << sh, (1 << (bd - 1)) - 1, (1 << bd) - 1,
                    [&](int k, int v) { pix[(k - 8) * across] = (PIX)v; });
        return;
    }
    int out[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        out[k] = px[k];
    unsigned ch = 0;
    vp9_lf_line(px, wd, ed.E << sh, ed.I << sh, ed.H << sh, 1 << sh, (1 << (bd - 1)) - 1, (1 << bd) - 1, [&](int k, int v) {
        out[k] = v;
        ch |= 1u << k;
    });
    if (!ch)
        return;
#pragma unroll
    for (int q = 0; q < 16 / SPW; q++)
        if (ch >> (SPW * q) & ((1u << SPW) - 1)) {
            if constexpr (PS == 1)
                w32[q] = (uint32_t)out[4 * q] | (uint32_t)out[4 * q + 1] << 8 | (uint32_t)out[4 * q + 2] << 16 | (uint32_t)out[4 * q + 3] << 24;
            else
                w32[q] = (uint32_t)out[2 * q] | (uint32_t)out[2 * q + 1] << 16;
        }
}

/* (a lane per 4 columns with dword rows for the row edges, the layout that doubled HEVC's horizontal edges, measured SLOWER here —
 * 2.5 vs 2.8 Tpixel/s: the 8 adjacent bytes a record's lanes read per row already coalesce, and the filter is heavier) */
template <typename PIX>
__global__ __launch_bounds__(256) void k_vp9_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, int bd)
{
    vp9_lf_lines<PIX>(base, stride, edges, n, (int)((blockIdx.x * 256 + threadIdx.x) >> 3), threadIdx.x & 7, bd);
}

int ffhip_launch_vp9_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, hipStream_t stream)
{
    return ffhip_launch_vp9_loop_filter_bd(8, base, stride, edges, n, stream);
}

int ffhip_launch_vp9_loop_filter_bd(int bd, uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (bd == 8)
        hipLaunchKernelGGL(k_vp9_loop_filter<uint8_t>, dim3(cdiv(n, 32)), dim3(256), 0, stream, base, stride, edges, n, 8);
    else if ((bd == 10 || bd == 12) && !(((uintptr_t)base | (size_t)stride) & 1))
        hipLaunchKernelGGL(k_vp9_loop_filter<uint16_t>, dim3(cdiv(n, 32)), dim3(256), 0, stream, base, stride, edges, n, bd);
    else {
        ffhip_set_error("ffhip_vp9_loop_filter: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}

/* ================================================================================================== */
/*
 * k_vp9_lf_frame — the loop filter of a picture in the decoder's order: ff_vp9_loopfilter_sb() (libavcodec/vp9lpf.c:180-203) for
 * every superblock, superblocks in raster order.  A superblock's filters rewrite samples of its left and upper neighbours (a
 * 16-wide filter reaches 8 samples either way), and the upper-right neighbour's column filters reach into the rows the
 * superblock's first row edge reads: superblock (x, y) needs (x - 1, y) and (x + 1, y - 1) — the wavefront of the H.264 kernels.
 * One wave per superblock row walks left to right; rows hand off through a progress counter with the deblocking kernels'
 * protocol (device-scope loads / write-through stores, acknowledged before the counter moves).
 * Inside a superblock the host's tables (host/vp9_lf_tables.c) say, per edge position and 8-line segment, which filter runs.
 * Column edges: lane = sample ROW, which walks its row's positions left to right by itself — different rows never share a sample,
 * so the whole column pass of the 64 rows needs no synchronisation; then one barrier, then the row edges with lane = sample COLUMN.
 * U and V ride in the same wave (lanes 0..31 / 32..63).  The superblock lives in an LDS tile with its 8 context samples to the
 * left and above; everything it may have changed is written back once.
 */
/* a rectangle of NR rows x ND dwords between the picture and an LDS tile, one dword per lane and step, all loads of a region in
 * flight before the first lands in LDS; only rows < nr and dwords < nd exist.  DEV: device-scope (the samples another wave wrote) */
template <int ND, int NR, bool DEV>
struct Vp9LfRegion {
    static constexpr int K = (ND * NR + 63) / 64;
    __device__ __forceinline__ static void issue(uint32_t (&v)[K], const uint8_t *g, ptrdiff_t gstride, int nr, int nd, int lane)
    {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int t = lane + 64 * k, r = t / ND, d = t % ND;
            v[k] = 0;
            if (r < nr && d < nd && t < ND * NR) {
                const uint32_t *a = reinterpret_cast<const uint32_t *>(g + r * gstride + 4 * d);
                v[k] = DEV ? __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *a;
            }
        }
    }
    template <typename LP>
    __device__ __forceinline__ static void commit(const uint32_t (&v)[K], LP lds, int pitch, int nr, int nd, int lane)
    {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int t = lane + 64 * k, r = t / ND, d = t % ND;
            if (r < nr && d < nd && t < ND * NR)
                lds[r * pitch + d] = v[k];
        }
    }
    /* picture <- tile; rows >= wt0 write-through (another workgroup of this launch reads them), the rest plain stores */
    template <typename LP>
    __device__ __forceinline__ static void store(LP lds, int pitch, uint8_t *g, ptrdiff_t gstride, int nr, int nd, int lane, int wt0 = 0)
    {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int t = lane + 64 * k, r = t / ND, d = t % ND;
            if (r < nr && d < nd && t < ND * NR) {
                uint32_t *a = reinterpret_cast<uint32_t *>(g + r * gstride + 4 * d);
                if (r >= wt0)
                    __hip_atomic_store(a, lds[r * pitch + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    *a = lds[r * pitch + d];
            }
        }
    }
};

/* one superblock row of luma (CHROMA = false: 64 lanes = the 64 lines of one plane) or of both chroma planes (CHROMA = true: U in
 * lanes 0..31, V in 32..63).  Luma and chroma never meet in the loop filter, so they are separate waves with separate counters:
 * the critical path of a picture is the luma chain alone. */
template <typename PIX, bool CHROMA>
__device__ __forceinline__ void vp9_lf_sb_row(uint8_t *p0, uint8_t *p1, ptrdiff_t stride, int cols, int rows, int row, const FFHipVp9LfSb *tabs,
                                              int *progress, int *fail, int bd)
{
    constexpr int PS = (int)sizeof(PIX), SPD = 4 / PS;             /* samples per dword */
    constexpr int N = CHROMA ? 32 : 64, NP = CHROMA ? 2 : 1;       /* samples per superblock side, planes in the wave */
    constexpr int P = CHROMA ? 44 : 76;                            /* tile row pitch in samples: an odd dword count at 8 bits */
    constexpr int NPOS = N / 4, NSEG = N / 8, TW = 2 * NPOS * NSEG; /* edge positions, 8-line segments, table words */
    constexpr int TOFF = CHROMA ? 256 : 0, TK = (TW + 63) / 64;
    constexpr int DN = N / SPD, D8 = 8 / SPD, PD = P / SPD;
    __shared__ __align__(16) PIX tile[NP][(N + 8) * P];            /* rows -8..N-1, columns -8..N-1: sample (r, c) at [(r + 8) * P + c + 8] */
    __shared__ uint32_t tab[TW];                                   /* [0 column / 1 row edges][position][segment] */
    using In = Vp9LfRegion<DN, N, false>;   /* the superblock's own samples: nobody has touched them in this launch yet */
    using Left = Vp9LfRegion<D8, N, true>;  /* 8 columns of the left neighbour: this wave's previous step rewrote them */
    using Top = Vp9LfRegion<DN, 8, true>;   /* 8 rows of the upper neighbour: the row above rewrote them */
    using Top7 = Vp9LfRegion<DN, 7, true>;
    const int lane = threadIdx.x, pl = CHROMA ? lane >> 5 : 0, line = CHROMA ? lane & 31 : lane;
    const int sh = bd - 8, F = 1 << sh, fmax = (1 << (bd - 1)) - 1, maxv = (1 << bd) - 1;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int sb_cols = (cols + 7) >> 3;
    /* the picture may end inside the last superblocks: nothing is read or written beyond its cols x rows 8x8 blocks (the
     * reference never does, and frame buffers are not padded to superblocks) */
    const int h = min(N, (CHROMA ? 4 : 8) * rows - N * row);
    uint8_t *const prow[2] = { p0 + (ptrdiff_t)row * N * stride, p1 + (ptrdiff_t)row * N * stride };
    uint32_t *t32[2] = { reinterpret_cast<uint32_t *>(tile[0]), reinterpret_cast<uint32_t *>(tile[NP - 1]) };
    /* the next superblock's own samples and tables travel while this one is filtered */
    uint32_t nin[NP][In::K], ntab[TK];
    auto prefetch = [&](int col) {
        const int w = min(N, (CHROMA ? 4 : 8) * cols - N * col);
#pragma unroll
        for (int q = 0; q < NP; q++)
            In::issue(nin[q], prow[q] + (ptrdiff_t)col * N * PS, stride, h, w / SPD, lane);
        const uint32_t *g = reinterpret_cast<const uint32_t *>(tabs + (size_t)row * sb_cols + col) + TOFF;
#pragma unroll
        for (int k = 0; k < TK; k++)
            ntab[k] = lane + 64 * k < TW ? g[lane + 64 * k] : 0;
    };
    prefetch(0);
    int known = 0;
    for (int col = 0; col < sb_cols; col++) {
        const int w = min(N, (CHROMA ? 4 : 8) * cols - N * col);
        uint8_t *sb[2] = { prow[0] + (ptrdiff_t)col * N * PS, prow[1] + (ptrdiff_t)col * N * PS };
        /* ---- tile <- picture: the prefetched N x N and the 8 columns to the left (right of the first superblock); the column
         *      edges need no more ---- */
        {
            uint32_t vl[NP][Left::K];
#pragma unroll
            for (int q = 0; q < NP; q++)
                if (col)
                    Left::issue(vl[q], sb[q] - 8 * PS, stride, h, D8, lane);
#pragma unroll
            for (int k = 0; k < TK; k++)
                if (lane + 64 * k < TW)
                    tab[lane + 64 * k] = ntab[k];
#pragma unroll
            for (int q = 0; q < NP; q++) {
                In::commit(nin[q], t32[q] + 8 * PD + D8, PD, h, w / SPD, lane);
                if (col)
                    Left::commit(vl[q], t32[q] + 8 * PD, PD, h, D8, lane);
            }
        }
        wave_sync();
        if (col + 1 < sb_cols)
            prefetch(col + 1);
        /* one line of one entry: line0 = the line's sample at position 0 of the filter axis, step = distance along that axis */
        auto run = [&](PIX *line0, int step, int pos, uint32_t e) {
            const int wd = ((e >> 24) & 3) == 0 ? 4 : ((e >> 24) & 3) == 1 ? 8 : 16;
            PIX *pix = line0 + 4 * pos * step;
            int px[16];
#pragma unroll
            for (int k = 0; k < 16; k++)
                px[k] = (wd >= 16 || (k >= 4 && k < 12)) ? (int)pix[(k - 8) * step] : 0;
            vp9_lf_line(px, wd, (int)(e & 0xFF) << sh, (int)((e >> 8) & 0xFF) << sh, (int)((e >> 16) & 0xFF) << sh, F, fmax, maxv,
                        [&](int k, int v) { pix[(k - 8) * step] = (PIX)v; });
        };
        /* ---- column edges: lane = sample row, walking its row's positions left to right by itself ---- */
        for (int p = 0; p < NPOS; p++) {
            const uint32_t e = tab[p * NSEG + (line >> 3)];
            if (e >> 31)
                run(tile[pl] + (line + 8) * P + 8, 1, p, e);
        }
        /* ---- only the row edges read (and rewrite) the upper neighbour's last rows: the wait for the row above — it must have
         *      finished superblock col + 1, whose column edges reach into those rows — sits behind the column pass, which it thus
         *      overlaps; then the 8 rows above (the corner is never read) ---- */
        if (row > 0) {
            const int want = min(col + 2, sb_cols);
            int spins = 0;
            while (known < want) {
                known = __hip_atomic_load(&progress[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (known >= want)
                    break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 24)) { /* never in a correct run; do not hang the device */
                    if (lane == 0)
                        __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (row) {
            uint32_t vt[NP][Top::K];
#pragma unroll
            for (int q = 0; q < NP; q++)
                Top::issue(vt[q], sb[q] - 8 * stride, stride, 8, w / SPD, lane);
#pragma unroll
            for (int q = 0; q < NP; q++)
                Top::commit(vt[q], t32[q] + D8, PD, 8, w / SPD, lane);
        }
        wave_sync();
        /* ---- row edges: lane = sample column ---- */
        for (int p = 0; p < NPOS; p++) {
            const uint32_t e = tab[(NPOS + p) * NSEG + (line >> 3)];
            if (e >> 31)
                run(tile[pl] + 8 * P + line + 8, P, p, e);
        }
        wave_sync();
        /* ---- picture <- tile: rows 0..N-1 x columns -8..N-1 (column edges reach into the left neighbour) and rows -7..-1 x
         *      columns 0..N-1 (row edges reach into the upper one); write-through ---- */
#pragma unroll
        for (int q = 0; q < NP; q++) {
            In::store(t32[q] + 8 * PD + D8, PD, sb[q], stride, h, w / SPD, lane);
            if (col)
                Left::store(t32[q] + 8 * PD, PD, sb[q] - 8 * PS, stride, h, D8, lane);
            if (row)
                Top7::store(t32[q] + PD + D8, PD, sb[q] - 7 * stride, stride, 7, w / SPD, lane);
        }
        /* acknowledged before the counter moves */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0)
            __hip_atomic_store(&progress[row], col + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_sync(); /* the tile and the tables are rewritten by the next step */
    }
}

/* ================================================================================================== */
/*
 * 4:2:2 / 4:4:0 (VP9 profiles 1 / 3 with ss_h != ss_v; round 4): a chroma superblock is 32 x 64 resp. 64 x 32 samples, filter_plane_cols /
 * _rows run with the two shifts apart (libavcodec/vp9lpf.c:27-178,185-201) — vp9_lf_sb_row's walk on a rectangular tile, one plane per
 * wave (the row pass of a 64-wide tile needs all 64 lanes): the luma plane keeps vp9_lf_sb_row<., false>, U and V are waves of their own
 * with counters of their own.  Tables: FFHipVp9LfSbC (host/vp9_lf_tables.c ffhip_vp9_lf_sb_ctables).
 */
template <typename PIX, int NW, int NH>
__device__ __forceinline__ void vp9_lf_plane_row(uint8_t *p0, ptrdiff_t stride, int cols, int rows, int row, const FFHipVp9LfSbC *tabs, int *progress,
                                                 int *fail, int bd)
{
    constexpr int PS = (int)sizeof(PIX), SPD = 4 / PS;
    constexpr int P = NW + 12;                                       /* tile row pitch in samples */
    constexpr int NPC = NW / 4, NSC = NH / 8, NPR = NH / 4, NSR = NW / 8, TW = NPC * NSC + NPR * NSR; /* column / row edge positions, segments */
    constexpr int TK = (TW + 63) / 64, DN = NW / SPD, D8 = 8 / SPD, PD = P / SPD;
    constexpr int UW = 64 / NW, UH = 64 / NH;                         /* 8x8 luma blocks per chroma sample step: samples = (8 / U) per block */
    static_assert(TW == 128 && sizeof(FFHipVp9LfSbC) == 4 * TW, "one table per superblock");
    __shared__ __align__(16) PIX tile[(NH + 8) * P];                  /* rows -8..NH-1, columns -8..NW-1: sample (r, c) at [(r + 8) * P + c + 8] */
    __shared__ uint32_t tab[TW];
    using In = Vp9LfRegion<DN, NH, false>;
    using Left = Vp9LfRegion<D8, NH, true>;
    using Top = Vp9LfRegion<DN, 8, true>;
    using Top7 = Vp9LfRegion<DN, 7, true>;
    const int lane = threadIdx.x;
    const int sh = bd - 8, F = 1 << sh, fmax = (1 << (bd - 1)) - 1, maxv = (1 << bd) - 1;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int sb_cols = (cols + 7) >> 3;
    const int h = min(NH, (8 / UH) * rows - NH * row);               /* the picture may end inside the last superblocks */
    uint8_t *const prow = p0 + (ptrdiff_t)row * NH * stride;
    uint32_t *const t32 = reinterpret_cast<uint32_t *>(tile);
    int known = 0;
    for (int col = 0; col < sb_cols; col++) {
        const int w = min(NW, (8 / UW) * cols - NW * col);
        uint8_t *const sb = prow + (ptrdiff_t)col * NW * PS;
        {
            uint32_t vin[In::K], vl[Left::K];
            In::issue(vin, sb, stride, h, w / SPD, lane);
            if (col)
                Left::issue(vl, sb - 8 * PS, stride, h, D8, lane);
            const uint32_t *g = tabs[(size_t)row * sb_cols + col].t;
#pragma unroll
            for (int k = 0; k < TK; k++)
                if (lane + 64 * k < TW)
                    tab[lane + 64 * k] = g[lane + 64 * k];
            In::commit(vin, t32 + 8 * PD + D8, PD, h, w / SPD, lane);
            if (col)
                Left::commit(vl, t32 + 8 * PD, PD, h, D8, lane);
        }
        wave_sync();
        auto run = [&](PIX *line0, int step, int pos, uint32_t e) {
            const int wd = ((e >> 24) & 3) == 0 ? 4 : ((e >> 24) & 3) == 1 ? 8 : 16;
            PIX *pix = line0 + 4 * pos * step;
            int px[16];
#pragma unroll
            for (int k = 0; k < 16; k++)
                px[k] = (wd >= 16 || (k >= 4 && k < 12)) ? (int)pix[(k - 8) * step] : 0;
            vp9_lf_line(px, wd, (int)(e & 0xFF) << sh, (int)((e >> 8) & 0xFF) << sh, (int)((e >> 16) & 0xFF) << sh, F, fmax, maxv,
                        [&](int k, int v) { pix[(k - 8) * step] = (PIX)v; });
        };
        /* ---- column edges: lane = sample row ---- */
        if (lane < NH)
            for (int p = 0; p < NPC; p++) {
                const uint32_t e = tab[p * NSC + (lane >> 3)];
                if (e >> 31)
                    run(tile + (lane + 8) * P + 8, 1, p, e);
            }
        /* ---- the row above has finished superblock col + 1; then its last 8 rows ---- */
        if (row > 0) {
            const int want = min(col + 2, sb_cols);
            int spins = 0;
            while (known < want) {
                known = __hip_atomic_load(&progress[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (known >= want)
                    break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 24)) {
                    if (lane == 0)
                        __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint32_t vt[Top::K];
            Top::issue(vt, sb - 8 * stride, stride, 8, w / SPD, lane);
            Top::commit(vt, t32 + D8, PD, 8, w / SPD, lane);
        }
        wave_sync();
        /* ---- row edges: lane = sample column ---- */
        if (lane < NW)
            for (int p = 0; p < NPR; p++) {
                const uint32_t e = tab[NPC * NSC + p * NSR + (lane >> 3)];
                if (e >> 31)
                    run(tile + 8 * P + lane + 8, P, p, e);
            }
        wave_sync();
        In::store(t32 + 8 * PD + D8, PD, sb, stride, h, w / SPD, lane);
        if (col)
            Left::store(t32 + 8 * PD, PD, sb - 8 * PS, stride, h, D8, lane);
        if (row)
            Top7::store(t32 + PD + D8, PD, sb - 7 * stride, stride, 7, w / SPD, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0)
            __hip_atomic_store(&progress[row], col + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_sync();
    }
}

/* blocks 0 .. sb_rows - 1: luma rows, then the U rows, then the V rows; ss422: 32 x 64 chroma superblocks (ss_h 1, ss_v 0), else 64 x 32 */
template <typename PIX>
__global__ __launch_bounds__(64) void k_vp9_lf_frame_ssc(uint8_t *py, uint8_t *pu, uint8_t *pv, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                                                         const FFHipVp9LfSb *tabs, const FFHipVp9LfSbC *ctabs, int *progress, int *fail, int bd, int ss422)
{
    const int sb_rows = (rows + 7) >> 3, b = (int)blockIdx.x;
    if (b < sb_rows) {
        vp9_lf_sb_row<PIX, false>(py, py, sy, cols, rows, b, tabs, progress, fail, bd);
        return;
    }
    const int pl = (b - sb_rows) / sb_rows, r = (b - sb_rows) % sb_rows;
    uint8_t *const p = pl ? pv : pu;
    if (ss422)
        vp9_lf_plane_row<PIX, 32, 64>(p, suv, cols, rows, r, ctabs, progress + (1 + pl) * sb_rows, fail, bd);
    else
        vp9_lf_plane_row<PIX, 64, 32>(p, suv, cols, rows, r, ctabs, progress + (1 + pl) * sb_rows, fail, bd);
}

/* N pictures of one geometry side by side (round 5): blockIdx.y = the picture, each with its planes, tables and progress counters */
template <typename PIX>
__global__ __launch_bounds__(64) void k_vp9_lf_frames_ssc(FFHipVp9LfPicsC S, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows, int *progress_all, int *fail,
                                                          int bd, int ss422)
{
    const int sb_rows = (rows + 7) >> 3, b = (int)blockIdx.x;
    uint8_t *const py = S.pic[blockIdx.y].y, *const pu = S.pic[blockIdx.y].u, *const pv = S.pic[blockIdx.y].v;
    const FFHipVp9LfSb *const tabs = S.pic[blockIdx.y].tables;
    const FFHipVp9LfSbC *const ctabs = S.pic[blockIdx.y].ctables;
    int *const progress = progress_all + (size_t)blockIdx.y * (size_t)(3 * sb_rows);
    if (b < sb_rows) {
        vp9_lf_sb_row<PIX, false>(py, py, sy, cols, rows, b, tabs, progress, fail, bd);
        return;
    }
    const int pl = (b - sb_rows) / sb_rows, r = (b - sb_rows) % sb_rows;
    uint8_t *const p = pl ? pv : pu;
    if (ss422)
        vp9_lf_plane_row<PIX, 32, 64>(p, suv, cols, rows, r, ctabs, progress + (1 + pl) * sb_rows, fail, bd);
    else
        vp9_lf_plane_row<PIX, 64, 32>(p, suv, cols, rows, r, ctabs, progress + (1 + pl) * sb_rows, fail, bd);
}

int ffhip_launch_vp9_lf_frames_ssc(int bd, int ss_h, int ss_v, int npics, const FFHipVp9LfPicC *pics, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                                   hipStream_t stream)
{
    const int sb_rows = (rows + 7) >> 3;
    if (cols <= 0 || rows <= 0 || npics <= 0)
        return 0;
    uintptr_t al = (size_t)sy | (size_t)suv;
    for (int i = 0; i < npics; i++) {
        if (!pics[i].y || !pics[i].u || !pics[i].v || !pics[i].tables || !pics[i].ctables)
            return FFHIP_EINVAL;
        al |= (uintptr_t)pics[i].y | (uintptr_t)pics[i].u | (uintptr_t)pics[i].v;
    }
    if ((bd != 8 && bd != 10 && bd != 12) || ss_h == ss_v || ((ss_h | ss_v) & ~1) || (al & 3)) {
        ffhip_set_error("ffhip_vp9_loopfilter_frames_ssc: bit depth %d (8, 10, 12), sub-sampling 1 x 0 or 0 x 1; planes and strides 4-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    const int per_pic = 3 * sb_rows;
    if (per_pic + 1 > FFHIP_PROGRESS_SLOT_INTS)
        return FFHIP_EINVAL;
    int per = (FFHIP_PROGRESS_SLOT_INTS - 1) / per_pic;
    per = per > FFHIP_VP9_LF_PICS ? FFHIP_VP9_LF_PICS : per;
    for (int p0 = 0; p0 < npics; p0 += per) {
        const int n = npics - p0 < per ? npics - p0 : per;
        FFHipProgressSlot ps;
        const int r = ffhip_progress_acquire(n * per_pic + 1, stream, &ps);
        if (r < 0)
            return r;
        FFHipVp9LfPicsC S;
        S.n = n;
        for (int i = 0; i < FFHIP_VP9_LF_PICS; i++)
            S.pic[i] = pics[p0 + (i < n ? i : 0)];
        if (bd == 8)
            hipLaunchKernelGGL(k_vp9_lf_frames_ssc<uint8_t>, dim3(3 * sb_rows, n), dim3(64), 0, stream, S, sy, suv, cols, rows, ps.prog, ps.fail, 8, ss_h);
        else
            hipLaunchKernelGGL(k_vp9_lf_frames_ssc<uint16_t>, dim3(3 * sb_rows, n), dim3(64), 0, stream, S, sy, suv, cols, rows, ps.prog, ps.fail, bd, ss_h);
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

int ffhip_launch_vp9_lf_frame_ssc(int bd, int ss_h, int ss_v, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                                  const FFHipVp9LfSb *tabs, const FFHipVp9LfSbC *ctabs, hipStream_t stream)
{
    const int sb_rows = (rows + 7) >> 3;
    if (cols <= 0 || rows <= 0)
        return 0;
    if ((bd != 8 && bd != 10 && bd != 12) || !y || !u || !v || !tabs || !ctabs || ss_h == ss_v || (ss_h | ss_v) & ~1 ||
        (((uintptr_t)y | (uintptr_t)u | (uintptr_t)v | (size_t)sy | (size_t)suv) & 3)) {
        ffhip_set_error("ffhip_vp9_loopfilter_frame_ssc: bit depth %d (8, 10, 12), sub-sampling 1 x 0 or 0 x 1; planes and strides 4-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    if (3 * sb_rows + 1 > FFHIP_PROGRESS_SLOT_INTS)
        return FFHIP_EINVAL;
    FFHipProgressSlot ps;
    const int r = ffhip_progress_acquire(3 * sb_rows + 1, stream, &ps);
    if (r < 0)
        return r;
    if (bd == 8)
        hipLaunchKernelGGL(k_vp9_lf_frame_ssc<uint8_t>, dim3(3 * sb_rows), dim3(64), 0, stream, y, u, v, sy, suv, cols, rows, tabs, ctabs, ps.prog, ps.fail, 8, ss_h);
    else
        hipLaunchKernelGGL(k_vp9_lf_frame_ssc<uint16_t>, dim3(3 * sb_rows), dim3(64), 0, stream, y, u, v, sy, suv, cols, rows, tabs, ctabs, ps.prog, ps.fail, bd, ss_h);
    const hipError_t e = hipGetLastError();
    const int r2 = ffhip_progress_release(&ps, stream, e == hipSuccess);
    if (e != hipSuccess) {
        ffhip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e), __FILE__, __LINE__);
        return FFHIP_EIO;
    }
    return r2 < 0 ? r2 : 0;
}

/* ---- the frame kernel's line filter: the same arithmetic as vp9_lf_line (vp9dsp_template.c:1777-1930) without per-lane control
 * flow.  A lane is a line of its own 8-line segment with its own width and limits, so a divergent `if` per test made the wave walk
 * every path behind exec-mask bookkeeping (530 VALU + 560 SALU + 107 branches per call site).  Here every test is a sign —
 * |a - b| <= t  <=>  v_sad_u32(a, b, ~t) < 0, a conjunction the sign of a v_max3_i32 — an absent filter (entry not valid, width below
 * 8 / 16) is a limit no difference meets, the 4-tap filter runs on every lane, the 8- and 16-wide ones behind one wave-uniform branch
 * each, and the caller stores by three nested masks (fm: p1..q1, flat: p2 / q2, flat16: p6..p3 / q3..q6). ---- */
__device__ __forceinline__ int vl_sad3(int a, int b, int c)
{
    int d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int vl_max3(int a, int b, int c)
{
    int d;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int vl_med3(int a, int lo, int hi)
{
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(lo), "v"(hi));
    return d;
}

/* o[1..14] <- the filtered line (o[k] = px[k] where nothing changes); mfm / mfi / mfo < 0: the line passes the filter mask / is also
 * flat over 8 / is also flat over 16.  ANY16: some lane of the wave has a 16-wide entry (px[0..3], px[12..15] are loaded) */
template <bool ANY16>
__device__ __forceinline__ void vp9_lf_line2(const int (&px)[16], int (&o)[16], uint32_t e, int sh, int F, int fmax, int maxv, int &mfm, int &mfi,
                                             int &mfo)
{
    constexpr int BIG = 1 << 24;
    const bool valid = (e >> 31) != 0;
    const int wdc = (e >> 24) & 3;
    const int nE = ~((int)(e & 0xFF) << sh), nH = ~((int)((e >> 16) & 0xFF) << sh);
    const int nI = valid ? ~((int)((e >> 8) & 0xFF) << sh) : BIG;
    const int nF8 = wdc >= 1 ? ~F : BIG, nF16 = wdc >= 2 ? ~F : BIG;
    const int p3 = px[4], p2 = px[5], p1 = px[6], p0 = px[7], q0 = px[8], q1 = px[9], q2 = px[10], q3 = px[11];
    const int nfmax = ~fmax, zero = 0;
#pragma unroll
    for (int k = 0; k < 16; k++)
        o[k] = px[k];
    /* fm: |p3-p2|, |p2-p1|, |p1-p0|, |q1-q0|, |q2-q1|, |q3-q2| <= I && 2 |p0-q0| + (|p1-q1| >> 1) <= E */
    const int d10p = vl_sad3(p1, p0, 0), d10q = vl_sad3(q1, q0, 0);
    const int t = (vl_sad3(p0, q0, 0) << 1) + (vl_sad3(p1, q1, 0) >> 1) + nE;
    mfm = vl_max3(vl_max3(vl_sad3(p3, p2, nI), vl_sad3(p2, p1, nI), d10p + nI), vl_max3(d10q + nI, vl_sad3(q2, q1, nI), vl_sad3(q3, q2, nI)), t);
    /* flat over 8: |p1-p0|, |p2-p0|, |p3-p0|, |q1-q0|, |q2-q0|, |q3-q0| <= F (width >= 8) */
    mfi = vl_max3(vl_max3(d10p + nF8, vl_sad3(p2, p0, nF8), vl_sad3(p3, p0, nF8)), vl_max3(d10q + nF8, vl_sad3(q2, q0, nF8), vl_sad3(q3, q0, nF8)), mfm);
    mfo = 0;
    {   /* the 4-tap filter, every lane; selected where fm holds (the flat filters overwrite their lanes below) */
        const int hev = max(d10p, d10q) + nH; /* >= 0: high edge variance */
        const int c = vl_med3(p1 - q1, nfmax, fmax);
        int f = vl_med3(3 * (q0 - p0) + (hev >= 0 ? c : 0), nfmax, fmax);
        const int f1 = min(f + 4, fmax) >> 3, f2 = min(f + 3, fmax) >> 3;
        const int g = (f1 + 1) >> 1;
        const int p0n = vl_med3(p0 + f2, zero, maxv), q0n = vl_med3(q0 - f1, zero, maxv);
        const int p1n = vl_med3(p1 + g, zero, maxv), q1n = vl_med3(q1 - g, zero, maxv);
        const bool fm = mfm < 0, soft = max(mfm, hev) < 0; /* fm && !hev */
        o[7] = fm ? p0n : p0;
        o[8] = fm ? q0n : q0;
        o[6] = soft ? p1n : p1;
        o[9] = soft ? q1n : q1;
    }
    if (__builtin_amdgcn_ballot_w64(mfi < 0)) {
        /* radius 3 over px[4..11]: window sums with clamped ends */
        int v8[6];
        int sum = 3 * px[4] + px[5] + px[6] + px[7] + px[8]; /* window of c = 5: indices 2..8 clamped to 4..11 */
#pragma unroll
        for (int cidx = 5; cidx <= 10; cidx++) {
            v8[cidx - 5] = (sum + px[cidx] + 4) >> 3;
            sum += px[cidx + 4 > 11 ? 11 : cidx + 4] - px[cidx - 3 < 4 ? 4 : cidx - 3];
        }
        asm("" : "+v"(v8[0]), "+v"(v8[1]), "+v"(v8[2]), "+v"(v8[3]), "+v"(v8[4]), "+v"(v8[5]));
        const bool fi = mfi < 0;
#pragma unroll
        for (int cidx = 5; cidx <= 10; cidx++)
            o[cidx] = fi ? v8[cidx - 5] : o[cidx];
    }
    if (ANY16) {
        /* flat over 16: |p4..p7 - p0|, |q4..q7 - q0| <= F (width 16) */
        mfo = vl_max3(vl_max3(vl_sad3(px[3], p0, nF16), vl_sad3(px[2], p0, nF16), vl_sad3(px[1], p0, nF16)),
                      vl_max3(vl_sad3(px[0], p0, nF16), vl_sad3(px[12], q0, nF16), vl_sad3(px[13], q0, nF16)),
                      vl_max3(vl_sad3(px[14], q0, nF16), vl_sad3(px[15], q0, nF16), mfi));
        if (__builtin_amdgcn_ballot_w64(mfo < 0)) {
            int v16[14];
            /* window sums of radius 7 with clamped ends: s(c+1) = s(c) + px[min(c+8, 15)] - px[max(c-7, 0)] */
            int sum = 8 * px[0];
#pragma unroll
            for (int tdx = 1; tdx <= 8; tdx++)
                sum += px[tdx];
            sum -= px[0];
#pragma unroll
            for (int cidx = 1; cidx <= 14; cidx++) {
                v16[cidx - 1] = (sum + px[cidx] + 8) >> 4;
                sum += px[cidx + 8 > 15 ? 15 : cidx + 8] - px[cidx - 7 < 0 ? 0 : cidx - 7];
            }
            const bool fo = mfo < 0;
#pragma unroll
            for (int cidx = 1; cidx <= 14; cidx++) {
                asm("" : "+v"(v16[cidx - 1]));
                o[cidx] = fo ? v16[cidx - 1] : o[cidx];
            }
        }
    }
}

/* ================================================================================================== */
/*
 * k_vp9_lf_frame_wg — the same order with W superblock rows per WORKGROUP (round 3; what the H.264 deblocking kernel taught):
 *   - the order needs less than "superblock (x + 1, y - 1) finished": the ROW edges of (x, y) need (x, y - 1) complete and the COLUMN
 *     edges of (x + 1, y - 1) — rows trail each other by one superblock, not two;
 *   - the W waves of a workgroup keep their superblocks in ONE LDS stack per parity of the column (rows 8 + W N, wave w at rows
 *     8 + w N ...): wave w's 8 context rows ARE the last rows of wave w - 1's superblock of the same column, in place.  Nothing of a
 *     hand-off inside the workgroup touches memory: two LDS counters per wave (cdone: column passes complete, read by the wave below
 *     before its row pass; hdone: steps complete, read by the wave above before it reuses a stack);
 *   - a superblock's last 8 columns are filtered by the NEXT step's column edges: they are copied into the next stack's context
 *     columns before that pass and back after it (8 samples per lane), so every filter address stays a compile-time offset;
 *   - every picture sample is written once, with plain stores: a wave stores rows -8 .. -1 of superblock c (the rows above it, final
 *     after its row pass) at the end of step c and its own rows 0 .. N - 9 of superblock c - 1 after the column pass of step c; only
 *     the last wave of a workgroup (and of the picture) also stores the last 8 rows — write-through, acknowledged, then the
 *     agent-scope counter the next workgroup's first wave polls (the old kernel's protocol, once per W rows).
 */
typedef __attribute__((address_space(3))) uint8_t vl_lds_u8;
typedef __attribute__((address_space(3))) uint32_t vl_lds_u32;
typedef __attribute__((address_space(3))) int vl_lds_int;
__device__ __forceinline__ bool vl_wait_lds(const vl_lds_int *ctr, int want, int *fail)
{
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { /* never in a correct run; do not hang the device */
            if ((threadIdx.x & 63) == 0)
                __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}

template <typename PIX, bool CHROMA>
__device__ __forceinline__ void vp9_lf_sb_rows(vl_lds_u8 *lds, int W, uint8_t *p0, uint8_t *p1, ptrdiff_t stride, int cols, int rows, int row0,
                                               int sb_rows, const FFHipVp9LfSb *tabs, int *progress, int *fail, int bd, int fault)
{
    constexpr int PS = (int)sizeof(PIX), SPD = 4 / PS;             /* samples per dword */
    constexpr int N = CHROMA ? 32 : 64, NP = CHROMA ? 2 : 1;       /* samples per superblock side, planes in the wave */
    constexpr int P = CHROMA ? 44 : 76;                            /* stack row pitch in samples */
    constexpr int NPOS = N / 4, NSEG = N / 8, TW = 2 * NPOS * NSEG; /* edge positions, 8-line segments, table words */
    constexpr int TOFF = CHROMA ? 256 : 0, TK = (TW + 63) / 64;
    constexpr int DN = N / SPD, D8 = 8 / SPD, PD = P / SPD;
    using In = Vp9LfRegion<DN, N, false>;   /* the superblock's own samples: nobody has touched them in this launch yet */
    using Top = Vp9LfRegion<DN, 8, true>;   /* 8 rows of the superblock row above, through memory (first wave of a workgroup) */
    using Rows8 = Vp9LfRegion<DN, 8, false>;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, pl = CHROMA ? lane >> 5 : 0, line = CHROMA ? lane & 31 : lane;
    const int sh = bd - 8, F = 1 << sh, fmax = (1 << (bd - 1)) - 1, maxv = (1 << bd) - 1;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    /* LDS: stacks[parity][plane]: (8 + W N) rows of P samples; then a table per wave; then cdone[W], hdone[W] */
    const int stack_dw = (8 + W * N) * PD;
    typedef __attribute__((address_space(3))) PIX LPIX; /* every tile pointer is an LDS pointer by type: ds_* instructions, 32-bit addresses */
    vl_lds_u32 *const stacks = (vl_lds_u32 *)lds;
    vl_lds_u32 *const tab = stacks + 2 * NP * stack_dw + wv * TW;
    vl_lds_int *const cdone = (vl_lds_int *)(stacks + 2 * NP * stack_dw + W * TW), *const hdone = cdone + W;
    if (threadIdx.x < 2u * (unsigned)W)
        cdone[threadIdx.x] = 0;
    __syncthreads();
    const int row = row0 + wv;
    if (row >= sb_rows)
        return; /* nobody waits for a row below the picture */
    const bool from_lds = wv > 0, from_mem = wv == 0 && row > 0;
    const bool below = row + 1 < sb_rows, to_lds = below && wv < W - 1, to_mem = below && wv == W - 1;
    const int sb_cols = (cols + 7) >> 3;
    /* the picture may end inside the last superblocks: nothing is read or written beyond its cols x rows 8x8 blocks */
    const int h = min(N, (CHROMA ? 4 : 8) * rows - N * row);
    const int hown = to_lds ? N - 8 : h; /* the rows of its superblocks a wave stores itself: the wave below stores the last 8 with its own */
    uint8_t *const prow[2] = { p0 + (ptrdiff_t)row * N * stride, p1 + (ptrdiff_t)row * N * stride };
    /* tile(parity, plane): this wave's superblock, sample (r, c) at [(r + 8) * P + c + 8]; rows -8 .. -1 = the wave above's last rows */
    auto tile32 = [&](int par, int q) { return stacks + (par * NP + q) * stack_dw + wv * N * PD; };
    uint32_t nin[NP][In::K], ntab[TK];
    auto prefetch = [&](int col) {
        const int w = min(N, (CHROMA ? 4 : 8) * cols - N * col);
#pragma unroll
        for (int q = 0; q < NP; q++)
            In::issue(nin[q], prow[q] + (ptrdiff_t)col * N * PS, stride, h, w / SPD, lane);
        const uint32_t *g = reinterpret_cast<const uint32_t *>(tabs + (size_t)row * sb_cols + col) + TOFF;
#pragma unroll
        for (int k = 0; k < TK; k++)
            ntab[k] = lane + 64 * k < TW ? g[lane + 64 * k] : 0;
    };
    prefetch(0);
    int known = 0;
    for (int col = 0; col <= sb_cols; col++) { /* step sb_cols: the row's last superblock leaves */
        const bool live = col < sb_cols;
        const int par = col & 1;
        const int w = live ? min(N, (CHROMA ? 4 : 8) * cols - N * col) : 0;
        vl_lds_u32 *t32[2] = { tile32(par, 0), tile32(par, NP - 1) }, *o32[2] = { tile32(par ^ 1, 0), tile32(par ^ 1, NP - 1) };
        LPIX *tile = (LPIX *)t32[pl];
        uint8_t *sb[2] = { prow[0] + (ptrdiff_t)col * N * PS, prow[1] + (ptrdiff_t)col * N * PS };
        /* ---- this stack held superblock col - 2, whose last rows the wave below reads until the end of its step col - 2 ---- */
        if (live && to_lds && col >= 2 && !(fault & 8) && !vl_wait_lds(&hdone[wv + 1], col - 1, fail))
            return;
        if (live) {
            /* ---- stack <- the prefetched N x N; its 8 context columns <- the last columns of superblock col - 1 ---- */
#pragma unroll
            for (int k = 0; k < TK; k++)
                if (lane + 64 * k < TW)
                    tab[lane + 64 * k] = ntab[k];
#pragma unroll
            for (int q = 0; q < NP; q++)
                In::commit(nin[q], t32[q] + 8 * PD + D8, PD, h, w / SPD, lane);
            if (col) {
                const vl_lds_u32 *s = o32[pl] + (line + 8) * PD + D8 + DN - D8;
                vl_lds_u32 *d = t32[pl] + (line + 8) * PD;
#pragma unroll
                for (int k = 0; k < D8; k++)
                    d[k] = s[k];
            }
            wave_sync();
            if (col + 1 < sb_cols)
                prefetch(col + 1);
        }
        /* one position of a pass: every lane its own line (line0 = the line's sample at position 0 of the filter axis, step = distance
         * along that axis); any16: some lane of the wave has a 16-wide entry here.  (Keeping a 16-sample window of the line in registers
         * across the positions — no LDS round trip between two dependent filters — was SLOWER: the kernel is bound by instruction
         * issue, and the window costs 12 moves per position and filters every lane at every position.) */
        auto run = [&](LPIX *line0, int step, int pos, uint32_t e) {
            LPIX *pix = line0 + 4 * pos * step;
            int px[16], o[16], mfm, mfi, mfo;
            const bool any16 = __builtin_amdgcn_ballot_w64((e >> 31) && ((e >> 24) & 3) == 2) != 0;
#pragma unroll
            for (int k = 4; k < 12; k++)
                px[k] = (int)pix[(k - 8) * step];
            if (any16) {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (k < 4 || k >= 12)
                        px[k] = (int)pix[(k - 8) * step];
                vp9_lf_line2<true>(px, o, e, sh, F, fmax, maxv, mfm, mfi, mfo);
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (k < 4 || k >= 12)
                        px[k] = 0;
                vp9_lf_line2<false>(px, o, e, sh, F, fmax, maxv, mfm, mfi, mfo);
            }
            if (mfm < 0) {
#pragma unroll
                for (int k = 6; k <= 9; k++)
                    pix[(k - 8) * step] = (PIX)o[k];
                if (mfi < 0) {
                    pix[-3 * step] = (PIX)o[5];
                    pix[2 * step] = (PIX)o[10];
                    if (mfo < 0) {
#pragma unroll
                        for (int k = 1; k <= 4; k++) {
                            pix[(k - 8) * step] = (PIX)o[k];
                            pix[(15 - k - 8) * step] = (PIX)o[15 - k];
                        }
                    }
                }
            }
        };
        if (live) {
            /* ---- column edges: lane = sample row, walking its row's positions left to right by itself ---- */
            for (int p = 0; p < NPOS; p++) {
                const uint32_t e = tab[p * NSEG + (line >> 3)];
                if (__builtin_amdgcn_ballot_w64((e >> 31) != 0))
                    run(tile + (line + 8) * P + 8, 1, p, e);
            }
            wave_sync();
            if (col) { /* the left neighbour's last columns go back where the row below and the store pass read them */
                const vl_lds_u32 *s = t32[pl] + (line + 8) * PD;
                vl_lds_u32 *d = o32[pl] + (line + 8) * PD + D8 + DN - D8;
#pragma unroll
                for (int k = 0; k < D8; k++)
                    d[k] = s[k];
            }
        }
        wave_sync();
        if (to_lds && lane == 0 && !(fault & 1)) /* superblock col - 1 of this row is final (at step sb_cols: the last one) */
            __hip_atomic_store(&cdone[wv], col + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        /* ---- picture <- superblock col - 1: the rows this wave owns ---- */
        if (col) {
            const int wp = min(N, (CHROMA ? 4 : 8) * cols - N * (col - 1));
#pragma unroll
            for (int q = 0; q < NP; q++) {
                uint8_t *g = prow[q] + (ptrdiff_t)(col - 1) * N * PS;
                if (to_mem) { /* the last 8 rows write-through: the next workgroup reads them */
                    Vp9LfRegion<DN, N, false>::store(o32[q] + 8 * PD + D8, PD, g, stride, N - 8, wp / SPD, lane, N);
                    Top::store(o32[q] + N * PD + D8, PD, g + (ptrdiff_t)(N - 8) * stride, stride, 8, wp / SPD, lane);
                } else {
                    Vp9LfRegion<DN, N, false>::store(o32[q] + 8 * PD + D8, PD, g, stride, hown, wp / SPD, lane, N);
                }
            }
        }
        if (!live)
            break;
        /* ---- only the row edges read (and rewrite) the last rows of the superblock above: it must have run the column edges of
         *      superblock col + 1, which reach into them ---- */
        if (from_lds && !(fault & 8) && !vl_wait_lds(&cdone[wv - 1], col + 2, fail))
            return;
        if (from_mem) {
            const int want = (fault & 8) ? 0 : col + 1;
            int spins = 0;
            while (known < want) {
                known = __hip_atomic_load(&progress[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (known >= want)
                    break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 24)) { /* never in a correct run; do not hang the device */
                    if (lane == 0)
                        __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint32_t vt[NP][Top::K];
#pragma unroll
            for (int q = 0; q < NP; q++)
                Top::issue(vt[q], sb[q] - 8 * stride, stride, 8, w / SPD, lane);
#pragma unroll
            for (int q = 0; q < NP; q++)
                Top::commit(vt[q], t32[q] + D8, PD, 8, w / SPD, lane);
        }
        wave_sync();
        /* ---- row edges: lane = sample column ---- */
        for (int p = 0; p < NPOS; p++) {
            const uint32_t e = tab[(NPOS + p) * NSEG + (line >> 3)];
            if (__builtin_amdgcn_ballot_w64((e >> 31) != 0))
                run(tile + 8 * P + line + 8, P, p, e);
            /* the last wave of a workgroup: the last rows of superblock col - 1 left (write-through) before this pass began; a few
             * positions later they are acknowledged, and the next workgroup's first wave need not wait for the end of the step */
            if (to_mem && p == NPOS / 4 && col >= 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
                if (lane == 0 && !(fault & 1))
                    __hip_atomic_store(&progress[row], col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        wave_sync();
        /* ---- picture <- rows -8 .. -1 of superblock col: final (the column edges of col + 1 of the row above have run) ---- */
        if (row)
#pragma unroll
            for (int q = 0; q < NP; q++)
                Rows8::store(t32[q] + D8, PD, sb[q] - 8 * stride, stride, 8, w / SPD, lane, 8);
        wave_sync();
        if (from_lds && lane == 0)
            __hip_atomic_store(&hdone[wv], col + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (to_mem && !(fault & 1)) { /* the row's last superblock */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0)
            __hip_atomic_store(&progress[row], sb_cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* blocks 0 .. nwg-1: luma (the long chain first), nwg .. 2 nwg-1: chroma; W superblock rows per block */
/* round 4: blockIdx.y = the picture of a batch (one geometry; each picture its own planes, tables and progress counters): a picture's
 * filter is a dependency chain through it that occupies 34 superblock rows' worth of waves — the pictures a decoder's frame threads
 * hold are filtered side by side for the latency of one */
template <typename PIX>
__global__ __launch_bounds__(256) void k_vp9_lf_frame_wg(FFHipVp9LfPics S, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows, int *progress_all,
                                                        int *fail, int bd, int fault, int planes444)
{
    extern __shared__ __align__(16) uint8_t vl_lds[];
    const int W = (int)(blockDim.x >> 6);
    const int sb_rows = (rows + 7) >> 3, nwg = (sb_rows + W - 1) / W;
    /* (read once: the set is indexed at run time) */
    uint8_t *const py = S.pic[blockIdx.y].y, *const pu = S.pic[blockIdx.y].u, *const pv = S.pic[blockIdx.y].v;
    const FFHipVp9LfSb *const tabs = S.pic[blockIdx.y].tables;
    int *const progress = progress_all + (size_t)blockIdx.y * (size_t)((planes444 ? 3 : 2) * sb_rows);
    if (planes444) { /* 4:4:4: the chroma planes are filtered exactly as luma, with luma's masks and levels (vp9lpf.c:185-201: uv_masks =
                      * lflvl->mask[ss_h | ss_v], filter_plane_cols / _rows with ss 0): three luma chains */
        const int pl = (int)blockIdx.x / nwg, b = (int)blockIdx.x - pl * nwg;
        uint8_t *p = pl == 0 ? py : pl == 1 ? pu : pv;
        vp9_lf_sb_rows<PIX, false>((vl_lds_u8 *)vl_lds, W, p, p, pl ? suv : sy, cols, rows, b * W, sb_rows, tabs, progress + pl * sb_rows, fail, bd, fault);
        return;
    }
    if ((int)blockIdx.x < nwg)
        vp9_lf_sb_rows<PIX, false>((vl_lds_u8 *)vl_lds, W, py, py, sy, cols, rows, blockIdx.x * W, sb_rows, tabs, progress, fail, bd, fault);
    else
        vp9_lf_sb_rows<PIX, true>((vl_lds_u8 *)vl_lds, W, pu, pv, suv, cols, rows, (blockIdx.x - nwg) * W, sb_rows, tabs, progress + sb_rows, fail, bd, fault);
}

/* blocks 0 .. sb_rows-1: luma rows (the long chain first), sb_rows .. 2 sb_rows-1: chroma rows */
template <typename PIX>
__global__ __launch_bounds__(64) void k_vp9_lf_frame(uint8_t *py, uint8_t *pu, uint8_t *pv, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                                                     const FFHipVp9LfSb *tabs, int *progress, int *fail, int bd)
{
    const int sb_rows = (rows + 7) >> 3;
    if ((int)blockIdx.x < sb_rows)
        vp9_lf_sb_row<PIX, false>(py, py, sy, cols, rows, blockIdx.x, tabs, progress, fail, bd);
    else
        vp9_lf_sb_row<PIX, true>(pu, pv, suv, cols, rows, blockIdx.x - sb_rows, tabs, progress + sb_rows, fail, bd);
}

int ffhip_launch_vp9_lf_frame(int bd, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                              const FFHipVp9LfSb *tabs, hipStream_t stream, int planes444)
{
    FFHipVp9LfPic one = { y, u, v, tabs };
    return ffhip_launch_vp9_lf_frames(bd, 1, &one, sy, suv, cols, rows, stream, planes444);
}

int ffhip_launch_vp9_lf_frames(int bd, int npics, const FFHipVp9LfPic *pics, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows, hipStream_t stream,
                               int planes444)
{
    const int sb_rows = (rows + 7) >> 3;
    if (cols <= 0 || rows <= 0 || npics <= 0)
        return 0;
    uintptr_t al = (size_t)sy | (size_t)suv;
    for (int i = 0; i < npics; i++) {
        if (!pics[i].y || !pics[i].u || !pics[i].v || !pics[i].tables)
            return FFHIP_EINVAL;
        al |= (uintptr_t)pics[i].y | (uintptr_t)pics[i].u | (uintptr_t)pics[i].v;
    }
    if ((bd != 8 && bd != 10 && bd != 12) || (al & 3)) {
        ffhip_set_error("ffhip_vp9_loopfilter_frame: bit depth %d (8, 10, 12); planes and strides must be 4-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    const int per_pic = (planes444 ? 3 : 2) * sb_rows;
    if (per_pic + 1 > FFHIP_PROGRESS_SLOT_INTS)
        return FFHIP_EINVAL;
    const char *eo = FFHIP_KNOB("FFHIP_VP9_LF_OLD"); /* 1: one wave per superblock row, every hand-off through memory (cross-check) */
    const char *ew = FFHIP_KNOB("FFHIP_VP9_LF_WPB"), *ef = FFHIP_KNOB("FFHIP_VP9_LF_FAULT");
    const int fault = ef ? atoi(ef) : 0; /* 1: the test hook (no hand-off is published); 8: no waiting (timing experiment, wrong output) */
    const bool old = eo && atoi(eo) == 1 && !planes444;
    int per = old ? 1 : (FFHIP_PROGRESS_SLOT_INTS - 1) / per_pic;
    per = per > FFHIP_VP9_LF_PICS ? FFHIP_VP9_LF_PICS : per;
    for (int p0 = 0; p0 < npics; p0 += per) {
        const int n = npics - p0 < per ? npics - p0 : per;
        FFHipProgressSlot ps;
        const int r = ffhip_progress_acquire(n * per_pic + 1, stream, &ps);
        if (r < 0)
            return r;
        int *const prog = ps.prog, *const fail = ps.fail;
        if (old) {
            const FFHipVp9LfPic &P = pics[p0];
            if (bd == 8)
                hipLaunchKernelGGL(k_vp9_lf_frame<uint8_t>, dim3(2 * sb_rows), dim3(64), 0, stream, P.y, P.u, P.v, sy, suv, cols, rows, P.tables, prog, fail, 8);
            else
                hipLaunchKernelGGL(k_vp9_lf_frame<uint16_t>, dim3(2 * sb_rows), dim3(64), 0, stream, P.y, P.u, P.v, sy, suv, cols, rows, P.tables, prog, fail, bd);
        } else {
            FFHipVp9LfPics S;
            S.n = n;
            for (int i = 0; i < FFHIP_VP9_LF_PICS; i++)
                S.pic[i] = pics[p0 + (i < n ? i : 0)];
            /* superblock rows per workgroup: what 64 KB of LDS hold (two stacks of 8 + 64 W rows, 76 samples wide) */
            const int wmax = bd == 8 ? 4 : 2;
            const int W = ew && atoi(ew) >= 1 && atoi(ew) <= wmax ? atoi(ew) : wmax;
            const int ps_ = bd == 8 ? 1 : 2, nwg = (sb_rows + W - 1) / W;
            const unsigned luma = (2u * (8 + W * 64) * (76 * ps_ / 4) + W * 256u + 2u * W) * 4u, chroma = (4u * (8 + W * 32) * (44 * ps_ / 4) + W * 64u + 2u * W) * 4u;
            const unsigned lds = ((luma > chroma ? luma : chroma) + 15u) & ~15u;
            if (bd == 8)
                hipLaunchKernelGGL(k_vp9_lf_frame_wg<uint8_t>, dim3((planes444 ? 3 : 2) * nwg, n), dim3(64 * W), lds, stream, S, sy, suv, cols, rows, prog, fail, 8, fault, planes444);
            else
                hipLaunchKernelGGL(k_vp9_lf_frame_wg<uint16_t>, dim3((planes444 ? 3 : 2) * nwg, n), dim3(64 * W), lds, stream, S, sy, suv, cols, rows, prog, fail, bd, fault, planes444);
        }
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
