/*
 * vp9_lf.hip — VP9 loop filter, 8 / 10 / 12 bits, batched: loop_filter() of libavcodec/vp9dsp_template.c:1780-1889,
 * the body of loop_filter_8[wd][dir], loop_filter_16[dir] and loop_filter_mix2[wd1][wd2][dir] (:1891-1966).
 * A record is one 8-sample segment of an edge; 8 lanes per segment, one per sample line; every read of a line happens before
 * any write of it.  The flat filters are evaluated as windows: radius 3 over p3..q3 (radius 7 over p7..q7) around the sample with
 * the ends repeated and the centre counted twice, as running sums.  Segments of one launch must not share samples (all edges of
 * one direction that are at least 16 apart, or checkasm-style tiles): VP9 orders overlapping edges inside a superblock.
 */
#include "common.h"
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
 * 1 << (bd - 8), the filter value clips to bd - 1 signed bits (vp9dsp_template.c:1784-1788,1866-1878); stride and offsets in bytes */
template <typename PIX>
__global__ __launch_bounds__(256) void k_vp9_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, int bd)
{
    const int e = (blockIdx.x * 256 + threadIdx.x) >> 3, line = threadIdx.x & 7;
    if (e >= n)
        return;
    const FFHipVp9Edge ed = edges[e];
    const int wd = ed.wd_idx == 0 ? 4 : ed.wd_idx == 1 ? 8 : 16;
    const ptrdiff_t st = stride / (ptrdiff_t)sizeof(PIX), along = ed.dir ? 1 : st, across = ed.dir ? st : 1;
    PIX *pix = reinterpret_cast<PIX *>(base + ed.offset) + line * along;
    const int sh = bd - 8;
    int px[16]; /* p7 .. p0, q0 .. q7 */
#pragma unroll
    for (int k = 0; k < 16; k++)
        px[k] = (wd >= 16 || (k >= 4 && k < 12)) ? pix[(k - 8) * across] : 0;
    vp9_lf_line(px, wd, ed.E << sh, ed.I << sh, ed.H << sh, 1 << sh, (1 << (bd - 1)) - 1, (1 << bd) - 1,
                [&](int k, int v) { pix[(k - 8) * across] = (PIX)v; });
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
template <typename PIX>
struct Vp9LfTile {
    static constexpr int PY = 76, PC = 44;  /* row pitches in samples: odd dword counts at 8 bits */
    PIX y[72 * PY];                         /* rows -8..63, columns -8..63: sample (r, c) at [(r + 8) * PY + c + 8] */
    PIX c[2][40 * PC];                      /* rows -8..31, columns -8..31 */
    uint32_t tab[320];                      /* FFHipVp9LfSb */
};

template <typename PIX>
__global__ __launch_bounds__(64) void k_vp9_lf_frame(uint8_t *py, uint8_t *pu, uint8_t *pv, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                                                     const FFHipVp9LfSb *tabs, int *progress, int *fail, int bd)
{
    using T = Vp9LfTile<PIX>;
    constexpr int PS = (int)sizeof(PIX), SPD = 4 / PS; /* samples per dword */
    __shared__ __align__(16) T tile;
    const int row = blockIdx.x, lane = threadIdx.x;
    const int sh = bd - 8, F = 1 << sh, fmax = (1 << (bd - 1)) - 1, maxv = (1 << bd) - 1;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto ld_dev = [](const uint8_t *p) { return __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st_dev = [](uint8_t *p, uint32_t v) { __hip_atomic_store(reinterpret_cast<uint32_t *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    uint32_t *ty32 = reinterpret_cast<uint32_t *>(tile.y);
    const int sb_cols = (cols + 7) >> 3;
    const int hl = min(64, 8 * rows - 64 * row), hc = hl >> 1; /* the picture may end inside the last superblocks: nothing is read or */
    int known = 0;                                             /* written beyond its cols x rows 8x8 blocks (the reference never does) */
    for (int col = 0; col < sb_cols; col++) {
        const int wl = min(64, 8 * cols - 64 * col), wc = wl >> 1;
        /* ---- the superblock's tables ---- */
        {
            const uint32_t *g = reinterpret_cast<const uint32_t *>(tabs + (size_t)row * sb_cols + col);
#pragma unroll
            for (int k = 0; k < 5; k++)
                tile.tab[lane + 64 * k] = g[lane + 64 * k];
        }
        /* ---- the row above has finished superblock col + 1 ---- */
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
        /* ---- tile <- picture: rows -8..63 (the 8 above only below the first superblock row) x columns -8..63 (the 8 to the
         *      left only right of the first superblock), as dwords; everything else of the tile is never read ---- */
        uint8_t *ysb = py + (ptrdiff_t)row * 64 * sy + (ptrdiff_t)col * 64 * PS;
        uint8_t *csb[2] = { pu + (ptrdiff_t)row * 32 * suv + (ptrdiff_t)col * 32 * PS, pv + (ptrdiff_t)row * 32 * suv + (ptrdiff_t)col * 32 * PS };
        {
            const int r0 = row ? -8 : 0, c0 = col ? -8 : 0;
            const int ndw = (wl - c0) / SPD, nrow = hl - r0;
            for (int t = lane; t < nrow * ndw; t += 64) {
                const int r = r0 + t / ndw, d = t % ndw, c = c0 + d * SPD;
                ty32[((r + 8) * T::PY + c + 8) / SPD] = ld_dev(ysb + (ptrdiff_t)r * sy + (ptrdiff_t)c * PS);
            }
            const int cndw = (wc - c0) / SPD, cnrow = hc - r0;
            for (int t = lane; t < 2 * cnrow * cndw; t += 64) {
                const int p = t / (cnrow * cndw), u = t % (cnrow * cndw), r = r0 + u / cndw, c = c0 + (u % cndw) * SPD;
                reinterpret_cast<uint32_t *>(tile.c[p])[((r + 8) * T::PC + c + 8) / SPD] = ld_dev(csb[p] + (ptrdiff_t)r * suv + (ptrdiff_t)c * PS);
            }
        }
        wave_sync();
        auto entry_ok = [](uint32_t e) { return (e >> 31) != 0; };
        /* one line of one entry: base = the line's first sample (position 0 of the filter axis), step = distance along that axis */
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
        /* ---- column edges: lane = sample row of luma (64) / of U (lanes 0..31) and V (32..63) ---- */
        for (int p = 0; p < 16; p++) {
            const uint32_t e = tile.tab[p * 8 + (lane >> 3)];
            if (entry_ok(e))
                run(tile.y + (lane + 8) * T::PY + 8, 1, p, e);
        }
        for (int p = 0; p < 8; p++) {
            const uint32_t e = tile.tab[256 + p * 4 + ((lane & 31) >> 3)];
            if (entry_ok(e))
                run(tile.c[lane >> 5] + ((lane & 31) + 8) * T::PC + 8, 1, p, e);
        }
        wave_sync();
        /* ---- row edges: lane = sample column ---- */
        for (int p = 0; p < 16; p++) {
            const uint32_t e = tile.tab[128 + p * 8 + (lane >> 3)];
            if (entry_ok(e))
                run(tile.y + 8 * T::PY + lane + 8, T::PY, p, e);
        }
        for (int p = 0; p < 8; p++) {
            const uint32_t e = tile.tab[288 + p * 4 + ((lane & 31) >> 3)];
            if (entry_ok(e))
                run(tile.c[lane >> 5] + 8 * T::PC + (lane & 31) + 8, T::PC, p, e);
        }
        wave_sync();
        /* ---- picture <- tile: rows 0..63 x columns -8..63 (column edges reach into the left neighbour) and rows -7..-1 x
         *      columns 0..63 (row edges reach into the upper one); write-through ---- */
        {
            const int c0 = col ? -8 : 0, ndw = (wl - c0) / SPD;
            for (int t = lane; t < hl * ndw; t += 64) {
                const int r = t / ndw, c = c0 + (t % ndw) * SPD;
                st_dev(ysb + (ptrdiff_t)r * sy + (ptrdiff_t)c * PS, ty32[((r + 8) * T::PY + c + 8) / SPD]);
            }
            if (row)
                for (int t = lane; t < 7 * (wl / SPD); t += 64) {
                    const int r = -7 + t / (wl / SPD), c = (t % (wl / SPD)) * SPD;
                    st_dev(ysb + (ptrdiff_t)r * sy + (ptrdiff_t)c * PS, ty32[((r + 8) * T::PY + c + 8) / SPD]);
                }
            const int cndw = (wc - c0) / SPD;
            for (int t = lane; t < 2 * hc * cndw; t += 64) {
                const int p = t / (hc * cndw), u = t % (hc * cndw), r = u / cndw, c = c0 + (u % cndw) * SPD;
                st_dev(csb[p] + (ptrdiff_t)r * suv + (ptrdiff_t)c * PS, reinterpret_cast<const uint32_t *>(tile.c[p])[((r + 8) * T::PC + c + 8) / SPD]);
            }
            if (row)
                for (int t = lane; t < 2 * 7 * (wc / SPD); t += 64) {
                    const int p = t / (7 * (wc / SPD)), u = t % (7 * (wc / SPD)), r = -7 + u / (wc / SPD), c = (u % (wc / SPD)) * SPD;
                    st_dev(csb[p] + (ptrdiff_t)r * suv + (ptrdiff_t)c * PS, reinterpret_cast<const uint32_t *>(tile.c[p])[((r + 8) * T::PC + c + 8) / SPD]);
                }
        }
        /* acknowledged before the counter moves */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0)
            __hip_atomic_store(&progress[row], col + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_sync(); /* the tile and the tables are rewritten by the next step */
    }
}

int ffhip_launch_vp9_lf_frame(int bd, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                              const FFHipVp9LfSb *tabs, hipStream_t stream)
{
    const int sb_rows = (rows + 7) >> 3;
    if (cols <= 0 || rows <= 0)
        return 0;
    if ((bd != 8 && bd != 10 && bd != 12) || (((uintptr_t)y | (uintptr_t)u | (uintptr_t)v | (size_t)sy | (size_t)suv) & 3)) {
        ffhip_set_error("ffhip_vp9_loopfilter_frame: bit depth %d (8, 10, 12); planes and strides must be 4-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    int *prog, *fail, slot;
    const int r = ffhip_h264_wavefront_slot(sb_rows + 1, &prog, &fail, &slot, stream);
    if (r < 0)
        return r;
    if (bd == 8)
        hipLaunchKernelGGL(k_vp9_lf_frame<uint8_t>, dim3(sb_rows), dim3(64), 0, stream, y, u, v, sy, suv, cols, rows, tabs, prog, fail, 8);
    else
        hipLaunchKernelGGL(k_vp9_lf_frame<uint16_t>, dim3(sb_rows), dim3(64), 0, stream, y, u, v, sy, suv, cols, rows, tabs, prog, fail, bd);
    const hipError_t e = hipGetLastError();
    const int r2 = ffhip_h264_wavefront_slot_done(slot, stream);
    if (e != hipSuccess) {
        ffhip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e), __FILE__, __LINE__);
        return FFHIP_EIO;
    }
    return r2;
}
