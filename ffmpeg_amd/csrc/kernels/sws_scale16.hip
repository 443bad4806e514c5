/*
 * sws_scale16.hip — the legacy scaler above 8 bits: fused horizontal + vertical scaling of 16-bit (and mixed 8 / 16-bit) YUV planes.
 *
 * Reference semantics (libswscale/swscale.c:69-160 hScale16To15_c / hScale16To19_c / hScale8To15_c / hScale8To19_c, output.c:150-200,
 * 330-360 yuv2plane1 / yuv2planeX at the target depth, yuv2nv12cX_16 / yuv2p01x*, input.c p010LEToY_c / p010LEToUV_c, swscale.c:42-52,
 * 291,519-522 the ordered dither of 8-bit targets fed from deeper sources):
 *     Hs[r][x]  = min( (sum_j src[r][hpos[x]+j] * hfilter[x][j]) >> sh, lim )
 *                 sh = 7 (8-bit source) or depth-1 (deeper), lim = 2^15-1;  16-bit targets: sh = 3 / depth-5, lim = 2^19-1
 *     out[y][x] = the target depth's yuv2planeX over Hs[vpos[y] + j][x] (yuv2plane1 when the vertical bank has one tap)
 * with wrap-around int32 accumulation, as the C code has it.
 *
 * One workgroup owns a TW x TH tile of one output plane (or of the interleaved UV plane of a semi-planar target) of one frame:
 *   stage 1  every source row the tile's vertical taps reach is filtered horizontally for the tile's TW columns straight from
 *            global memory (16-bit loads; neighbouring lanes' windows overlap and are served by L1 / L2) into LDS as int32;
 *   stage 2  a lane per output column runs down the tile's rows over the LDS intermediates and stores 16-bit (or 8-bit) samples.
 * The intermediates never touch HBM.  Bit-exact; the 8-bit kernels (sws_up2 / colwalk / ...) remain the 8-bit fast paths.
 */
#include "common.h"
#include "sws_kernels.h"

#define S16_TW 64
#define S16_TH 32
#define S16_MAXROWS 96  /* source rows of one chunk of a tile: (rows - 1) * step + taps; larger reaches run in several chunks */
#define S16_SW 320      /* source columns the staged form holds per row: 64 windows at up to ~4.5 : 1 */

__constant__ uint8_t s16_dither[8][8] = { /* ff_dither_8x8_128, libswscale/swscale.c:42-52 */
    { 36, 68, 60, 92, 34, 66, 58, 90 },  { 100, 4, 124, 28, 98, 2, 122, 26 }, { 52, 84, 44, 76, 50, 82, 42, 74 },
    { 116, 20, 108, 12, 114, 18, 106, 10 }, { 32, 64, 56, 88, 38, 70, 62, 94 }, { 96, 0, 120, 24, 102, 6, 126, 30 },
    { 48, 80, 40, 72, 54, 86, 46, 78 }, { 112, 16, 104, 8, 118, 22, 110, 14 },
};

__device__ __forceinline__ int s16_src(const uint8_t *row, const FFHipScale16Plane &p, int i)
{
    if (p.sdepth == 8)
        return row[(size_t)i * p.sstep + p.schan];
    const int v = reinterpret_cast<const uint16_t *>(row)[(size_t)i * p.sstep + p.schan];
    return p.smsb ? v >> (16 - p.sdepth) : v;
}

__device__ __forceinline__ int s16_clipu(int v, int bits) { return min(max(v, 0), (1 << bits) - 1); }

/* STAGED: the tile's source footprint goes through LDS first (coalesced loads, samples normalised to their values once: the P01x
 * shift, the 8-bit widening, the de-interleave), the horizontal pass then reads LDS; a lane keeps its column's coefficients in
 * registers and runs down the rows.  !STAGED: horizontal windows wider than the LDS tile (steep down-scaling): taps from global memory. */
/* the vertical sum over the bank's taps: unrolled when the tap count is a template argument */
#define S16_VSUM(STMT)                                           \
    do {                                                         \
        if (VFS) {                                               \
            _Pragma("unroll") for (int j = 0; j < VFS; j++) { STMT; } \
        } else {                                                 \
            for (int j = 0; j < vfs; j++) { STMT; }              \
        }                                                        \
    } while (0)

template <bool STAGED, int HFS, int VFS> /* HFS / VFS: the banks' tap counts when they are 4 / 8 (unrolled), 0 = any */
__global__ __launch_bounds__(256) void k_sws_scale16(FFHipScale16Args a)
{
    extern __shared__ int32_t hs[]; /* [max_rows][S16_TW] int32, then (STAGED) [max_rows][S16_SW] uint16 */
    const FFHipScale16Plane p = a.pl[blockIdx.z % a.nplanes];
    const int f = blockIdx.z / a.nplanes;
    const int x0 = blockIdx.x * S16_TW, y0 = blockIdx.y * S16_TH;
    if (x0 >= p.dstW || y0 >= p.dstH)
        return;
    uint16_t *const srcT = reinterpret_cast<uint16_t *>(hs + a.max_rows * S16_TW);
    const int SWP = a.sw_pitch; /* LDS pitch of a staged source row, in samples */
    const int tw = min(S16_TW, p.dstW - x0), th = min(S16_TH, p.dstH - y0);
    const int tid = threadIdx.x, lx = tid & (S16_TW - 1), lg = tid / S16_TW; /* column of the tile, row group 0..3 */
    const uint8_t *src = p.src + (size_t)f * p.src_fp;
    uint8_t *dst = p.dst + (size_t)f * p.dst_fp;
    const int wide = p.ddepth == 16;
    const int sh = p.sdepth == 8 ? (wide ? 3 : 7) : (wide ? p.sdepth - 5 : p.sdepth - 1);
    const int lim = wide ? (1 << 19) - 1 : (1 << 15) - 1;
    const int hfs = HFS ? HFS : p.h.size, vfs = VFS ? VFS : p.v.size;
    const int c0 = p.h.pos[x0], sw = p.h.pos[x0 + tw - 1] + hfs - c0; /* source columns the tile's windows reach (positions ascend) */
    /* this lane's window and coefficients (up to 16 taps in registers; longer banks read theirs from memory) */
    const bool colv = lx < tw;
    const int sp = colv ? p.h.pos[x0 + lx] - c0 : 0;
    const int16_t *hf = p.h.filter + (size_t)(x0 + (colv ? lx : 0)) * hfs;
    constexpr int HC = HFS ? HFS : 16;
    int hc[HC];
#pragma unroll
    for (int j = 0; j < HC; j++)
        hc[j] = j < hfs ? (int)hf[j] : 0;
    /* output rows in chunks whose source reach fits the LDS rows */
    for (int yc = 0; yc < th;) {
        const int r0 = p.v.pos[y0 + yc];
        int yn = yc + 1;
        while (yn < th && p.v.pos[y0 + yn] + vfs - r0 <= a.max_rows)
            yn++;
        const int nrows = p.v.pos[y0 + yn - 1] + vfs - r0;
        if (STAGED) {
            /* stage 0: rows r0 .. r0 + nrows, columns c0 .. c0 + sw as sample values */
            for (int it = tid; it < nrows * sw; it += 256) {
                const int r = it / sw, c = it - r * sw;
                srcT[r * SWP + c] = (uint16_t)s16_src(src + (ptrdiff_t)(r0 + r) * p.src_stride, p, c0 + c);
            }
            __syncthreads();
        }
        /* stage 1: a lane per column, rows lg, lg + 4, ... */
        if (colv)
            for (int r = lg; r < nrows; r += 256 / S16_TW) {
                unsigned acc = 0;
                if (STAGED) {
                    const uint16_t *w = srcT + r * SWP + sp;
                    if (HFS) {
#pragma unroll
                        for (int j = 0; j < HC; j++)
                            acc += (unsigned)((int)w[j] * hc[j]);
                    } else if (hfs <= 16) {
#pragma unroll
                        for (int j = 0; j < HC; j++)
                            if (j < hfs)
                                acc += (unsigned)((int)w[j] * hc[j]);
                    } else
                        for (int j = 0; j < hfs; j++)
                            acc += (unsigned)((int)w[j] * (int)hf[j]);
                } else {
                    const uint8_t *row = src + (ptrdiff_t)(r0 + r) * p.src_stride;
                    for (int j = 0; j < hfs; j++)
                        acc += (unsigned)(s16_src(row, p, c0 + sp + j) * (int)hf[j]);
                }
                int hv = min((int)acc >> sh, lim);
                if (p.rc_coeff) {
                    /* lum / chrRangeTo / FromJpeg[16]_c (swscale.c:160-255): 15-bit intermediates live in int16 line buffers and use
                     * 32-bit products, 19-bit ones in int32 with 64-bit products */
                    if (lim == (1 << 15) - 1) {
                        hv = ((int)(int16_t)hv * (int)(uint16_t)p.rc_coeff + (int)p.rc_offset) >> 14;
                        hv = (int)(int16_t)(p.rc_clip ? min(hv, lim) : hv);
                    } else {
                        hv = (int)(((int64_t)hv * p.rc_coeff + p.rc_offset) >> 18);
                        if (p.rc_clip)
                            hv = min(hv, lim);
                    }
                }
                hs[r * S16_TW + lx] = hv;
            }
        __syncthreads();
        /* stage 2: a lane per column, output rows yc + lg, + 4, ... */
        if (colv)
            for (int yy = yc + lg; yy < yn; yy += 256 / S16_TW) {
                const int y = y0 + yy, x = lx;
                const int32_t *col = hs + (p.v.pos[y] - r0) * S16_TW + x;
                const int16_t *vf = p.v.filter + (size_t)y * vfs;
                int out;
                if (p.ddepth == 8) {
                    const int dz = p.dither ? s16_dither[y & 7][((x0 + x) + p.dither_off) & 7] : 64;
                    if (vfs == 1)
                        out = s16_clipu((col[0] + dz) >> 7, 8);
                    else {
                        unsigned acc = (unsigned)dz << 12;
                        S16_VSUM(acc += (unsigned)(col[j * S16_TW] * (int)vf[j]));
                        out = s16_clipu((int)acc >> 19, 8);
                    }
                    dst[(ptrdiff_t)y * p.dst_stride + (size_t)(x0 + x) * p.dstep + p.dchan] = (uint8_t)out;
                    continue;
                }
                if (wide) {
                    if (vfs == 1)
                        out = s16_clipu((col[0] + 4) >> 3, 16);
                    else {
                        unsigned acc = (1u << 14) - 0x40000000u;
                        S16_VSUM(acc += (unsigned)col[j * S16_TW] * (unsigned)(int)vf[j]);
                        out = 0x8000 + min(max((int)acc >> 15, -32768), 32767);
                    }
                } else if (vfs == 1) {
                    const int shift = 15 - p.ddepth;
                    out = s16_clipu((col[0] + (1 << (shift - 1))) >> shift, p.ddepth);
                } else {
                    const int shift = 27 - p.ddepth;
                    unsigned acc = 1u << (shift - 1);
                    S16_VSUM(acc += (unsigned)(col[j * S16_TW] * (int)vf[j]));
                    out = s16_clipu((int)acc >> shift, p.ddepth);
                }
                if (p.dmsb)
                    out <<= 16 - p.ddepth;
                reinterpret_cast<uint16_t *>(dst + (ptrdiff_t)y * p.dst_stride)[(size_t)(x0 + x) * p.dstep + p.dchan] = (uint16_t)out;
            }
        __syncthreads();
        yc = yn;
    }
}

int ffhip_launch_scale16(const FFHipScale16Args &a0, int nframes, hipStream_t stream)
{
    if (nframes <= 0 || a0.nplanes <= 0)
        return 0;
    FFHipScale16Args a = a0;
    int maxw = 0, maxh = 0, need = 0;
    for (int i = 0; i < a.nplanes; i++) {
        maxw = max(maxw, a.pl[i].dstW);
        maxh = max(maxh, a.pl[i].dstH);
        need = max(need, a.pl[i].v.size);
    }
    if (need > S16_MAXROWS) {
        ffhip_set_error("ffhip_sws: a vertical bank of %d taps exceeds the %d rows a tile holds", need, S16_MAXROWS);
        return FFHIP_EINVAL;
    }
    /* LDS as the context needs it (a.max_rows, a.sw_pitch from the banks): the exact-2x case holds 20 rows of 64 + 36 samples, ~7 KB
     * a workgroup — many workgroups per CU — where the general bound would be 86 KB and one */
    if (a.max_rows < need) a.max_rows = need;
    if (a.max_rows > S16_MAXROWS) a.max_rows = S16_MAXROWS;
    if (a.sw_pitch < 8 || a.sw_pitch > S16_SW) a.sw_pitch = S16_SW;
    const dim3 g(cdiv(maxw, S16_TW), cdiv(maxh, S16_TH), (unsigned)(a.nplanes * nframes));
    const size_t lds_plain = (size_t)a.max_rows * S16_TW * 4, lds_staged = lds_plain + (size_t)a.max_rows * a.sw_pitch * 2;
    /* every plane of a launch shares the kernel: the unrolled forms need all planes' banks at that size */
    int hfs = a.pl[0].h.size, vfs = a.pl[0].v.size;
    for (int i = 1; i < a.nplanes; i++) {
        if (a.pl[i].h.size != hfs) hfs = 0;
        if (a.pl[i].v.size != vfs) vfs = 0;
    }
    const int H = hfs == 4 || hfs == 8 ? hfs : 0, V = vfs == 4 ? 4 : 0;
    const size_t lds = a.staged ? lds_staged : lds_plain;
#define S16_GO(ST, HH, VV)                                                                                                                   \
    do {                                                                                                                                     \
        static FFHipPerDeviceOnce attr;                                                                                                      \
        if (attr.enter()) {                                                                                                                  \
            (void)hipFuncSetAttribute((const void *)k_sws_scale16<ST, HH, VV>, hipFuncAttributeMaxDynamicSharedMemorySize,                   \
                                      S16_MAXROWS * S16_TW * 4 + S16_MAXROWS * S16_SW * 2);                                                  \
            attr.leave(true);                                                                                                                \
        }                                                                                                                                    \
        hipLaunchKernelGGL((k_sws_scale16<ST, HH, VV>), g, dim3(256), lds, stream, a);                                                       \
    } while (0)
    if (!a.staged) S16_GO(false, 0, 0);
    else if (H == 4 && V == 4) S16_GO(true, 4, 4);
    else if (H == 8 && V == 4) S16_GO(true, 8, 4);
    else if (H == 4) S16_GO(true, 4, 0);
    else if (H == 8) S16_GO(true, 8, 0);
    else S16_GO(true, 0, 0);
#undef S16_GO
    LAUNCH_CHECK();
    return 0;
}
