/*
 * sws_scale.hip — fused horizontal + vertical scaling of 8-bit planes, planar / NV12-style output.
 *
 * Reference semantics (what ff_swscale()'s line ring buffers compute, libswscale/swscale.c:263-567,
 * slice.c, hscale.c:39-205, vscale.c:41-107):
 *     Hs[r][x] = min( (sum_j src[r][hpos[x]+j] * hfilter[x][j]) >> 7, 32767 )        int16   (swscale.c:128-142)
 *     out[y][x] = clip_u8( (64<<12 + sum_j Hs[vpos[y]+j][x] * vfilter[y][j]) >> 19 )          (output.c:468-483)
 *     out[y][x] = clip_u8( (Hs[vpos[y]][x] + 64) >> 7 )                 when the vertical bank has 1 tap (:485-493)
 * with the filter banks initFilter() produced (consumed verbatim), U/V de-interleaved on read
 * (nv12ToUV_c, input.c:936) and re-interleaved on write (yuv2nv12cX_c, output.c:495-529).
 *
 * GPU design: one workgroup owns a TW x TH output tile of one channel group of one frame.
 *   stage 1  the tile's source footprint (rows vpos[y0]..vpos[y0+TH-1]+vfs, columns hpos[x0]..) is
 *            loaded once from HBM into LDS with dword / dwordx2 loads (interleaved chroma is split there);
 *   stage 2  every thread owns one output column pair, keeps its 2 x hfs coefficients in registers
 *            and runs down the footprint rows, writing 15-bit intermediates to LDS as packed int16x2;
 *   stage 3  every thread owns 4 adjacent output columns, reads int16x4 from LDS per tap and stores
 *            4 (planar) or 8 (interleaved UV) output bytes with one dword / dwordx2 store.
 * The 15-bit intermediate never touches HBM: traffic is source-in + destination-out, the
 * algorithmic 1.875 B per output pixel of the nv12 1080p->4K case plus tile halos (served by L2/MALL).
 */
#include "common.h"
#include "sws_kernels.h"

#define NT 256

__device__ __forceinline__ int s16lo(uint32_t v) { return (int)(int16_t)(v & 0xFFFF); }
__device__ __forceinline__ int s16hi(uint32_t v) { return (int)v >> 16; }

/* The channel-group source description shared by the planar and the packed-RGB kernels. */
struct SrcGroup {
    const uint8_t *src[2];
    ptrdiff_t stride[2];
    size_t fp[2];
    int step;      /* 1 planar, 2 interleaved pair */
    int srcW;
};

/* stage 1: rows [r0, r0+nrows) x columns [c0a, c1) of C channels -> LDS bytes, row pitch `spitch` */
template <int C>
__device__ __forceinline__ void tile_load(uint8_t *srcT, int spitch, int max_rows, const SrcGroup &g, int f, int r0,
                                          int nrows, int c0a, int c1, bool src_vec)
{
    const int tid = threadIdx.x;
    const int ndw = (c1 - c0a + 3) >> 2; /* dwords per row per channel */
    const int items = nrows * ndw;
    if (g.step == 1) {
        for (int c = 0; c < C; c++) {
            const uint8_t *base = g.src[c] + (size_t)f * g.fp[c];
            for (int it = tid; it < items; it += NT) {
                const int r = it / ndw, d = it - r * ndw;
                const int col = c0a + 4 * d;
                const uint8_t *p = base + (ptrdiff_t)(r0 + r) * g.stride[c] + col;
                uint32_t w;
                if (src_vec && col + 4 <= g.srcW) {
                    w = *reinterpret_cast<const uint32_t *>(p);
                } else {
                    w = 0;
                    for (int b = 0; b < 4; b++)
                        if (col + b < g.srcW)
                            w |= (uint32_t)p[b] << (8 * b);
                }
                *reinterpret_cast<uint32_t *>(srcT + (c * max_rows + r) * spitch + 4 * d) = w;
            }
        }
    } else {
        /* interleaved pair: src[0]/src[1] point at the first byte of their channel */
        const bool swapped = g.src[1] < g.src[0];
        const uint8_t *base = (swapped ? g.src[1] : g.src[0]) + (size_t)f * g.fp[0];
        for (int it = tid; it < items; it += NT) {
            const int r = it / ndw, d = it - r * ndw;
            const int col = c0a + 4 * d;
            const uint8_t *p = base + (ptrdiff_t)(r0 + r) * g.stride[0] + 2 * col;
            uint32_t lo, hi;
            if (src_vec && col + 4 <= g.srcW) {
                const uint2 w = *reinterpret_cast<const uint2 *>(p);
                lo = w.x;
                hi = w.y;
            } else {
                lo = hi = 0;
                for (int b = 0; b < 8; b++)
                    if (col + (b >> 1) < g.srcW) {
                        if (b < 4) lo |= (uint32_t)p[b] << (8 * b);
                        else       hi |= (uint32_t)p[b] << (8 * (b - 4));
                    }
            }
            /* bytes 0,2 of lo and 0,2 of hi -> first channel; 1,3 / 1,3 -> second */
            uint32_t e = __builtin_amdgcn_perm(hi, lo, 0x06040200);
            uint32_t o = __builtin_amdgcn_perm(hi, lo, 0x07050301);
            if (swapped) { uint32_t t = e; e = o; o = t; }
            *reinterpret_cast<uint32_t *>(srcT + (0 * max_rows + r) * spitch + 4 * d) = e;
            if (C == 2)
                *reinterpret_cast<uint32_t *>(srcT + (1 * max_rows + r) * spitch + 4 * d) = o;
        }
    }
}

/* stage 2: horizontal pass over the LDS footprint; thread = one output column pair, all rows.
 * `tw_full` is the nominal tile width (thread mapping), `tw` the valid width of this tile. */
/* the reference's range conversion on a 15-bit horizontal sum, int16 in and out as its line buffers are (swscale.c:160-207) */
__device__ __forceinline__ int range15(int v, int coeff, int offset, int clip)
{
    if (!coeff)
        return v;
    v = ((int)(int16_t)v * coeff + offset) >> 14;
    return clip ? min(v, 32767) : v;
}

template <int C, int HFS>
__device__ __forceinline__ void tile_hscale(const uint8_t *srcT, int spitch, int16_t *hs, int hpitch, int max_rows,
                                            int nrows, const FFHipDevFilter &h, int x0, int tw_full, int tw, int c0a,
                                            int rc_coeff = 0, int rc_offset = 0, int rc_clip = 0)
{
    const int tid = threadIdx.x;
    const int hfs = HFS ? HFS : h.size;
    const int halfw = tw_full >> 1; /* threads per intermediate row */
    const int cp = tid % halfw, rsub = tid / halfw, nsub = NT / halfw;
    const int xa = 2 * cp, xb = 2 * cp + 1;
    if (xa >= tw)
        return;
    const bool has_b = xb < tw;
    const int pa = h.pos[x0 + xa] - c0a;
    const int pb = has_b ? h.pos[x0 + xb] - c0a : pa;
    const int16_t *fa = h.filter + (size_t)(x0 + xa) * hfs;
    const int16_t *fb = h.filter + (size_t)(x0 + (has_b ? xb : xa)) * hfs;
    if (HFS == 4) {
        int ka[4], kb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            ka[j] = fa[j];
            kb[j] = fb[j];
        }
        const int da = pa & ~3, sa = pa & 3, db = pb & ~3, sb = pb & 3;
        for (int c = 0; c < C; c++)
            for (int r = rsub; r < nrows; r += nsub) {
                const uint8_t *row = srcT + (c * max_rows + r) * spitch;
                const uint32_t a0 = *reinterpret_cast<const uint32_t *>(row + da);
                const uint32_t a1 = *reinterpret_cast<const uint32_t *>(row + da + 4);
                const uint32_t b0 = *reinterpret_cast<const uint32_t *>(row + db);
                const uint32_t b1 = *reinterpret_cast<const uint32_t *>(row + db + 4);
                const uint32_t wa = __builtin_amdgcn_alignbyte(a1, a0, sa);
                const uint32_t wb = __builtin_amdgcn_alignbyte(b1, b0, sb);
                int va = (int)(wa & 0xFF) * ka[0] + (int)((wa >> 8) & 0xFF) * ka[1] +
                         (int)((wa >> 16) & 0xFF) * ka[2] + (int)(wa >> 24) * ka[3];
                int vb = (int)(wb & 0xFF) * kb[0] + (int)((wb >> 8) & 0xFF) * kb[1] +
                         (int)((wb >> 16) & 0xFF) * kb[2] + (int)(wb >> 24) * kb[3];
                va = range15(min(va >> 7, 32767), rc_coeff, rc_offset, rc_clip);
                vb = range15(min(vb >> 7, 32767), rc_coeff, rc_offset, rc_clip);
                *reinterpret_cast<uint32_t *>(hs + (c * max_rows + r) * hpitch + xa) =
                    ((uint32_t)va & 0xFFFF) | ((uint32_t)vb << 16);
            }
    } else {
        for (int c = 0; c < C; c++)
            for (int r = rsub; r < nrows; r += nsub) {
                const uint8_t *row = srcT + (c * max_rows + r) * spitch;
                int va = 0, vb = 0;
                for (int j = 0; j < hfs; j++) {
                    va += (int)row[pa + j] * fa[j];
                    vb += (int)row[pb + j] * fb[j];
                }
                va = range15(min(va >> 7, 32767), rc_coeff, rc_offset, rc_clip);
                vb = range15(min(vb >> 7, 32767), rc_coeff, rc_offset, rc_clip);
                *reinterpret_cast<uint32_t *>(hs + (c * max_rows + r) * hpitch + xa) =
                    ((uint32_t)va & 0xFFFF) | ((uint32_t)vb << 16);
            }
    }
}

template <int C, int HFS, int VFS>
__device__ __forceinline__ void scale_plane_body(const FFHipScalePlaneArgs &a, uint8_t *lds, int bx, int by, int f,
                                                 int spitch, int hs_off, int flags)
{
    const int x0 = bx * a.tw, y0 = by * a.th;
    const int tw = min(a.tw, a.dstW - x0), th = min(a.th, a.dstH - y0);
    const int hfs = HFS ? HFS : a.h.size, vfs = VFS ? VFS : a.v.size;
    const int tid = threadIdx.x;

    const int c0 = a.h.pos[x0], c1 = min(a.h.pos[x0 + tw - 1] + hfs, a.srcW);
    const int r0 = a.v.pos[y0], r1 = min(a.v.pos[y0 + th - 1] + vfs, a.srcH);
    const int c0a = c0 & ~3;
    const int nrows = r1 - r0;
    const int hpitch = a.tw; /* int16 elements per intermediate row */
    uint8_t *srcT = lds;
    int16_t *hs = reinterpret_cast<int16_t *>(lds + hs_off);
    const bool src_vec = flags & 1, dst_vec = flags & 2;

    SrcGroup g;
    g.src[0] = a.src[0]; g.src[1] = a.src[1];
    g.stride[0] = a.src_stride[0]; g.stride[1] = a.src_stride[1];
    g.fp[0] = a.src_fp[0]; g.fp[1] = a.src_fp[1];
    g.step = a.src_step; g.srcW = a.srcW;
    tile_load<C>(srcT, spitch, a.max_rows, g, f, r0, nrows, c0a, c1, src_vec);
    __syncthreads();
    tile_hscale<C, HFS>(srcT, spitch, hs, hpitch, a.max_rows, nrows, a.h, x0, a.tw, tw, c0a, a.rc_coeff, a.rc_offset, a.rc_clip);
    __syncthreads();

    /* ---------------- stage 3: vertical pass, LDS int16 -> HBM u8 ---------------- */
    {
        const int quarterw = a.tw >> 2;
        const int q = tid % quarterw, rsub = tid / quarterw, nsub = NT / quarterw;
        const int xq = 4 * q;
        if (xq < tw) {
            const int nvalid = min(4, tw - xq);
            for (int y = rsub; y < th; y += nsub) {
                const int vr = a.v.pos[y0 + y] - r0;
                const int16_t *vf = a.v.filter + (size_t)(y0 + y) * vfs;
                int out[C][4];
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const int16_t *col = hs + (c * a.max_rows + vr) * hpitch + xq;
                    if (vfs == 1) {
                        const uint2 w = *reinterpret_cast<const uint2 *>(col);
                        out[c][0] = clip_u8((s16lo(w.x) + 64) >> 7);
                        out[c][1] = clip_u8((s16hi(w.x) + 64) >> 7);
                        out[c][2] = clip_u8((s16lo(w.y) + 64) >> 7);
                        out[c][3] = clip_u8((s16hi(w.y) + 64) >> 7);
                    } else {
                        uint32_t acc[4] = { 64u << 12, 64u << 12, 64u << 12, 64u << 12 };
                        if (VFS == 4) {
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const uint2 w = *reinterpret_cast<const uint2 *>(col + j * hpitch);
                                const int k = vf[j];
                                acc[0] += (uint32_t)(s16lo(w.x) * k);
                                acc[1] += (uint32_t)(s16hi(w.x) * k);
                                acc[2] += (uint32_t)(s16lo(w.y) * k);
                                acc[3] += (uint32_t)(s16hi(w.y) * k);
                            }
                        } else {
                            for (int j = 0; j < vfs; j++) {
                                const uint2 w = *reinterpret_cast<const uint2 *>(col + j * hpitch);
                                const int k = vf[j];
                                acc[0] += (uint32_t)(s16lo(w.x) * k);
                                acc[1] += (uint32_t)(s16hi(w.x) * k);
                                acc[2] += (uint32_t)(s16lo(w.y) * k);
                                acc[3] += (uint32_t)(s16hi(w.y) * k);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            out[c][i] = clip_u8((int32_t)acc[i] >> 19);
                    }
                }
                if (a.dst_step == 1) {
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        uint8_t *d = a.dst[c] + (size_t)f * a.dst_fp[c] + (ptrdiff_t)(y0 + y) * a.dst_stride[c] + x0 + xq;
                        if (dst_vec && nvalid == 4) {
                            *reinterpret_cast<uint32_t *>(d) = pack4(out[c][0], out[c][1], out[c][2], out[c][3]);
                        } else {
                            for (int i = 0; i < nvalid; i++)
                                d[i] = (uint8_t)out[c][i];
                        }
                    }
                } else {
                    const bool swapped = a.dst[1] < a.dst[0];
                    uint8_t *d = (swapped ? a.dst[1] : a.dst[0]) + (size_t)f * a.dst_fp[0] +
                                 (ptrdiff_t)(y0 + y) * a.dst_stride[0] + 2 * (x0 + xq);
                    const int e = swapped ? 1 : 0, o = swapped ? 0 : 1;
                    if (dst_vec && nvalid == 4) {
                        uint2 w;
                        w.x = pack4(out[e][0], out[o % C][0], out[e][1], out[o % C][1]);
                        w.y = pack4(out[e][2], out[o % C][2], out[e][3], out[o % C][3]);
                        *reinterpret_cast<uint2 *>(d) = w;
                    } else {
                        for (int i = 0; i < nvalid; i++) {
                            d[2 * i] = (uint8_t)out[e][i];
                            d[2 * i + 1] = (uint8_t)out[o % C][i];
                        }
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
static size_t plane_lds(const FFHipScalePlaneArgs &a, int C, int *spitch, int *hs_off)
{
    *spitch = ((a.max_cols + 3 + 3) & ~3) + 8;
    size_t s = (size_t)C * a.max_rows * *spitch;
    s = (s + 15) & ~(size_t)15;
    *hs_off = (int)s;
    return s + (size_t)C * a.max_rows * a.tw * sizeof(int16_t) + 16;
}

int ffhip_plan_scale_plane(FFHipScalePlaneArgs *a, int C, const int32_t *hpos, const int32_t *vpos)
{
    static const int tws[] = { 256, 128, 64 };
    static const int ths[] = { 32, 16, 8, 4, 2, 1 };
    for (unsigned i = 0; i < sizeof(tws) / sizeof(*tws); i++)
        for (unsigned j = 0; j < sizeof(ths) / sizeof(*ths); j++) {
            const int tw = tws[i], th = ths[j];
            int mc = 0, mr = 0, sp, ho;
            if (tw > 64 && tw >= 2 * a->dstW && i + 1 < sizeof(tws) / sizeof(*tws))
                break; /* narrower tile fits the whole row */
            for (int x0 = 0; x0 < a->dstW; x0 += tw) {
                int xe = x0 + tw < a->dstW ? x0 + tw : a->dstW;
                int n = hpos[xe - 1] + a->h.size - (hpos[x0] & ~3);
                if (n > mc) mc = n;
            }
            for (int y0 = 0; y0 < a->dstH; y0 += th) {
                int ye = y0 + th < a->dstH ? y0 + th : a->dstH;
                int n = vpos[ye - 1] + a->v.size - vpos[y0];
                if (n > mr) mr = n;
            }
            a->tw = tw; a->th = th; a->max_cols = mc; a->max_rows = mr;
            if (plane_lds(*a, C, &sp, &ho) <= 64 * 1024) {
                a->tiles_x = cdiv(a->dstW, tw);
                a->tiles_y = cdiv(a->dstH, th);
                return 0;
            }
        }
    ffhip_set_error("ffhip_sws: filter footprint does not fit LDS (hfs %d vfs %d)", a->h.size, a->v.size);
    return FFHIP_EINVAL;
}

/*
 * One launch scales a whole batch: blockIdx.y first walks the luma tile rows, then the chroma tile
 * rows (U and V together); blockIdx.z is the frame.  The dominant kernel of the nv12 1080p->4K case.
 */
struct GroupLaunch { int spitch, hs_off, flags; };

template <int HFS, int VFS>
__global__ __launch_bounds__(NT) void k_sws_scale_yuv(FFHipScalePlaneArgs lum, FFHipScalePlaneArgs chr, GroupLaunch gl,
                                                      GroupLaunch gc)
{
    extern __shared__ __align__(16) uint8_t lds[];
    const int f = blockIdx.z;
    if ((int)blockIdx.y < lum.tiles_y) {
        if ((int)blockIdx.x < lum.tiles_x)
            scale_plane_body<1, HFS, VFS>(lum, lds, blockIdx.x, blockIdx.y, f, gl.spitch, gl.hs_off, gl.flags);
    } else {
        if ((int)blockIdx.x < chr.tiles_x)
            scale_plane_body<2, HFS, VFS>(chr, lds, blockIdx.x, blockIdx.y - lum.tiles_y, f, gc.spitch, gc.hs_off,
                                          gc.flags);
    }
}

static int group_flags(const FFHipScalePlaneArgs &a, int C)
{
    int flags = 0;
    /* vector paths need naturally aligned rows */
    if (a.src_step == 1) {
        bool ok = true;
        for (int c = 0; c < C; c++)
            ok = ok && !(((uintptr_t)a.src[c] | (size_t)a.src_stride[c] | a.src_fp[c]) & 3);
        flags |= ok ? 1 : 0;
    } else {
        const uint8_t *b = a.src[1] < a.src[0] ? a.src[1] : a.src[0];
        flags |= !(((uintptr_t)b | (size_t)a.src_stride[0] | a.src_fp[0]) & 7) ? 1 : 0;
    }
    if (a.dst_step == 1) {
        bool ok = true;
        for (int c = 0; c < C; c++)
            ok = ok && !(((uintptr_t)a.dst[c] | (size_t)a.dst_stride[c] | a.dst_fp[c]) & 3);
        flags |= ok ? 2 : 0;
    } else {
        const uint8_t *b = a.dst[1] < a.dst[0] ? a.dst[1] : a.dst[0];
        flags |= !(((uintptr_t)b | (size_t)a.dst_stride[0] | a.dst_fp[0]) & 7) ? 2 : 0;
    }
    return flags;
}

int ffhip_launch_scale_yuv(const FFHipScalePlaneArgs &lum, const FFHipScalePlaneArgs &chr, hipStream_t stream)
{
    if (lum.nframes <= 0)
        return 0;
    GroupLaunch gl, gc;
    const size_t lds_l = plane_lds(lum, 1, &gl.spitch, &gl.hs_off);
    const size_t lds_c = plane_lds(chr, 2, &gc.spitch, &gc.hs_off);
    gl.flags = group_flags(lum, 1);
    gc.flags = group_flags(chr, 2);
    const size_t lds = lds_l > lds_c ? lds_l : lds_c;
    const dim3 grid(lum.tiles_x > chr.tiles_x ? lum.tiles_x : chr.tiles_x, lum.tiles_y + chr.tiles_y, lum.nframes);
    const dim3 block(NT);
    const bool h4 = lum.h.size == 4 && chr.h.size == 4, v4 = lum.v.size == 4 && chr.v.size == 4;
    if (h4 && v4)
        hipLaunchKernelGGL((k_sws_scale_yuv<4, 4>), grid, block, lds, stream, lum, chr, gl, gc);
    else if (h4)
        hipLaunchKernelGGL((k_sws_scale_yuv<4, 0>), grid, block, lds, stream, lum, chr, gl, gc);
    else
        hipLaunchKernelGGL((k_sws_scale_yuv<0, 0>), grid, block, lds, stream, lum, chr, gl, gc);
    LAUNCH_CHECK();
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/*
 * Packed RGB24/BGR24 output of a scaled (or ACCURATE_RND "unscaled") conversion:
 * packed_vscale()'s per-line dispatch (libswscale/vscale.c:109-171) over yuv2rgb24_{1,2,X}_c
 * (libswscale/output.c:1789-1939) with yuv2rgb_write (:1697-1714) and the LUTs in closed form
 * (see sws_yuv2rgb.hip).  Chroma is kept at half horizontal resolution (chrDstHSubSample == 1,
 * utils.c:1359-1360) and full vertical resolution; U,V are clamped to [0,255] exactly as the
 * tables' headroom does (fill_table, yuv2rgb.c:688-690), Y is not.
 */
template <int HFS, bool FULL = false>
__global__ __launch_bounds__(NT) void k_scale_rgb(FFHipScaleRgbArgs a, int spitch_l, int spitch_c, int off_sc, int off_hl,
                                                  int off_hc, int flags, int off_sa = 0, int off_ha = 0)
{
    extern __shared__ __align__(16) uint8_t lds[];
    const int x0 = blockIdx.x * a.tw, y0 = blockIdx.y * a.th, f = blockIdx.z;
    const int tw = min(a.tw, a.dstW - x0), th = min(a.th, a.dstH - y0);
    /* FULL (SWS_FULL_CHR_H_INT): a chroma sample per pixel */
    const int twc_full = FULL ? a.tw : a.tw >> 1, xc0 = FULL ? x0 : x0 >> 1, twc = FULL ? tw : (tw + 1) >> 1;
    const int tid = threadIdx.x;
    const int lfs = a.vl.size, cfs = a.vc.size;

    const int c0a_l = a.hl.pos[x0] & ~3, c1_l = min(a.hl.pos[x0 + tw - 1] + a.hl.size, a.srcW);
    const int c0a_c = a.hc.pos[xc0] & ~3, c1_c = min(a.hc.pos[xc0 + twc - 1] + a.hc.size, a.chrSrcW);
    const int r0_l = a.vl.pos[y0], r1_l = min(a.vl.pos[y0 + th - 1] + lfs, a.srcH);
    const int r0_c = a.vc.pos[y0], r1_c = min(a.vc.pos[y0 + th - 1] + cfs, a.chrSrcH);
    uint8_t *src_l = lds, *src_c = lds + off_sc;
    int16_t *hs_l = reinterpret_cast<int16_t *>(lds + off_hl), *hs_c = reinterpret_cast<int16_t *>(lds + off_hc);

    SrcGroup gl, gc;
    gl.src[0] = a.src[0]; gl.src[1] = a.src[0]; gl.stride[0] = gl.stride[1] = a.src_stride[0];
    gl.fp[0] = gl.fp[1] = a.src_fp[0]; gl.step = 1; gl.srcW = a.srcW;
    gc.src[0] = a.src[1]; gc.src[1] = a.src[2]; gc.stride[0] = a.src_stride[1]; gc.stride[1] = a.src_stride[2];
    gc.fp[0] = a.src_fp[1]; gc.fp[1] = a.src_fp[2]; gc.step = a.chr_step; gc.srcW = a.chrSrcW;
    tile_load<1>(src_l, spitch_l, a.max_rows_l, gl, f, r0_l, r1_l - r0_l, c0a_l, c1_l, flags & 1);
    tile_load<2>(src_c, spitch_c, a.max_rows_c, gc, f, r0_c, r1_c - r0_c, c0a_c, c1_c, flags & 2);
    /* the alpha plane: a second plane of the luma's geometry through the luma banks */
    uint8_t *src_a = lds + off_sa;
    int16_t *hs_a = reinterpret_cast<int16_t *>(lds + off_ha);
    if (a.alpha) {
        SrcGroup ga;
        ga.src[0] = ga.src[1] = a.alpha; ga.stride[0] = ga.stride[1] = a.alpha_stride;
        ga.fp[0] = ga.fp[1] = a.alpha_fp; ga.step = 1; ga.srcW = a.srcW;
        tile_load<1>(src_a, spitch_l, a.max_rows_l, ga, f, r0_l, r1_l - r0_l, c0a_l, c1_l, flags & 8);
    }
    __syncthreads();
    tile_hscale<1, HFS>(src_l, spitch_l, hs_l, a.tw, a.max_rows_l, r1_l - r0_l, a.hl, x0, a.tw, tw, c0a_l);
    tile_hscale<2, HFS>(src_c, spitch_c, hs_c, twc_full, a.max_rows_c, r1_c - r0_c, a.hc, xc0, twc_full, twc, c0a_c);
    if (a.alpha)
        tile_hscale<1, HFS>(src_a, spitch_l, hs_a, a.tw, a.max_rows_l, r1_l - r0_l, a.hl, x0, a.tw, tw, c0a_l);
    __syncthreads();

    const int quarterw = a.tw >> 2;
    const int q = tid % quarterw, rsub = tid / quarterw, nsub = NT / quarterw;
    const int xq = 4 * q;
    if (xq >= tw)
        return;
    const int npx = FULL ? min(4, tw - xq) : min(4, tw - xq) & ~1; /* without full chroma RGB widths are even */
    const FFHipYuv2RgbK k = a.k;
    if (FULL) {
        /* yuv2rgb_full_{1,2,X}_c_template + yuv2rgb_write_full (libswscale/output.c:1998-2310): packed_vscale()'s dispatch as below,
         * Y / U / V in the writers' scale (the vertical sum >> 10), R = Y' + V v2r ... in 32-bit wrapping arithmetic, clipped to 30
         * bits, >> 22 */
        const int lay = a.bgr, bp = lay < 2 ? 3 : 4;
        for (int y = rsub; y < th; y += nsub) {
            const int lr = a.vl.pos[y0 + y] - r0_l, cr = a.vc.pos[y0 + y] - r0_c;
            const uint16_t *lf = reinterpret_cast<const uint16_t *>(a.vl.filter) + (size_t)(y0 + y) * lfs;
            const uint16_t *cf = reinterpret_cast<const uint16_t *>(a.vc.filter) + (size_t)(y0 + y) * cfs;
            const int16_t *lcol = hs_l + lr * a.tw + xq;
            const int16_t *ucol = hs_c + (0 * a.max_rows_c + cr) * twc_full + xq;
            const int16_t *vcol = hs_c + (1 * a.max_rows_c + cr) * twc_full + xq;
            const bool chr_bilin = cfs == 2 && (int)cf[0] + (int)cf[1] == 4096 && cf[1] <= 4096u;
            const bool lum_bilin = lfs == 2 && (int)lf[0] + (int)lf[1] == 4096 && lf[1] <= 4096u;
            const int16_t *acol = hs_a + lr * a.tw + xq;
            uint8_t px[16];
            for (int i = 0; i < npx; i++) {
                int Y, U, V, A = 255;
                if (a.alpha) {
                    /* yuv2rgb_full_{1,2,X}_c_template's alpha (output.c:2193-2200, 2241-2245, 2278-2282, 2298-2302): the luma's case */
                    if (lfs == 1 && (cfs == 1 || chr_bilin)) {
                        A = (acol[i] + 64) >> 7;
                    } else if (lum_bilin && chr_bilin) {
                        A = (acol[i] * (4096 - (int)lf[1]) + acol[i + a.tw] * (int)lf[1] + (1 << 18)) >> 19;
                    } else {
                        uint32_t aa = 1u << 18;
                        for (int j = 0; j < lfs; j++)
                            aa += (uint32_t)((int)acol[i + j * a.tw] * (int)(int16_t)lf[j]);
                        A = (int32_t)aa >> 19;
                    }
                    if (A & 0x100)
                        A = clip_u8(A);
                    A &= 0xFF;
                }
                if (lfs == 1 && (cfs == 1 || chr_bilin)) {
                    Y = lcol[i] * 4;
                    if (cfs == 1 || cf[1] == 0) {
                        U = (ucol[i] - (128 << 7)) * 4;
                        V = (vcol[i] - (128 << 7)) * 4;
                    } else {
                        const int al = cf[1], al1 = 4096 - al;
                        U = (ucol[i] * al1 + ucol[i + twc_full] * al - (128 << 19)) >> 10;
                        V = (vcol[i] * al1 + vcol[i + twc_full] * al - (128 << 19)) >> 10;
                    }
                } else if (lum_bilin && chr_bilin) {
                    const int ya = lf[1], ya1 = 4096 - ya, ua = cf[1], ua1 = 4096 - ua;
                    Y = (lcol[i] * ya1 + lcol[i + a.tw] * ya) >> 10;
                    U = (ucol[i] * ua1 + ucol[i + twc_full] * ua - (128 << 19)) >> 10;
                    V = (vcol[i] * ua1 + vcol[i + twc_full] * ua - (128 << 19)) >> 10;
                } else {
                    uint32_t ay = 1u << 9, au = (1u << 9) - (128u << 19), av = (1u << 9) - (128u << 19);
                    for (int j = 0; j < lfs; j++)
                        ay += (uint32_t)((int)lcol[i + j * a.tw] * (int)(int16_t)lf[j]);
                    for (int j = 0; j < cfs; j++) {
                        const int c = (int16_t)cf[j];
                        au += (uint32_t)((int)ucol[i + j * twc_full] * c);
                        av += (uint32_t)((int)vcol[i + j * twc_full] * c);
                    }
                    Y = (int32_t)ay >> 10; U = (int32_t)au >> 10; V = (int32_t)av >> 10;
                }
                const uint32_t yy = (uint32_t)(Y - a.fk[1]) * (uint32_t)a.fk[0] + (1u << 21);
                int R = (int)(yy + (uint32_t)V * (uint32_t)a.fk[2]);
                int G = (int)(yy + (uint32_t)V * (uint32_t)a.fk[3] + (uint32_t)U * (uint32_t)a.fk[4]);
                int B = (int)(yy + (uint32_t)U * (uint32_t)a.fk[5]);
                R = min(max(R, 0), (1 << 30) - 1);      /* av_clip_uintp2(., 30), a no-op when the top two bits are clear */
                G = min(max(G, 0), (1 << 30) - 1);
                B = min(max(B, 0), (1 << 30) - 1);
                const uint8_t r = (uint8_t)(R >> 22), g = (uint8_t)(G >> 22), b = (uint8_t)(B >> 22);
                uint8_t *q = px + bp * i;
                switch (lay) {
                case 0: q[0] = r; q[1] = g; q[2] = b; break;
                case 1: q[0] = b; q[1] = g; q[2] = r; break;
                case 2: q[0] = (uint8_t)A; q[1] = r; q[2] = g; q[3] = b; break;
                case 3: q[0] = r; q[1] = g; q[2] = b; q[3] = (uint8_t)A; break;
                case 4: q[0] = (uint8_t)A; q[1] = b; q[2] = g; q[3] = r; break;
                default: q[0] = b; q[1] = g; q[2] = r; q[3] = (uint8_t)A; break;
                }
            }
            uint8_t *d = a.dst + (size_t)f * a.dst_fp + (ptrdiff_t)(y0 + y) * a.dst_stride + bp * (x0 + xq);
            for (int i = 0; i < bp * npx; i++)
                d[i] = px[i];
        }
        return;
    }
    for (int y = rsub; y < th; y += nsub) {
        const int lr = a.vl.pos[y0 + y] - r0_l, cr = a.vc.pos[y0 + y] - r0_c;
        const uint16_t *lf = reinterpret_cast<const uint16_t *>(a.vl.filter) + (size_t)(y0 + y) * lfs;
        const uint16_t *cf = reinterpret_cast<const uint16_t *>(a.vc.filter) + (size_t)(y0 + y) * cfs;
        const int16_t *lcol = hs_l + lr * a.tw + xq;
        const int16_t *ucol = hs_c + (0 * a.max_rows_c + cr) * twc_full + (xq >> 1);
        const int16_t *vcol = hs_c + (1 * a.max_rows_c + cr) * twc_full + (xq >> 1);
        int Y[4], U[2], V[2];
        const bool chr_bilin = cfs == 2 && (int)cf[0] + (int)cf[1] == 4096 && cf[1] <= 4096u;
        const bool lum_bilin = lfs == 2 && (int)lf[0] + (int)lf[1] == 4096 && lf[1] <= 4096u;
        if (lfs == 1 && (cfs == 1 || chr_bilin)) {
            /* yuv2rgb_1_c_template */
            const uint2 w = *reinterpret_cast<const uint2 *>(lcol);
            Y[0] = (s16lo(w.x) + 64) >> 7; Y[1] = (s16hi(w.x) + 64) >> 7;
            Y[2] = (s16lo(w.y) + 64) >> 7; Y[3] = (s16hi(w.y) + 64) >> 7;
            const uint32_t u0 = *reinterpret_cast<const uint32_t *>(ucol), v0 = *reinterpret_cast<const uint32_t *>(vcol);
            if (cfs == 1 || cf[1] == 0) {
                U[0] = (s16lo(u0) + 64) >> 7; U[1] = (s16hi(u0) + 64) >> 7;
                V[0] = (s16lo(v0) + 64) >> 7; V[1] = (s16hi(v0) + 64) >> 7;
            } else {
                const int al = cf[1], al1 = 4096 - al;
                const uint32_t u1 = *reinterpret_cast<const uint32_t *>(ucol + twc_full);
                const uint32_t v1 = *reinterpret_cast<const uint32_t *>(vcol + twc_full);
                U[0] = (s16lo(u0) * al1 + s16lo(u1) * al + (128 << 11)) >> 19;
                U[1] = (s16hi(u0) * al1 + s16hi(u1) * al + (128 << 11)) >> 19;
                V[0] = (s16lo(v0) * al1 + s16lo(v1) * al + (128 << 11)) >> 19;
                V[1] = (s16hi(v0) * al1 + s16hi(v1) * al + (128 << 11)) >> 19;
            }
        } else if (lum_bilin && chr_bilin) {
            /* yuv2rgb_2_c_template */
            const int ya = lf[1], ya1 = 4096 - ya, ua = cf[1], ua1 = 4096 - ua;
            const uint2 w0 = *reinterpret_cast<const uint2 *>(lcol), w1 = *reinterpret_cast<const uint2 *>(lcol + a.tw);
            Y[0] = (s16lo(w0.x) * ya1 + s16lo(w1.x) * ya) >> 19; Y[1] = (s16hi(w0.x) * ya1 + s16hi(w1.x) * ya) >> 19;
            Y[2] = (s16lo(w0.y) * ya1 + s16lo(w1.y) * ya) >> 19; Y[3] = (s16hi(w0.y) * ya1 + s16hi(w1.y) * ya) >> 19;
            const uint32_t u0 = *reinterpret_cast<const uint32_t *>(ucol), u1 = *reinterpret_cast<const uint32_t *>(ucol + twc_full);
            const uint32_t v0 = *reinterpret_cast<const uint32_t *>(vcol), v1 = *reinterpret_cast<const uint32_t *>(vcol + twc_full);
            U[0] = (s16lo(u0) * ua1 + s16lo(u1) * ua) >> 19; U[1] = (s16hi(u0) * ua1 + s16hi(u1) * ua) >> 19;
            V[0] = (s16lo(v0) * ua1 + s16lo(v1) * ua) >> 19; V[1] = (s16hi(v0) * ua1 + s16hi(v1) * ua) >> 19;
        } else {
            /* yuv2rgb_X_c_template */
            uint32_t ay[4] = { 1u << 18, 1u << 18, 1u << 18, 1u << 18 }, au[2] = { 1u << 18, 1u << 18 },
                     av[2] = { 1u << 18, 1u << 18 };
            for (int j = 0; j < lfs; j++) {
                const uint2 w = *reinterpret_cast<const uint2 *>(lcol + j * a.tw);
                const int c = (int16_t)lf[j];
                ay[0] += (uint32_t)(s16lo(w.x) * c); ay[1] += (uint32_t)(s16hi(w.x) * c);
                ay[2] += (uint32_t)(s16lo(w.y) * c); ay[3] += (uint32_t)(s16hi(w.y) * c);
            }
            for (int j = 0; j < cfs; j++) {
                const uint32_t uw = *reinterpret_cast<const uint32_t *>(ucol + j * twc_full);
                const uint32_t vw = *reinterpret_cast<const uint32_t *>(vcol + j * twc_full);
                const int c = (int16_t)cf[j];
                au[0] += (uint32_t)(s16lo(uw) * c); au[1] += (uint32_t)(s16hi(uw) * c);
                av[0] += (uint32_t)(s16lo(vw) * c); av[1] += (uint32_t)(s16hi(vw) * c);
            }
            for (int i = 0; i < 4; i++) Y[i] = (int32_t)ay[i] >> 19;
            for (int i = 0; i < 2; i++) { U[i] = (int32_t)au[i] >> 19; V[i] = (int32_t)av[i] >> 19; }
        }
        /* the alpha byte: 255, or the source's alpha plane through the writer's own case (yuv2rgb_{1,2,X}_c_template with hasAlpha,
         * output.c:1823-1835, 1875-1880, 1911-1916, 1939-1944: three different roundings, and X clips a PAIR when either value needs it) */
        int A[4] = { 255, 255, 255, 255 };
        if (a.alpha) {
            const int16_t *acol = hs_a + lr * a.tw + xq;
            if (lfs == 1 && (cfs == 1 || chr_bilin)) {
                const bool uv0 = cfs == 1 || cf[1] == 0;
                for (int i = 0; i < 4; i++)
                    A[i] = clip_u8(uv0 ? (acol[i] * 255 + 16384) >> 15 : (acol[i] + 64) >> 7);
            } else if (lum_bilin && chr_bilin) {
                const int ya = lf[1], ya1 = 4096 - ya;
                for (int i = 0; i < 4; i++)
                    A[i] = clip_u8((acol[i] * ya1 + acol[i + a.tw] * ya) >> 19);
            } else {
                for (int i = 0; i < 4; i++) {
                    uint32_t aa = 1u << 18;
                    for (int j = 0; j < lfs; j++)
                        aa += (uint32_t)((int)acol[i + j * a.tw] * (int)(int16_t)lf[j]);
                    A[i] = (int32_t)aa >> 19;
                }
                for (int m = 0; m < 2; m++)
                    if ((A[2 * m] | A[2 * m + 1]) & 0x100) {
                        A[2 * m] = clip_u8(A[2 * m]);
                        A[2 * m + 1] = clip_u8(A[2 * m + 1]);
                    }
            }
        }
        uint8_t px[16];
        const int lay = a.bgr; /* 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra */
        const int bp = lay < 2 ? 3 : 4;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int Uc = clip_u8(U[m]), Vc = clip_u8(V[m]);
            const int br = k.kb + (k.off_r + ((Vc * k.crv) >> 16)) * k.cy;
            const int bb = k.kb + (k.off_b + ((Uc * k.cbu) >> 16)) * k.cy;
            const int bg = k.kb + (k.off_g + ((Uc * k.cgu) >> 16) + ((Vc * k.cgv) >> 16)) * k.cy;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int yc = Y[2 * m + e] * k.cy;
                const int r = clip_u8((br + yc) >> 16), g = clip_u8((bg + yc) >> 16), b = clip_u8((bb + yc) >> 16);
                uint8_t *q = px + bp * (2 * m + e);
                const uint8_t al = (uint8_t)A[2 * m + e];
                switch (lay) {
                case 0: q[0] = r; q[1] = g; q[2] = b; break;
                case 1: q[0] = b; q[1] = g; q[2] = r; break;
                case 2: q[0] = al; q[1] = r; q[2] = g; q[3] = b; break;
                case 3: q[0] = r; q[1] = g; q[2] = b; q[3] = al; break;
                case 4: q[0] = al; q[1] = b; q[2] = g; q[3] = r; break;
                default: q[0] = b; q[1] = g; q[2] = r; q[3] = al; break;
                }
            }
        }
        uint8_t *d = a.dst + (size_t)f * a.dst_fp + (ptrdiff_t)(y0 + y) * a.dst_stride + bp * (x0 + xq);
        if ((flags & 4) && npx == 4) {
            uint32_t *dw = reinterpret_cast<uint32_t *>(d);
            dw[0] = pack4(px[0], px[1], px[2], px[3]);
            dw[1] = pack4(px[4], px[5], px[6], px[7]);
            dw[2] = pack4(px[8], px[9], px[10], px[11]);
            if (bp == 4)
                dw[3] = pack4(px[12], px[13], px[14], px[15]);
        } else {
            for (int i = 0; i < bp * npx; i++)
                d[i] = px[i];
        }
    }
}

static size_t rgb_lds(const FFHipScaleRgbArgs &a, int *spl, int *spc, int *osc, int *ohl, int *ohc, int *osa = nullptr, int *oha = nullptr)
{
    *spl = ((a.max_cols_l + 6) & ~3) + 8;
    *spc = ((a.max_cols_c + 6) & ~3) + 8;
    size_t s = (size_t)a.max_rows_l * *spl;
    s = (s + 15) & ~(size_t)15; *osc = (int)s;
    s += (size_t)2 * a.max_rows_c * *spc;
    s = (s + 15) & ~(size_t)15; *ohl = (int)s;
    s += (size_t)a.max_rows_l * a.tw * 2;
    s = (s + 15) & ~(size_t)15; *ohc = (int)s;
    s += (size_t)2 * a.max_rows_c * (a.full ? a.tw : a.tw >> 1) * 2;
    if (a.has_alpha) { /* the alpha plane's source tile and horizontal samples: the luma's sizes once more */
        s = (s + 15) & ~(size_t)15;
        if (osa) *osa = (int)s;
        s += (size_t)a.max_rows_l * *spl;
        s = (s + 15) & ~(size_t)15;
        if (oha) *oha = (int)s;
        s += (size_t)a.max_rows_l * a.tw * 2;
    }
    return s + 16;
}

int ffhip_plan_scale_rgb(FFHipScaleRgbArgs *a, const int32_t *hl, const int32_t *hc, const int32_t *vl, const int32_t *vc)
{
    static const int tws[] = { 256, 128, 64 };
    static const int ths[] = { 32, 16, 8, 4, 2, 1 };
    for (unsigned i = 0; i < 3; i++)
        for (unsigned j = 0; j < 6; j++) {
            const int tw = tws[i], th = ths[j];
            int mcl = 0, mcc = 0, mrl = 0, mrc = 0, t0, t1, t2, t3, t4;
            if (tw > 64 && tw >= 2 * a->dstW && i + 1 < 3)
                break;
            for (int x0 = 0; x0 < a->dstW; x0 += tw) {
                int xe = x0 + tw < a->dstW ? x0 + tw : a->dstW;
                int n = hl[xe - 1] + a->hl.size - (hl[x0] & ~3);
                int ce = a->full ? xe : (xe + 1) >> 1, c0 = a->full ? x0 : x0 >> 1;
                int m = hc[ce - 1] + a->hc.size - (hc[c0] & ~3);
                if (n > mcl) mcl = n;
                if (m > mcc) mcc = m;
            }
            for (int y0 = 0; y0 < a->dstH; y0 += th) {
                int ye = y0 + th < a->dstH ? y0 + th : a->dstH;
                int n = vl[ye - 1] + a->vl.size - vl[y0], m = vc[ye - 1] + a->vc.size - vc[y0];
                if (n > mrl) mrl = n;
                if (m > mrc) mrc = m;
            }
            a->tw = tw; a->th = th;
            a->max_cols_l = mcl; a->max_cols_c = mcc; a->max_rows_l = mrl; a->max_rows_c = mrc;
            if (rgb_lds(*a, &t0, &t1, &t2, &t3, &t4) <= 64 * 1024) {
                a->tiles_x = cdiv(a->dstW, tw);
                a->tiles_y = cdiv(a->dstH, th);
                return 0;
            }
        }
    ffhip_set_error("ffhip_sws: rgb filter footprint does not fit LDS");
    return FFHIP_EINVAL;
}

int ffhip_launch_scale_rgb(const FFHipScaleRgbArgs &a, hipStream_t stream)
{
    if (a.nframes <= 0)
        return 0;
    int spl, spc, osc, ohl, ohc, osa = 0, oha = 0, flags = 0;
    const size_t lds = rgb_lds(a, &spl, &spc, &osc, &ohl, &ohc, &osa, &oha);
    if (a.alpha && !(((uintptr_t)a.alpha | (size_t)a.alpha_stride | a.alpha_fp) & 3))
        flags |= 8;
    if (!(((uintptr_t)a.src[0] | (size_t)a.src_stride[0] | a.src_fp[0]) & 3))
        flags |= 1;
    if (a.chr_step == 1) {
        if (!(((uintptr_t)a.src[1] | (uintptr_t)a.src[2] | (size_t)a.src_stride[1] | (size_t)a.src_stride[2] |
               a.src_fp[1] | a.src_fp[2]) & 3))
            flags |= 2;
    } else {
        const uint8_t *b = a.src[2] < a.src[1] ? a.src[2] : a.src[1];
        if (!(((uintptr_t)b | (size_t)a.src_stride[1] | a.src_fp[1]) & 7))
            flags |= 2;
    }
    if (!(((uintptr_t)a.dst | (size_t)a.dst_stride | a.dst_fp) & 3))
        flags |= 4;
    const dim3 grid(a.tiles_x, a.tiles_y, a.nframes), block(NT);
    if (a.full) {
        if (a.hl.size == 4 && a.hc.size == 4)
            hipLaunchKernelGGL((k_scale_rgb<4, true>), grid, block, lds, stream, a, spl, spc, osc, ohl, ohc, flags, osa, oha);
        else
            hipLaunchKernelGGL((k_scale_rgb<0, true>), grid, block, lds, stream, a, spl, spc, osc, ohl, ohc, flags, osa, oha);
    } else if (a.hl.size == 4 && a.hc.size == 4)
        hipLaunchKernelGGL((k_scale_rgb<4>), grid, block, lds, stream, a, spl, spc, osc, ohl, ohc, flags, osa, oha);
    else
        hipLaunchKernelGGL((k_scale_rgb<0>), grid, block, lds, stream, a, spl, spc, osc, ohl, ohc, flags, osa, oha);
    LAUNCH_CHECK();
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* per-line parity faces: the reference's function-pointer granularity                         */
/* ------------------------------------------------------------------------------------------ */
__global__ void k_hscale8to15(int16_t *dst, int dstW, ptrdiff_t dstPitch, const uint8_t *src, ptrdiff_t srcPitch,
                              const int16_t *filter, const int32_t *pos, int fs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int line = blockIdx.y;
    if (i >= dstW)
        return;
    const uint8_t *s = src + line * srcPitch + pos[i];
    int v = 0;
    for (int j = 0; j < fs; j++)
        v += (int)s[j] * filter[(size_t)fs * i + j];
    *reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(dst) + line * dstPitch + 2 * (ptrdiff_t)i) =
        (int16_t)min(v >> 7, 32767);
}

int ffhip_launch_hscale8to15(int16_t *dst, int dstW, ptrdiff_t dstPitch, const uint8_t *src, ptrdiff_t srcPitch,
                             int nlines, const int16_t *filter, const int32_t *pos, int fs, hipStream_t stream)
{
    if (dstW <= 0 || nlines <= 0)
        return 0;
    hipLaunchKernelGGL(k_hscale8to15, dim3(cdiv(dstW, 256), nlines), dim3(256), 0, stream, dst, dstW, dstPitch, src,
                       srcPitch, filter, pos, fs);
    LAUNCH_CHECK();
    return 0;
}

__global__ void k_yuv2planeX8(const int16_t *filter, int fs, const int16_t *src, ptrdiff_t srcPitch, uint8_t *dest,
                              int dstW, const uint8_t *dither, int offset)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dstW)
        return;
    if (fs == 1) { /* yuv2plane1_8_c */
        dest[i] = (uint8_t)clip_u8((src[i] + dither[(i + offset) & 7]) >> 7);
        return;
    }
    uint32_t acc = (uint32_t)dither[(i + offset) & 7] << 12;
    for (int j = 0; j < fs; j++)
        acc += (uint32_t)(*reinterpret_cast<const int16_t *>(reinterpret_cast<const uint8_t *>(src) + j * srcPitch +
                                                             2 * (ptrdiff_t)i) * (int)filter[j]);
    dest[i] = (uint8_t)clip_u8((int32_t)acc >> 19);
}

int ffhip_launch_yuv2planeX8(const int16_t *filter, int fs, const int16_t *src, ptrdiff_t srcPitch, uint8_t *dest,
                             int dstW, const uint8_t *dither8, int offset, hipStream_t stream)
{
    if (dstW <= 0)
        return 0;
    hipLaunchKernelGGL(k_yuv2planeX8, dim3(cdiv(dstW, 256)), dim3(256), 0, stream, filter, fs, src, srcPitch, dest,
                       dstW, dither8, offset);
    LAUNCH_CHECK();
    return 0;
}

/* yuv2nv12cX_c (libswscale/output.c:500-529): one interleaved chroma line from fs int16 U lines and fs int16 V lines
 * (lines at usrc / vsrc + j * srcPitch bytes); swap: V first (NV21) */
__global__ void k_yuv2nv12cX(int swap, const uint8_t *dither, const int16_t *filter, int fs, const int16_t *usrc, const int16_t *vsrc,
                             ptrdiff_t srcPitch, uint8_t *dest, int chrDstW)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chrDstW)
        return;
    uint32_t au = (uint32_t)dither[i & 7] << 12, av = (uint32_t)dither[(i + 3) & 7] << 12;
    for (int j = 0; j < fs; j++) {
        const int c = filter[j];
        au += (uint32_t)(*reinterpret_cast<const int16_t *>(reinterpret_cast<const uint8_t *>(usrc) + j * srcPitch + 2 * (ptrdiff_t)i) * c);
        av += (uint32_t)(*reinterpret_cast<const int16_t *>(reinterpret_cast<const uint8_t *>(vsrc) + j * srcPitch + 2 * (ptrdiff_t)i) * c);
    }
    dest[2 * i + (swap ? 1 : 0)] = (uint8_t)clip_u8((int32_t)au >> 19);
    dest[2 * i + (swap ? 0 : 1)] = (uint8_t)clip_u8((int32_t)av >> 19);
}

int ffhip_launch_yuv2nv12cX(int swap, const uint8_t *dither8, const int16_t *filter, int fs, const int16_t *usrc, const int16_t *vsrc,
                            ptrdiff_t srcPitch, uint8_t *dest, int chrDstW, hipStream_t stream)
{
    if (chrDstW <= 0)
        return 0;
    hipLaunchKernelGGL(k_yuv2nv12cX, dim3(cdiv(chrDstW, 256)), dim3(256), 0, stream, swap, dither8, filter, fs, usrc, vsrc, srcPitch, dest, chrDstW);
    LAUNCH_CHECK();
    return 0;
}

/*
 * One packed RGB line from int16 luma / chroma lines: yuv2rgb_X_c_template / _2_ / _1_ (libswscale/output.c:1789-1939) with the
 * closed form of the reference's table lookup (sws_yuv2rgb.hip).  mode 0: X (lfs luma lines x lf[], cfs chroma lines x cf[]);
 * mode 1: _2 (two lines each, weights 4096 - alpha / alpha, no rounding term); mode 2: _1 (one luma line; chroma one line when
 * uvalpha == 0, else two).  One thread per pixel PAIR (the pair shares its chroma sample).  layout as everywhere: 0 rgb24 ... 5 bgra.
 */
__global__ void k_yuv2packed_line(int mode, const int16_t *lf, const int16_t *lum, int lfs, const int16_t *cf, const int16_t *cu,
                                  const int16_t *cv, int cfs, ptrdiff_t pitch, int yalpha, int uvalpha, uint8_t *dest, int dstW, int layout,
                                  FFHipYuv2RgbK k)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (dstW >> 1))
        return;
    auto L = [&](const int16_t *base, int j, int x) {
        return (int)*reinterpret_cast<const int16_t *>(reinterpret_cast<const uint8_t *>(base) + j * pitch + 2 * (ptrdiff_t)x);
    };
    int Y1, Y2, U, V;
    if (mode == 0) {
        uint32_t y1 = 1 << 18, y2 = 1 << 18, u = 1 << 18, v = 1 << 18;
        for (int j = 0; j < lfs; j++) {
            y1 += (uint32_t)(L(lum, j, 2 * i) * (int)lf[j]);
            y2 += (uint32_t)(L(lum, j, 2 * i + 1) * (int)lf[j]);
        }
        for (int j = 0; j < cfs; j++) {
            u += (uint32_t)(L(cu, j, i) * (int)cf[j]);
            v += (uint32_t)(L(cv, j, i) * (int)cf[j]);
        }
        Y1 = (int32_t)y1 >> 19; Y2 = (int32_t)y2 >> 19; U = (int32_t)u >> 19; V = (int32_t)v >> 19;
    } else if (mode == 1) {
        const int ya1 = 4096 - yalpha, uva1 = 4096 - uvalpha;
        Y1 = (L(lum, 0, 2 * i) * ya1 + L(lum, 1, 2 * i) * yalpha) >> 19;
        Y2 = (L(lum, 0, 2 * i + 1) * ya1 + L(lum, 1, 2 * i + 1) * yalpha) >> 19;
        U = (L(cu, 0, i) * uva1 + L(cu, 1, i) * uvalpha) >> 19;
        V = (L(cv, 0, i) * uva1 + L(cv, 1, i) * uvalpha) >> 19;
    } else {
        const int uva1 = 4096 - uvalpha;
        Y1 = (L(lum, 0, 2 * i) + 64) >> 7;
        Y2 = (L(lum, 0, 2 * i + 1) + 64) >> 7;
        if (!uvalpha) {
            U = (L(cu, 0, i) + 64) >> 7;
            V = (L(cv, 0, i) + 64) >> 7;
        } else {
            U = (L(cu, 0, i) * uva1 + L(cu, 1, i) * uvalpha + (128 << 11)) >> 19;
            V = (L(cv, 0, i) * uva1 + L(cv, 1, i) * uvalpha + (128 << 11)) >> 19;
        }
    }
    /* the reference indexes its tables with U, V as they are (they are in range for any sane bank); the closed form clips as the
     * fused kernels do */
    const int Uc = clip_u8(U), Vc = clip_u8(V);
    const int br = k.kb + (k.off_r + ((Vc * k.crv) >> 16)) * k.cy;
    const int bb = k.kb + (k.off_b + ((Uc * k.cbu) >> 16)) * k.cy;
    const int bg = k.kb + (k.off_g + ((Uc * k.cgu) >> 16) + ((Vc * k.cgv) >> 16)) * k.cy;
    const int bp = layout < 2 ? 3 : 4;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int yc = (e ? Y2 : Y1) * k.cy;
        const int r = clip_u8((br + yc) >> 16), g = clip_u8((bg + yc) >> 16), b = clip_u8((bb + yc) >> 16);
        uint8_t *q = dest + (size_t)bp * (2 * i + e);
        switch (layout) {
        case 0: q[0] = r; q[1] = g; q[2] = b; break;
        case 1: q[0] = b; q[1] = g; q[2] = r; break;
        case 2: q[0] = 255; q[1] = r; q[2] = g; q[3] = b; break;
        case 3: q[0] = r; q[1] = g; q[2] = b; q[3] = 255; break;
        case 4: q[0] = 255; q[1] = b; q[2] = g; q[3] = r; break;
        default: q[0] = b; q[1] = g; q[2] = r; q[3] = 255; break;
        }
    }
}

int ffhip_launch_yuv2packed_line(int mode, const int16_t *lf, const int16_t *lum, int lfs, const int16_t *cf, const int16_t *cu,
                                 const int16_t *cv, int cfs, ptrdiff_t pitch, int yalpha, int uvalpha, uint8_t *dest, int dstW, int layout,
                                 const FFHipYuv2RgbK &k, hipStream_t stream)
{
    if (dstW < 2)
        return 0;
    hipLaunchKernelGGL(k_yuv2packed_line, dim3(cdiv(dstW >> 1, 256)), dim3(256), 0, stream, mode, lf, lum, lfs, cf, cu, cv, cfs, pitch, yalpha,
                       uvalpha, dest, dstW, layout, k);
    LAUNCH_CHECK();
    return 0;
}
