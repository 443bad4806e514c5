/*
 * shims.hip — the signature-exact, HOST-pointer faces of h264dsp / h264qpel / h264chroma / h264pred / hevcdsp / vp9dsp / me_cmp /
 * float_dsp.
 *
 * These are the function pointers an `ff_h264dsp_init_hip()` / `ff_h264qpel_init_hip()` / `ff_me_cmp_init_hip()` ... installs next
 * to the x86/neon ones (libavcodec/h264dsp.c:155-169, h264qpel.c:105-119, me_cmp.c:1014-1026): same names, same argument
 * meaning, same side effects (coefficient blocks are cleared, dst is updated in place).  One call = one launch: the operands are
 * staged into device scratch, the SAME kernels as the batched faces run with n = 1, the results travel back.  That is the
 * reference's granularity — right for parity harnesses (checkasm exercises exactly these pointers), hopeless for speed;
 * throughput comes from the *_batch_dev entry points.
 *
 * These signatures cannot report an error, so a face must never fail silently:
 *   - every ff_*_init_hip() REMEMBERS the table it displaces (the reference calls an arch init with the C functions already
 *     in place), and a call that cannot run on the device — a HIP error anywhere, an argument outside the staged range, or
 *     FFHIP_FAULT=1 (test hook) — is answered by the displaced C function; ffhip_shim_fallbacks() counts them, and a face
 *     with nothing to fall back on records the member's name in ffhip_last_error();
 *   - host memory is written only after everything has come back: the whole scratch arena returns in ONE device-to-host copy
 *     into a host bounce buffer, results are committed from there (a failure can therefore not leave half a block behind for
 *     the C function to run on top of).
 * Thread-safe: one mutex (runtime.hip) guards the arena and the bounce buffer for a whole stage / run / copy-back sequence.
 */
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "kernels/common.h"
#include "kernels/h264_kernels.h"
#include "kernels/me_kernels.h"

#include "kernels/shim_arena.h"

extern "C" long ffhip_shim_fallbacks(void) { return g_fallbacks.load(); }

#include "kernels/shim_rect.h"

/* ---- h264dsp: single blocks ---------------------------------------------------------------------- */
static bool idct_single(int kind, uint8_t *dst, int16_t *block, ptrdiff_t stride)
{
    const int size = (kind & 1) ? 8 : 4, ncoef = size * size; /* IDCT4 0, IDCT8 1, IDCT4_DC 2, IDCT8_DC 3, ADD_PIXELS4 4, ADD_PIXELS8 5 */
    Rect d = { dst, stride, 0, size - 1, 0, size - 1, nullptr };
    Arena A(rect_bytes(d) + 256 + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    int16_t *dblk = (int16_t *)buf;
    int32_t *doff = (int32_t *)(buf + 128);
    const int32_t zero = 0;
    if (!rect_up(d, buf + 256) || hipMemcpy(dblk, block, ncoef * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(doff, &zero, 4, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_idct_add(kind, d.dev, DP, doff, dblk, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, d, 0, size - 1, 0, size - 1);
    memcpy(block, A.host(dblk), ncoef * 2); /* cleared by the kernel */
    return true;
}
static FFHipH264DSPContext g_fb_h264; /* the C functions ff_h264dsp_init_hip() displaced */
static void s_idct_add(uint8_t *d, int16_t *b, ptrdiff_t s) { if (!idct_single(FFHIP_H264_IDCT4, d, b, s)) SHIM_FB(g_fb_h264, idct_add, d, b, s); }
static void s_idct8_add(uint8_t *d, int16_t *b, ptrdiff_t s) { if (!idct_single(FFHIP_H264_IDCT8, d, b, s)) SHIM_FB(g_fb_h264, idct8_add, d, b, s); }
static void s_idct_dc_add(uint8_t *d, int16_t *b, ptrdiff_t s) { if (!idct_single(FFHIP_H264_IDCT4_DC, d, b, s)) SHIM_FB(g_fb_h264, idct_dc_add, d, b, s); }
static void s_idct8_dc_add(uint8_t *d, int16_t *b, ptrdiff_t s) { if (!idct_single(FFHIP_H264_IDCT8_DC, d, b, s)) SHIM_FB(g_fb_h264, idct8_dc_add, d, b, s); }
static void s_add_pixels4(uint8_t *d, int16_t *b, ptrdiff_t s) { if (!idct_single(FFHIP_H264_ADD_PIXELS4_CLEAR, d, b, s)) SHIM_FB(g_fb_h264, add_pixels4_clear, d, b, s); }
static void s_add_pixels8(uint8_t *d, int16_t *b, ptrdiff_t s) { if (!idct_single(FFHIP_H264_ADD_PIXELS8_CLEAR, d, b, s)) SHIM_FB(g_fb_h264, add_pixels8_clear, d, b, s); }

/* ---- h264dsp: macroblock dispatchers (idct_add16 / idct8_add4 / idct_add16intra) ---------------------- */
static bool idct_mb(int which, uint8_t *dst, const int *blockoffset, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc)
{
    if (stride <= 0 || stride > (1 << 16))
        return false; /* decoders hand these a positive linesize (h264_mb.c:728-779); anything else is the C function's */
    const int bs = which == 1 ? 8 : 4;
    int lo = blockoffset[0], hi = blockoffset[0];
    for (int i = 0; i < 16; i += (which == 1 ? 4 : 1)) {
        if (blockoffset[i] < lo) lo = blockoffset[i];
        if (blockoffset[i] > hi) hi = blockoffset[i];
    }
    const size_t span = (size_t)(hi - lo) + (size_t)(bs - 1) * stride + bs;
    Arena A(span + 512 + 64 + 64 + 64 + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    int16_t *dblk = (int16_t *)buf;              /* 512 B */
    int32_t *dbo = (int32_t *)(buf + 512);       /* 64 B  */
    uint8_t *dnn = buf + 576;                    /* 40 B  */
    int32_t *dmb = (int32_t *)(buf + 640);       /* 4 B   */
    uint8_t *dpix = buf + 704;                   /* flat copy of the touched span, same stride */
    const int32_t mboff = 0;
    int32_t bo[16];
    for (int i = 0; i < 16; i++)
        bo[i] = blockoffset[i] - lo;
    if (hipMemcpy(dpix, dst + lo, span, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dblk, block, 512, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dbo, bo, 64, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dnn, nnzc, 40, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dmb, &mboff, 4, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_idct_add_mb(which, dpix, stride, dmb, dbo, dblk, dnn, 1, 0) < 0 || !A.down())
        return false;
    for (int i = 0; i < 16; i += (which == 1 ? 4 : 1)) /* only this macroblock's own blocks travel back */
        commit2d(A, dst + blockoffset[i], stride, dpix + bo[i], stride, bs, bs);
    memcpy(block, A.host(dblk), 512);
    return true;
}
static void s_idct_add16(uint8_t *d, const int *bo, int16_t *b, ptrdiff_t s, const uint8_t n[5 * 8]) { if (!idct_mb(0, d, bo, b, s, n)) SHIM_FB(g_fb_h264, idct_add16, d, bo, b, s, n); }
static void s_idct8_add4(uint8_t *d, const int *bo, int16_t *b, ptrdiff_t s, const uint8_t n[5 * 8]) { if (!idct_mb(1, d, bo, b, s, n)) SHIM_FB(g_fb_h264, idct8_add4, d, bo, b, s, n); }
static void s_idct_add16intra(uint8_t *d, const int *bo, int16_t *b, ptrdiff_t s, const uint8_t n[5 * 8]) { if (!idct_mb(2, d, bo, b, s, n)) SHIM_FB(g_fb_h264, idct_add16intra, d, bo, b, s, n); }

/* idct_add8 (4:2:0): the two chroma planes' four blocks, blocks 16..19 / 32..35 of the macroblock's coefficient array, staged like
 * idct_mb (the span of each plane's four blocks, the 768 coefficients, the 48 offsets, the 15 x 8 cache) */
static bool idct_add8_gpu(uint8_t **dest, const int *blockoffset, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc)
{
    if (stride <= 0 || stride > (1 << 16))
        return false;
    int lo[2], hi[2];
    for (int j = 0; j < 2; j++) {
        lo[j] = hi[j] = blockoffset[16 * (j + 1)];
        for (int i = 16 * (j + 1); i < 16 * (j + 1) + 4; i++) {
            if (blockoffset[i] < lo[j]) lo[j] = blockoffset[i];
            if (blockoffset[i] > hi[j]) hi[j] = blockoffset[i];
        }
    }
    const size_t span[2] = { (size_t)(hi[0] - lo[0]) + 3 * (size_t)stride + 4, (size_t)(hi[1] - lo[1]) + 3 * (size_t)stride + 4 };
    const size_t sp0 = (span[0] + 63) & ~(size_t)63, sp1 = (span[1] + 63) & ~(size_t)63;
    Arena A(1536 + 192 + 128 + 64 + sp0 + sp1 + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    int16_t *dblk = (int16_t *)buf;              /* 1536 B */
    int32_t *dbo = (int32_t *)(buf + 1536);      /* 192 B  */
    uint8_t *dnn = buf + 1728;                   /* 120 B  */
    int32_t *dmb = (int32_t *)(buf + 1856);      /* 4 B    */
    uint8_t *dpix[2] = { buf + 1920, buf + 1920 + sp0 };
    const int32_t mboff = 0;
    int32_t bo[48] = { 0 };
    for (int j = 0; j < 2; j++)
        for (int i = 16 * (j + 1); i < 16 * (j + 1) + 4; i++)
            bo[i] = blockoffset[i] - lo[j];
    if (hipMemcpy(dpix[0], dest[0] + lo[0], span[0], hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dpix[1], dest[1] + lo[1], span[1], hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dblk, block, 1536, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dbo, bo, 192, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dnn, nnzc, 120, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dmb, &mboff, 4, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_idct_add8(dpix[0], dpix[1], stride, dmb, dbo, dblk, dnn, 1, 0) < 0 || !A.down())
        return false;
    for (int j = 0; j < 2; j++)
        for (int i = 16 * (j + 1); i < 16 * (j + 1) + 4; i++)
            commit2d(A, dest[j] + blockoffset[i], stride, dpix[j] + bo[i], stride, 4, 4);
    memcpy(block + 256, A.host(dblk + 256), 2 * 256 * sizeof(int16_t)); /* the chroma planes' coefficients (cleared where consumed) */
    return true;
}
static void s_idct_add8(uint8_t **d, const int *bo, int16_t *b, ptrdiff_t s, const uint8_t n[15 * 8]) { if (!idct_add8_gpu(d, bo, b, s, n)) SHIM_FB(g_fb_h264, idct_add8, d, bo, b, s, n); }

/* the DC transforms: 16 (luma) / 4 (chroma) values in, dequantised values out */
static bool dc_dequant_gpu(int luma, int16_t *output, int16_t *input, int qmul)
{
    Arena A(512 + 64 + 64 + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    int16_t *dout = (int16_t *)buf, *din = (int16_t *)(buf + 512);
    int32_t *dq = (int32_t *)(buf + 576), *doff = (int32_t *)(buf + 608);
    const int32_t q = qmul, zero = 0;
    if (hipMemcpy(dq, &q, 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(doff, &zero, 4, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (luma) {
        if (hipMemcpy(din, input, 32, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dout, output, 512, hipMemcpyHostToDevice) != hipSuccess ||
            ffhip_launch_h264_luma_dc_dequant(dout, 256, din, 16, dq, 1, 0) < 0 || !A.down())
            return false;
        for (int i = 0; i < 16; i++) /* the 16 DC positions are all the function writes */
            output[16 * i] = reinterpret_cast<const int16_t *>(A.host(dout))[16 * i];
    } else {
        if (hipMemcpy(dout, output, 128, hipMemcpyHostToDevice) != hipSuccess || ffhip_launch_h264_chroma_dc_dequant(dout, doff, dq, 1, 0) < 0 ||
            !A.down())
            return false;
        for (int i = 0; i < 4; i++)
            output[16 * i] = reinterpret_cast<const int16_t *>(A.host(dout))[16 * i];
    }
    return true;
}
static void s_luma_dc_dequant(int16_t *o, int16_t *i, int q) { if (!dc_dequant_gpu(1, o, i, q)) SHIM_FB(g_fb_h264, luma_dc_dequant_idct, o, i, q); }
static void s_chroma_dc_dequant(int16_t *b, int q) { if (!dc_dequant_gpu(0, b, nullptr, q)) SHIM_FB(g_fb_h264, chroma_dc_dequant_idct, b, q); }

/* ---- h264dsp: loop filters --------------------------------------------------------------------------- */
static bool lf_single(int kind, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, const int8_t *tc0)
{
    const bool chroma = kind & 2, vert_edge = kind & 1;
    const int along = chroma ? 8 : 16, across = chroma ? 2 : 4; /* samples read each side of the edge */
    Rect r = { pix, stride, vert_edge ? 0 : -across, vert_edge ? along - 1 : across - 1,
               vert_edge ? -across : 0, vert_edge ? across - 1 : along - 1, nullptr };
    Arena A(rect_bytes(r) + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    FFHipH264Edge e;
    memset(&e, 0, sizeof(e));
    e.kind = (uint8_t)kind;
    e.alpha = (uint8_t)(alpha < 0 ? 0 : alpha > 255 ? 255 : alpha); /* 8-bit tables top out at 255 / 18 */
    e.beta = (uint8_t)(beta < 0 ? 0 : beta > 255 ? 255 : beta);
    if (tc0)
        memcpy(e.tc0, tc0, 4);
    if (!rect_up(r, buf + 64) || hipMemcpy(buf, &e, sizeof(e), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_loop_filter(r.dev, DP, (const FFHipH264Edge *)buf, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, r, r.r0, r.r1, r.c0, r.c1);
    return true;
}
static void s_v_lf_luma(uint8_t *p, ptrdiff_t s, int a, int b, int8_t *t) { if (!lf_single(FFHIP_H264_LF_V_LUMA, p, s, a, b, t)) SHIM_FB(g_fb_h264, v_loop_filter_luma, p, s, a, b, t); }
static void s_h_lf_luma(uint8_t *p, ptrdiff_t s, int a, int b, int8_t *t) { if (!lf_single(FFHIP_H264_LF_H_LUMA, p, s, a, b, t)) SHIM_FB(g_fb_h264, h_loop_filter_luma, p, s, a, b, t); }
static void s_v_lf_chroma(uint8_t *p, ptrdiff_t s, int a, int b, int8_t *t) { if (!lf_single(FFHIP_H264_LF_V_CHROMA, p, s, a, b, t)) SHIM_FB(g_fb_h264, v_loop_filter_chroma, p, s, a, b, t); }
static void s_h_lf_chroma(uint8_t *p, ptrdiff_t s, int a, int b, int8_t *t) { if (!lf_single(FFHIP_H264_LF_H_CHROMA, p, s, a, b, t)) SHIM_FB(g_fb_h264, h_loop_filter_chroma, p, s, a, b, t); }
static void s_v_lf_luma_i(uint8_t *p, ptrdiff_t s, int a, int b) { if (!lf_single(FFHIP_H264_LF_V_LUMA_INTRA, p, s, a, b, nullptr)) SHIM_FB(g_fb_h264, v_loop_filter_luma_intra, p, s, a, b); }
static void s_h_lf_luma_i(uint8_t *p, ptrdiff_t s, int a, int b) { if (!lf_single(FFHIP_H264_LF_H_LUMA_INTRA, p, s, a, b, nullptr)) SHIM_FB(g_fb_h264, h_loop_filter_luma_intra, p, s, a, b); }
static void s_v_lf_chroma_i(uint8_t *p, ptrdiff_t s, int a, int b) { if (!lf_single(FFHIP_H264_LF_V_CHROMA_INTRA, p, s, a, b, nullptr)) SHIM_FB(g_fb_h264, v_loop_filter_chroma_intra, p, s, a, b); }
static void s_h_lf_chroma_i(uint8_t *p, ptrdiff_t s, int a, int b) { if (!lf_single(FFHIP_H264_LF_H_CHROMA_INTRA, p, s, a, b, nullptr)) SHIM_FB(g_fb_h264, h_loop_filter_chroma_intra, p, s, a, b); }

extern "C" int ff_h264dsp_init_hip(FFHipH264DSPContext *c, int bit_depth, int chroma_format_idc)
{
    if (!c || chroma_format_idc < 0 || chroma_format_idc > 3)
        return FFHIP_EINVAL;
    if (bit_depth != 8 && bit_depth != 9 && bit_depth != 10 && bit_depth != 12 && bit_depth != 14)
        return FFHIP_EINVAL; /* h264dsp.c:135-147: the depths H.264 defines */
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    FFHipH264DSPContext o = *c;
    if (bit_depth > 8) { /* every member from the depth-generic faces (shims_h264_hbd.hip) */
        const int r = ffhip_h264dsp_fill_generic(c, &o, bit_depth, chroma_format_idc == 2 ? 2 : 1);
        if (r >= 0)
            *c = o;
        return r;
    }
    o.v_loop_filter_luma = s_v_lf_luma;                 o.h_loop_filter_luma = s_h_lf_luma;
    o.v_loop_filter_luma_intra = s_v_lf_luma_i;         o.h_loop_filter_luma_intra = s_h_lf_luma_i;
    o.v_loop_filter_chroma = s_v_lf_chroma;             o.h_loop_filter_chroma = s_h_lf_chroma;
    o.v_loop_filter_chroma_intra = s_v_lf_chroma_i;     o.h_loop_filter_chroma_intra = s_h_lf_chroma_i;
    o.idct_add = s_idct_add;                            o.idct8_add = s_idct8_add;
    o.idct_dc_add = s_idct_dc_add;                      o.idct8_dc_add = s_idct8_dc_add;
    o.idct_add16 = s_idct_add16;                        o.idct8_add4 = s_idct8_add4;
    o.idct_add16intra = s_idct_add16intra;              o.idct_add8 = s_idct_add8;
    o.luma_dc_dequant_idct = s_luma_dc_dequant;         o.chroma_dc_dequant_idct = s_chroma_dc_dequant;
    o.add_pixels4_clear = s_add_pixels4;                o.add_pixels8_clear = s_add_pixels8;
    /* the MBAFF members, and for 4:2:2 the chroma forms the reference switches, come from the depth-generic faces at 8 bits */
    const int r = ffhip_h264dsp_fill_generic(c, &o, 8, chroma_format_idc == 2 ? 2 : 1);
    if (r < 0)
        return r;
    if (chroma_format_idc == 2) {
        /* the six members the reference picks by chroma format carry depth-generic 4:2:2 faces here, which keep their displaced C
         * functions in a table of their own (shims_h264_hbd.hip): g_fb_h264's slots stay the 4:2:0 faces' */
        FFHipH264DSPContext in = *c, ours = o;
        ours.h_loop_filter_chroma = in.h_loop_filter_chroma;               ours.h_loop_filter_chroma_intra = in.h_loop_filter_chroma_intra;
        ours.h_loop_filter_chroma_mbaff = in.h_loop_filter_chroma_mbaff;   ours.h_loop_filter_chroma_mbaff_intra = in.h_loop_filter_chroma_mbaff_intra;
        ours.idct_add8 = in.idct_add8;                                     ours.chroma_dc_dequant_idct = in.chroma_dc_dequant_idct;
        fb_snapshot(g_fb_h264, in, ours);
    } else {
        fb_snapshot(g_fb_h264, *c, o);
    }
    *c = o;
    return 0;
}

/* ---- h264qpel ---------------------------------------------------------------------------------------- */
static bool qpel_single(int avg, int size_idx, int mcxy, uint8_t *dst, const uint8_t *src, ptrdiff_t stride)
{
    const int n = 16 >> size_idx;
    Rect d = { dst, stride, 0, n - 1, 0, n - 1, nullptr };
    /* only what the reference function of this slot reads: the 6-tap margin exists on an axis only when that axis is
     * filtered (mc00 / mc0y / mcx0 read no margin on the other one; mc_dir_part emulates edges only then, h264_mb.c:206-300) */
    const bool fx = mcxy & 3, fy = mcxy >> 2;
    Rect s = { const_cast<uint8_t *>(src), stride, fy ? -2 : 0, fy ? n + 2 : n - 1, fx ? -2 : 0, fx ? n + 2 : n - 1, nullptr };
    Rect full = s;
    full.r0 = -2; full.r1 = n + 2; full.c0 = -2; full.c1 = n + 2; /* the device tile always has the margin (unread part: whatever) */
    Arena A(rect_bytes(d) + rect_bytes(full) + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    /* the (possibly margin-less) source rectangle lands where a full-margin one would: row -2, column -2 at the tile's origin */
    if (!rect_up(d, buf + 64) || !rect_up(s, buf + 64 + rect_bytes(d) + (size_t)(s.r0 + 2) * DP + (s.c0 + 2)))
        return false;
    FFHipQpelBlock b;
    memset(&b, 0, sizeof(b));
    /* both rectangles sit in one scratch arena: offsets relative to its start, one shared pitch */
    b.dst_offset = (int32_t)(d.dev - buf); b.src_offset = (int32_t)(s.dev - buf);
    b.mcxy = (uint8_t)mcxy; b.size_idx = (uint8_t)size_idx; b.avg = (uint8_t)avg;
    if (hipMemcpy(buf, &b, sizeof(b), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_qpel(buf, buf, DP, (const FFHipQpelBlock *)buf, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, d, 0, n - 1, 0, n - 1);
    return true;
}
static FFHipH264QpelContext g_fb_qpel;
#define QPEL_FN(op, avg, sz, idx, mc) \
    static void s_##op##_qpel##sz##_mc##mc(uint8_t *d, const uint8_t *s, ptrdiff_t st) \
    { if (!qpel_single(avg, idx, mc, d, s, st)) SHIM_FB(g_fb_qpel, op##_h264_qpel_pixels_tab[idx][mc], d, s, st); }
#define QPEL_16(op, avg, sz, idx) \
    QPEL_FN(op, avg, sz, idx, 0) QPEL_FN(op, avg, sz, idx, 1) QPEL_FN(op, avg, sz, idx, 2) QPEL_FN(op, avg, sz, idx, 3) \
    QPEL_FN(op, avg, sz, idx, 4) QPEL_FN(op, avg, sz, idx, 5) QPEL_FN(op, avg, sz, idx, 6) QPEL_FN(op, avg, sz, idx, 7) \
    QPEL_FN(op, avg, sz, idx, 8) QPEL_FN(op, avg, sz, idx, 9) QPEL_FN(op, avg, sz, idx, 10) QPEL_FN(op, avg, sz, idx, 11) \
    QPEL_FN(op, avg, sz, idx, 12) QPEL_FN(op, avg, sz, idx, 13) QPEL_FN(op, avg, sz, idx, 14) QPEL_FN(op, avg, sz, idx, 15)
QPEL_16(put, 0, 16, 0) QPEL_16(put, 0, 8, 1) QPEL_16(put, 0, 4, 2)
QPEL_16(avg, 1, 16, 0) QPEL_16(avg, 1, 8, 1) QPEL_16(avg, 1, 4, 2)
#define QPEL_ROW(op, sz) { s_##op##_qpel##sz##_mc0, s_##op##_qpel##sz##_mc1, s_##op##_qpel##sz##_mc2, s_##op##_qpel##sz##_mc3, \
    s_##op##_qpel##sz##_mc4, s_##op##_qpel##sz##_mc5, s_##op##_qpel##sz##_mc6, s_##op##_qpel##sz##_mc7, s_##op##_qpel##sz##_mc8, \
    s_##op##_qpel##sz##_mc9, s_##op##_qpel##sz##_mc10, s_##op##_qpel##sz##_mc11, s_##op##_qpel##sz##_mc12, \
    s_##op##_qpel##sz##_mc13, s_##op##_qpel##sz##_mc14, s_##op##_qpel##sz##_mc15 }

extern "C" int ff_h264qpel_init_hip(FFHipH264QpelContext *c, int bit_depth)
{
    if (!c || (bit_depth != 8 && bit_depth != 9 && bit_depth != 10 && bit_depth != 12 && bit_depth != 14))
        return FFHIP_EINVAL; /* h264qpel.c:87-103: the depths H.264 defines */
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    if (bit_depth > 8)
        return ffhip_h264qpel_init_generic(c, bit_depth); /* 16-bit samples: shims_h264_hbd.hip */
    /* table index = X + 4*Y, [0] 16x16 [1] 8x8 [2] 4x4 (h264qpel.c:55-70) */
    static const ffhip_qpel_mc_func put[3][16] = { QPEL_ROW(put, 16), QPEL_ROW(put, 8), QPEL_ROW(put, 4) };
    static const ffhip_qpel_mc_func avg[3][16] = { QPEL_ROW(avg, 16), QPEL_ROW(avg, 8), QPEL_ROW(avg, 4) };
    FFHipH264QpelContext o = *c;
    memcpy(o.put_h264_qpel_pixels_tab, put, sizeof(put));
    memcpy(o.avg_h264_qpel_pixels_tab, avg, sizeof(avg));
    fb_snapshot(g_fb_qpel, *c, o);
    *c = o;
    return 0;
}

/* ---- h264chroma / weighted prediction ------------------------------------------------------------------ */
static bool chroma_single(int avg, int w_idx, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h, int x, int y)
{
    const int w = 8 >> w_idx;
    if (h <= 0 || h > 16)
        return false;
    Rect d = { dst, stride, 0, h - 1, 0, w - 1, nullptr };
    Rect s = { const_cast<uint8_t *>(src), stride, 0, (y & 7) ? h : h - 1, 0, (x & 7) ? w : w - 1, nullptr };
    Arena A(rect_bytes(d) + rect_bytes(s) + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    if (!rect_up(d, buf + 64) || !rect_up(s, buf + 64 + rect_bytes(d)))
        return false;
    FFHipChromaBlock b;
    memset(&b, 0, sizeof(b));
    b.dst_offset = (int32_t)(d.dev - buf); b.src_offset = (int32_t)(s.dev - buf);
    b.w_idx = (uint8_t)w_idx; b.h = (uint8_t)h; b.x = (uint8_t)x; b.y = (uint8_t)y; b.avg = (uint8_t)avg;
    if (hipMemcpy(buf, &b, sizeof(b), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_chroma_mc(buf, buf, DP, (const FFHipChromaBlock *)buf, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, d, 0, h - 1, 0, w - 1);
    return true;
}
static FFHipH264ChromaContext g_fb_chroma;
#define CHROMA_FN(op, avg, idx) \
    static void s_##op##_chroma##idx(uint8_t *d, const uint8_t *s, ptrdiff_t st, int h, int x, int y) \
    { if (!chroma_single(avg, idx, d, s, st, h, x, y)) SHIM_FB(g_fb_chroma, op##_h264_chroma_pixels_tab[idx], d, s, st, h, x, y); }
CHROMA_FN(put, 0, 0) CHROMA_FN(put, 0, 1) CHROMA_FN(put, 0, 2) CHROMA_FN(avg, 1, 0) CHROMA_FN(avg, 1, 1) CHROMA_FN(avg, 1, 2)

extern "C" int ff_h264chroma_init_hip(FFHipH264ChromaContext *c, int bit_depth)
{
    if (!c || bit_depth < 8 || bit_depth > 16)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    if (bit_depth > 8) /* h264chroma.c:40-52: ONE 16-bit instantiation serves every depth above 8 (bilinear: nothing depends on the depth) */
        return ffhip_h264chroma_init_generic(c, 10);
    FFHipH264ChromaContext o = *c;
    o.put_h264_chroma_pixels_tab[0] = s_put_chroma0; o.put_h264_chroma_pixels_tab[1] = s_put_chroma1;
    o.put_h264_chroma_pixels_tab[2] = s_put_chroma2;
    o.avg_h264_chroma_pixels_tab[0] = s_avg_chroma0; o.avg_h264_chroma_pixels_tab[1] = s_avg_chroma1;
    o.avg_h264_chroma_pixels_tab[2] = s_avg_chroma2;
    fb_snapshot(g_fb_chroma, *c, o);
    *c = o;
    return 0;
}

static bool weight_single(int bi, int w_idx, uint8_t *dst, uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd,
                          int weights, int offset)
{
    const int w = 16 >> w_idx;
    if (height <= 0 || height > 16)
        return false;
    Rect d = { dst, stride, 0, height - 1, 0, w - 1, nullptr };
    Rect s = { bi ? src : dst, stride, 0, height - 1, 0, w - 1, nullptr };
    Arena A(rect_bytes(d) + rect_bytes(s) + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    if (!rect_up(d, buf + 64) || (bi && !rect_up(s, buf + 64 + rect_bytes(d))))
        return false;
    FFHipWeightBlock b;
    memset(&b, 0, sizeof(b));
    b.dst_offset = (int32_t)(d.dev - buf); b.src_offset = bi ? (int32_t)(s.dev - buf) : b.dst_offset;
    b.w_idx = (uint8_t)w_idx; b.height = (uint8_t)height; b.log2_denom = (uint8_t)log2_denom; b.bi = (uint8_t)bi;
    b.weightd = (int16_t)weightd; b.weights = (int16_t)weights; b.offset = (int16_t)offset;
    if (hipMemcpy(buf, &b, sizeof(b), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_weight(buf, buf, DP, (const FFHipWeightBlock *)buf, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, d, 0, height - 1, 0, w - 1);
    return true;
}
static FFHipH264WeightContext g_fb_weight;
#define WEIGHT_FN(idx) \
    static void s_weight##idx(uint8_t *b, ptrdiff_t st, int h, int ld, int w, int o) \
    { if (!weight_single(0, idx, b, nullptr, st, h, ld, w, 0, o)) SHIM_FB(g_fb_weight, weight_pixels_tab[idx], b, st, h, ld, w, o); } \
    static void s_biweight##idx(uint8_t *d, uint8_t *s, ptrdiff_t st, int h, int ld, int wd, int ws, int o) \
    { if (!weight_single(1, idx, d, s, st, h, ld, wd, ws, o)) SHIM_FB(g_fb_weight, biweight_pixels_tab[idx], d, s, st, h, ld, wd, ws, o); }
WEIGHT_FN(0) WEIGHT_FN(1) WEIGHT_FN(2) WEIGHT_FN(3)

extern "C" int ff_h264dsp_weight_init_hip(FFHipH264WeightContext *c, int bit_depth)
{
    if (!c || (bit_depth != 8 && bit_depth != 9 && bit_depth != 10 && bit_depth != 12 && bit_depth != 14))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    if (bit_depth > 8)
        return ffhip_h264weight_init_generic(c, bit_depth);
    FFHipH264WeightContext o = *c;
    o.weight_pixels_tab[0] = s_weight0; o.weight_pixels_tab[1] = s_weight1; o.weight_pixels_tab[2] = s_weight2;
    o.weight_pixels_tab[3] = s_weight3;
    o.biweight_pixels_tab[0] = s_biweight0; o.biweight_pixels_tab[1] = s_biweight1; o.biweight_pixels_tab[2] = s_biweight2;
    o.biweight_pixels_tab[3] = s_biweight3;
    fb_snapshot(g_fb_weight, *c, o);
    *c = o;
    return 0;
}

/* ---- hevcdsp: every face exists per bit depth (8: uint8_t samples; 10 / 12: uint16_t, hevc/dsp.c:133-196) — the members carry
 * no context, so the depth is baked into the function like the reference's template instantiations.  ps = bytes per sample; the
 * staged rectangles and pitches below are in bytes. ------------------------------------------------------------------------- */
static constexpr int hevc_bdi(int bd) { return bd == 8 ? 0 : bd == 10 ? 1 : 2; }
static FFHipHEVCDSPContext g_fb_hevc[3];
#define HEVC_FB(BD) g_fb_hevc[hevc_bdi(BD)]

/* layout in scratch: [0,64) the TU record, [64, 64+2*n*n) coefficients, then the picture rectangle */
static bool hevc_single(int bd, int kind, int log2_size, int16_t *coeffs, int col_limit, uint8_t *dst, ptrdiff_t stride)
{
    const int n = 1 << log2_size, ps = bd > 8 ? 2 : 1;
    const size_t cbytes = (size_t)n * n * 2;
    Rect d = { dst, stride, 0, n - 1, 0, n * ps - 1, nullptr };
    Arena A(64 + cbytes + (dst ? rect_bytes(d) : 0) + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    if (hipMemcpy(buf + 64, coeffs, cbytes, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (dst && !rect_up(d, buf + 64 + cbytes))
        return false;
    FFHipHevcTU tu;
    tu.coeff_offset = 0;
    tu.dst_offset = dst ? (int32_t)(d.dev - (buf + 64 + cbytes)) : -1;
    tu.col_limit = col_limit;
    if (hipMemcpy(buf, &tu, sizeof(tu), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_hevc_idct_bd(bd, kind, log2_size, (int16_t *)(buf + 64), dst ? buf + 64 + cbytes : nullptr, DP, (const FFHipHevcTU *)buf, 1, 0) < 0 ||
        !A.down())
        return false;
    if (kind != FFHIP_HEVC_ADD_ONLY)
        memcpy(coeffs, A.host(buf + 64), cbytes);
    if (dst)
        rect_commit(A, d, 0, n - 1, 0, n * ps - 1);
    return true;
}
template <int BD, int IDX> static void s_hevc_idct(int16_t *c, int col_limit)
{ if (!hevc_single(BD, FFHIP_HEVC_IDCT, IDX + 2, c, col_limit, nullptr, 0)) SHIM_FB(HEVC_FB(BD), idct[IDX], c, col_limit); }
template <int BD, int IDX> static void s_hevc_dc(int16_t *c)
{ if (!hevc_single(BD, FFHIP_HEVC_IDCT_DC, IDX + 2, c, 0, nullptr, 0)) SHIM_FB(HEVC_FB(BD), idct_dc[IDX], c); }
template <int BD, int IDX> static void s_hevc_add(uint8_t *d, const int16_t *r, ptrdiff_t st)
{ if (!hevc_single(BD, FFHIP_HEVC_ADD_ONLY, IDX + 2, const_cast<int16_t *>(r), 0, d, st)) SHIM_FB(HEVC_FB(BD), add_residual[IDX], d, r, st); }
template <int BD> static void s_hevc_dst4(int16_t *c)
{ if (!hevc_single(BD, FFHIP_HEVC_DST_4X4, 2, c, 0, nullptr, 0)) SHIM_FB(HEVC_FB(BD), transform_4x4_luma, c); }

/* one edge segment: the 8 lines x 8 samples around it staged as a rectangle (h_: rows -4..3 x cols 0..7, v_: rows 0..7 x cols -4..3) */
static bool hevc_lf_single(int bd, int kind, uint8_t *pix, ptrdiff_t stride, int beta, const int32_t *tc, const uint8_t *no_p, const uint8_t *no_q)
{
    const bool vertical = kind & 1;
    const int ps = bd > 8 ? 2 : 1;
    Rect d = { pix, stride, vertical ? 0 : -4, vertical ? 7 : 3, vertical ? -4 * ps : 0, vertical ? 4 * ps - 1 : 8 * ps - 1, nullptr };
    Arena A(rect_bytes(d) + 128);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    if (!rect_up(d, buf + 64))
        return false;
    FFHipHevcEdge e;
    memset(&e, 0, sizeof(e));
    e.offset = (int32_t)(d.dev - buf); e.kind = (uint8_t)kind; e.beta = (uint8_t)beta;
    for (int j = 0; j < 2; j++) { e.tc[j] = (int16_t)tc[j]; e.no_p[j] = no_p[j]; e.no_q[j] = no_q[j]; }
    if (hipMemcpy(buf, &e, sizeof(e), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_hevc_loop_filter_bd(bd, buf, DP, (const FFHipHevcEdge *)buf, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, d, d.r0, d.r1, d.c0, d.c1);
    return true;
}
template <int BD> static void s_hevc_lf_hl(uint8_t *p, ptrdiff_t st, int beta, const int32_t *tc, const uint8_t *np_, const uint8_t *nq)
{ if (!hevc_lf_single(BD, FFHIP_HEVC_LF_H_LUMA, p, st, beta, tc, np_, nq)) SHIM_FB(HEVC_FB(BD), hevc_h_loop_filter_luma, p, st, beta, tc, np_, nq); }
template <int BD> static void s_hevc_lf_vl(uint8_t *p, ptrdiff_t st, int beta, const int32_t *tc, const uint8_t *np_, const uint8_t *nq)
{ if (!hevc_lf_single(BD, FFHIP_HEVC_LF_V_LUMA, p, st, beta, tc, np_, nq)) SHIM_FB(HEVC_FB(BD), hevc_v_loop_filter_luma, p, st, beta, tc, np_, nq); }
template <int BD> static void s_hevc_lf_hc(uint8_t *p, ptrdiff_t st, const int32_t *tc, const uint8_t *np_, const uint8_t *nq)
{ if (!hevc_lf_single(BD, FFHIP_HEVC_LF_H_CHROMA, p, st, 0, tc, np_, nq)) SHIM_FB(HEVC_FB(BD), hevc_h_loop_filter_chroma, p, st, tc, np_, nq); }
template <int BD> static void s_hevc_lf_vc(uint8_t *p, ptrdiff_t st, const int32_t *tc, const uint8_t *np_, const uint8_t *nq)
{ if (!hevc_lf_single(BD, FFHIP_HEVC_LF_V_CHROMA, p, st, 0, tc, np_, nq)) SHIM_FB(HEVC_FB(BD), hevc_v_loop_filter_chroma, p, st, tc, np_, nq); }

/* SAO: source rows -1..height (edge: with one sample of margin) and the destination block packed at a pitch of 192 bytes */
static bool hevc_sao_single(int bd, int edge, uint8_t *dst, const uint8_t *src, ptrdiff_t sd, ptrdiff_t ss, const int16_t *off, int cls, int w, int h)
{
    if (w <= 0 || h <= 0 || w > 64 || h > 64)
        return false;
    const int P = 192, mg = edge ? 1 : 0, ps = bd > 8 ? 2 : 1;
    Arena A(64 + (size_t)(h + 2) * P * 2 + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf, *dsrc = buf + 64, *ddst = dsrc + (size_t)(h + 2) * P;
    const size_t rowb = (size_t)(w + 2 * mg) * ps;
    if (ss >= (ptrdiff_t)rowb) {
        if (hipMemcpy2D(dsrc + (size_t)(1 - mg) * P + (1 - mg) * ps, P, src - mg * ss - mg * ps, ss, rowb, h + 2 * mg, hipMemcpyHostToDevice) != hipSuccess)
            return false;
    } else {
        for (int y = -mg; y < h + mg; y++)
            if (hipMemcpy(dsrc + (size_t)(y + 1) * P + (1 - mg) * ps, src + y * ss - mg * ps, rowb, hipMemcpyHostToDevice) != hipSuccess)
                return false;
    }
    FFHipHevcSao k;
    memset(&k, 0, sizeof(k));
    k.dst_offset = 0; k.src_offset = P + ps;
    for (int i = 0; i < 5; i++) k.offset_val[i] = off[i];
    k.edge = (uint8_t)edge; k.cls = (uint8_t)cls; k.width = (uint8_t)w; k.height = (uint8_t)h;
    if (hipMemcpy(buf, &k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_hevc_sao_bd(bd, ddst, P, dsrc, P, (const FFHipHevcSao *)buf, 1, 0) < 0 || !A.down())
        return false;
    commit2d(A, dst, sd, ddst, P, (size_t)w * ps, h);
    return true;
}
/* the reference's table index of a block width: sao_tab[(FFALIGN(width, 8) >> 3) - 1] (libavcodec/hevc/filter.c) */
static int hevc_sao_tab(int w) { static const uint8_t t[8] = { 0, 1, 2, 2, 3, 3, 4, 4 }; const int k = ((w + 7) >> 3) - 1; return t[k < 0 ? 0 : k > 7 ? 7 : k]; }
template <int BD> static void s_hevc_sao_band(uint8_t *d, const uint8_t *s, ptrdiff_t sd, ptrdiff_t ss, const int16_t *o, int lc, int w, int h)
{ if (!hevc_sao_single(BD, 0, d, s, sd, ss, o, lc, w, h)) SHIM_FB(HEVC_FB(BD), sao_band_filter[hevc_sao_tab(w)], d, s, sd, ss, o, lc, w, h); }
template <int BD> static void s_hevc_sao_edge(uint8_t *d, const uint8_t *s, ptrdiff_t sd, const int16_t *o, int eo, int w, int h)
{ if (!hevc_sao_single(BD, 1, d, s, sd, 192, o, eo, w, h)) SHIM_FB(HEVC_FB(BD), sao_edge_filter[hevc_sao_tab(w)], d, s, sd, o, eo, w, h); }

/* MC: source rows -3..height+4 x columns -3..width+4 at a pitch of 128 samples; destination after it (pixels: pitch 64 samples;
 * int16: 64 elements); modes 2..4 (FFHIP_HEVC_MC_*): src2's height x 64 int16 after the destination */
static bool hevc_mc_single(int bd, int chroma, int uni, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, int mx,
                           int my, int width, const int16_t *src2 = nullptr, int denom = 0, int wx0 = 0, int wx1 = 0, int ox = 0)
{
    if (width <= 0 || height <= 0 || width > 64 || height > 64)
        return false;
    const int ps = bd > 8 ? 2 : 1, P = 128 * ps, DPX = 64 * ps, before = chroma ? 1 : 3, after = chroma ? 2 : 4;
    const size_t sbytes = (size_t)(height + before + after) * P, dbytes = (size_t)height * (uni ? DPX : 128);
    const size_t s2bytes = uni >= 3 ? (size_t)height * 128 : 0;
    Arena A(64 + sbytes + dbytes + s2bytes + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf, *dsrc = buf + 64, *ddst = dsrc + sbytes, *dsrc2 = ddst + dbytes;
    if (s2bytes && (!src2 || hipMemcpy(dsrc2, src2, s2bytes - (size_t)(64 - width) * 2, hipMemcpyHostToDevice) != hipSuccess))
        return false;
    {
        /* only what the reference function of this slot reads: margins exist on an axis only when that axis is filtered */
        const int by = my ? before : 0, ay = my ? after : 0, bx = mx ? before : 0, ax = mx ? after : 0;
        const int cols = width + bx + ax, rows = height + by + ay;
        uint8_t *d0 = dsrc + (size_t)(before - by) * P + (size_t)(before - bx) * ps;
        const uint8_t *s0 = src - by * srcstride - bx * ps;
        if (srcstride >= (ptrdiff_t)cols * ps) {
            if (hipMemcpy2D(d0, P, s0, srcstride, (size_t)cols * ps, rows, hipMemcpyHostToDevice) != hipSuccess)
                return false;
        } else {
            for (int y = 0; y < rows; y++)
                if (hipMemcpy(d0 + (size_t)y * P, s0 + y * srcstride, (size_t)cols * ps, hipMemcpyHostToDevice) != hipSuccess)
                    return false;
        }
    }
    if (uni >= 2) {
        FFHipHevcMcWBlock k = {};
        k.src_offset = before * P + before * ps;
        k.width = (uint8_t)width; k.height = (uint8_t)height; k.mx = (uint8_t)mx; k.my = (uint8_t)my;
        k.wx0 = (int16_t)wx0; k.wx1 = (int16_t)wx1; k.ox = (int16_t)ox; k.denom = (uint8_t)denom;
        if (hipMemcpy(buf, &k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess)
            return false;
    } else {
        FFHipHevcMcBlock k;
        k.dst_offset = 0; k.src_offset = before * P + before * ps;
        k.width = (uint8_t)width; k.height = (uint8_t)height; k.mx = (uint8_t)mx; k.my = (uint8_t)my;
        if (hipMemcpy(buf, &k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess)
            return false;
    }
    if (ffhip_launch_hevc_mc_bd(bd, chroma, uni, ddst, DPX, dsrc, P, (const int16_t *)dsrc2, buf, 1, 0) < 0 || !A.down())
        return false;
    if (uni)
        commit2d(A, dst, dststride, ddst, DPX, (size_t)width * ps, height);
    else
        commit2d(A, dst, 128, ddst, 128, (size_t)width * 2, height); /* int16 rows of MAX_PB_SIZE = 64 elements */
    return true;
}
/* table slot of a call: [ff_hevc_pel_weight[width]][!!my][!!mx] (libavcodec/hevc/dsp.c, hevcdec.c) */
static int hevc_pw(int w) { return w <= 2 ? 0 : w <= 4 ? 1 : w <= 6 ? 2 : w <= 8 ? 3 : w <= 12 ? 4 : w <= 16 ? 5 : w <= 24 ? 6 : w <= 32 ? 7 : w <= 48 ? 8 : 9; }
#define HEVC_SLOT(tab, w, mx, my) tab[hevc_pw(w)][(my) != 0][(mx) != 0]
template <int BD> static void s_hevc_qpel(int16_t *d, const uint8_t *s, ptrdiff_t ss, int h, intptr_t mx, intptr_t my, int w)
{ if (!hevc_mc_single(BD, 0, 0, d, 0, s, ss, h, (int)mx, (int)my, w)) SHIM_FB(HEVC_FB(BD), HEVC_SLOT(put_hevc_qpel, w, mx, my), d, s, ss, h, mx, my, w); }
template <int BD> static void s_hevc_epel(int16_t *d, const uint8_t *s, ptrdiff_t ss, int h, intptr_t mx, intptr_t my, int w)
{ if (!hevc_mc_single(BD, 1, 0, d, 0, s, ss, h, (int)mx, (int)my, w)) SHIM_FB(HEVC_FB(BD), HEVC_SLOT(put_hevc_epel, w, mx, my), d, s, ss, h, mx, my, w); }
template <int BD> static void s_hevc_qpel_uni(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int h, intptr_t mx, intptr_t my, int w)
{ if (!hevc_mc_single(BD, 0, 1, d, ds, s, ss, h, (int)mx, (int)my, w)) SHIM_FB(HEVC_FB(BD), HEVC_SLOT(put_hevc_qpel_uni, w, mx, my), d, ds, s, ss, h, mx, my, w); }
template <int BD> static void s_hevc_epel_uni(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int h, intptr_t mx, intptr_t my, int w)
{ if (!hevc_mc_single(BD, 1, 1, d, ds, s, ss, h, (int)mx, (int)my, w)) SHIM_FB(HEVC_FB(BD), HEVC_SLOT(put_hevc_epel_uni, w, mx, my), d, ds, s, ss, h, mx, my, w); }
template <int BD> static void s_hevc_dequant(int16_t *c, int16_t log2_size)
{ if (!hevc_single(BD, FFHIP_HEVC_DEQUANT, log2_size, c, 0, nullptr, 0)) SHIM_FB(HEVC_FB(BD), dequant, c, log2_size); }
template <int BD> static void s_hevc_rdpcm(int16_t *c, int16_t log2_size, int mode)
{
    if (!hevc_single(BD, mode ? FFHIP_HEVC_RDPCM_V : FFHIP_HEVC_RDPCM_H, log2_size, c, 0, nullptr, 0))
        SHIM_FB(HEVC_FB(BD), transform_rdpcm, c, log2_size, mode);
}
/* sao_edge_restore: rows of the block at a pitch of 64 samples for both buffers; only the block's own samples travel */
static bool hevc_restore_single(int bd, int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t sd, ptrdiff_t ss, const FFHipSAOParams *sao,
                                const int *borders, int w, int h, int c_idx, const uint8_t *ve, const uint8_t *he, const uint8_t *de)
{
    if (w <= 0 || h <= 0 || w > 64 || h > 64 || c_idx < 0 || c_idx > 2)
        return false;
    const int ps = bd > 8 ? 2 : 1, P = 64 * ps;
    Arena A(64 + 2 * (size_t)h * P + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf, *dsrc = buf + 64, *ddst = dsrc + (size_t)h * P;
    if (hipMemcpy2D(dsrc, P, src, ss, (size_t)w * ps, h, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy2D(ddst, P, dst, sd, (size_t)w * ps, h, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    FFHipHevcSaoRestore k = {};
    k.offset0 = sao->offset_val[c_idx][0];
    k.width = (uint8_t)w; k.height = (uint8_t)h; k.eo = (uint8_t)sao->eo_class[c_idx]; k.variant = (uint8_t)variant;
    for (int i = 0; i < 4; i++) {
        k.borders |= (borders[i] != 0) << i;
        if (variant)
            k.diag_edge |= (de[i] != 0) << i;
    }
    if (variant) {
        k.vert_edge = (ve[0] != 0) | (ve[1] != 0) << 1;
        k.horiz_edge = (he[0] != 0) | (he[1] != 0) << 1;
    }
    if (hipMemcpy(buf, &k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_hevc_sao_restore_bd(bd, ddst, P, dsrc, P, (const FFHipHevcSaoRestore *)buf, 1, 0) < 0 || !A.down())
        return false;
    commit2d(A, dst, sd, ddst, P, (size_t)w * ps, h);
    return true;
}
template <int BD, int VAR>
static void s_hevc_restore(uint8_t *d, const uint8_t *s, ptrdiff_t sd, ptrdiff_t ss, const FFHipSAOParams *sao, const int *b, int w, int h,
                           int c, const uint8_t *ve, const uint8_t *he, const uint8_t *de)
{ if (!hevc_restore_single(BD, VAR, d, s, sd, ss, sao, b, w, h, c, ve, he, de)) SHIM_FB(HEVC_FB(BD), sao_edge_restore[VAR], d, s, sd, ss, sao, b, w, h, c, ve, he, de); }
#define HEVC_W_SHIMS(name, chroma)                                                                                                          \
template <int BD>                                                                                                                           \
static void s_hevc_##name##_uni_w(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int h, int denom, int wx, int ox, intptr_t mx,  \
                                  intptr_t my, int w)                                                                                       \
{ if (!hevc_mc_single(BD, chroma, FFHIP_HEVC_MC_UNI_W, d, ds, s, ss, h, (int)mx, (int)my, w, nullptr, denom, wx, 0, ox))                   \
      SHIM_FB(HEVC_FB(BD), HEVC_SLOT(put_hevc_##name##_uni_w, w, mx, my), d, ds, s, ss, h, denom, wx, ox, mx, my, w); }                       \
template <int BD>                                                                                                                           \
static void s_hevc_##name##_bi(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, const int16_t *s2, int h, intptr_t mx,             \
                               intptr_t my, int w)                                                                                          \
{ if (!hevc_mc_single(BD, chroma, FFHIP_HEVC_MC_BI, d, ds, s, ss, h, (int)mx, (int)my, w, s2))                                             \
      SHIM_FB(HEVC_FB(BD), HEVC_SLOT(put_hevc_##name##_bi, w, mx, my), d, ds, s, ss, s2, h, mx, my, w); }                                     \
template <int BD>                                                                                                                           \
static void s_hevc_##name##_bi_w(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, const int16_t *s2, int h, int denom, int wx0,    \
                                 int wx1, int ox, intptr_t mx, intptr_t my, int w)                                                          \
{ if (!hevc_mc_single(BD, chroma, FFHIP_HEVC_MC_BI_W, d, ds, s, ss, h, (int)mx, (int)my, w, s2, denom, wx0, wx1, ox))                      \
      SHIM_FB(HEVC_FB(BD), HEVC_SLOT(put_hevc_##name##_bi_w, w, mx, my), d, ds, s, ss, s2, h, denom, wx0, wx1, ox, mx, my, w); }
HEVC_W_SHIMS(qpel, 0)
HEVC_W_SHIMS(epel, 1)

template <int BD>
static void hevc_fill(FFHipHEVCDSPContext &o)
{
    o.idct[0] = s_hevc_idct<BD, 0>; o.idct[1] = s_hevc_idct<BD, 1>; o.idct[2] = s_hevc_idct<BD, 2>; o.idct[3] = s_hevc_idct<BD, 3>;
    o.idct_dc[0] = s_hevc_dc<BD, 0>; o.idct_dc[1] = s_hevc_dc<BD, 1>; o.idct_dc[2] = s_hevc_dc<BD, 2>; o.idct_dc[3] = s_hevc_dc<BD, 3>;
    o.add_residual[0] = s_hevc_add<BD, 0>; o.add_residual[1] = s_hevc_add<BD, 1>; o.add_residual[2] = s_hevc_add<BD, 2>;
    o.add_residual[3] = s_hevc_add<BD, 3>;
    o.transform_4x4_luma = s_hevc_dst4<BD>;
    o.dequant = s_hevc_dequant<BD>; o.transform_rdpcm = s_hevc_rdpcm<BD>;
    o.sao_edge_restore[0] = s_hevc_restore<BD, 0>; o.sao_edge_restore[1] = s_hevc_restore<BD, 1>;
    o.hevc_h_loop_filter_luma = o.hevc_h_loop_filter_luma_c = s_hevc_lf_hl<BD>;
    o.hevc_v_loop_filter_luma = o.hevc_v_loop_filter_luma_c = s_hevc_lf_vl<BD>;
    o.hevc_h_loop_filter_chroma = o.hevc_h_loop_filter_chroma_c = s_hevc_lf_hc<BD>;
    o.hevc_v_loop_filter_chroma = o.hevc_v_loop_filter_chroma_c = s_hevc_lf_vc<BD>;
    for (int i = 0; i < 5; i++) {
        o.sao_band_filter[i] = s_hevc_sao_band<BD>;
        o.sao_edge_filter[i] = s_hevc_sao_edge<BD>;
    }
    /* the [!!my][!!mx] slots all take (mx, my): one function per table serves every slot */
    for (int i = 0; i < 10; i++)
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                o.put_hevc_qpel[i][a][b] = s_hevc_qpel<BD>; o.put_hevc_qpel_uni[i][a][b] = s_hevc_qpel_uni<BD>;
                o.put_hevc_epel[i][a][b] = s_hevc_epel<BD>; o.put_hevc_epel_uni[i][a][b] = s_hevc_epel_uni<BD>;
                o.put_hevc_qpel_uni_w[i][a][b] = s_hevc_qpel_uni_w<BD>; o.put_hevc_epel_uni_w[i][a][b] = s_hevc_epel_uni_w<BD>;
                o.put_hevc_qpel_bi[i][a][b] = s_hevc_qpel_bi<BD>; o.put_hevc_epel_bi[i][a][b] = s_hevc_epel_bi<BD>;
                o.put_hevc_qpel_bi_w[i][a][b] = s_hevc_qpel_bi_w<BD>; o.put_hevc_epel_bi_w[i][a][b] = s_hevc_epel_bi_w<BD>;
            }
}

extern "C" int ff_hevc_dsp_init_hip(FFHipHEVCDSPContext *c, int bit_depth)
{
    if (!c || (bit_depth != 8 && bit_depth != 10 && bit_depth != 12))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    FFHipHEVCDSPContext o = *c;
    if (bit_depth == 8)
        hevc_fill<8>(o);
    else if (bit_depth == 10)
        hevc_fill<10>(o);
    else
        hevc_fill<12>(o);
    fb_snapshot(g_fb_hevc[hevc_bdi(bit_depth)], *c, o);
    *c = o;
    return 0;
}

/* ---- AVFloatDSPContext ------------------------------------------------------------------------------------ */
/* operands packed one after another in scratch, each rounded up to 16 bytes */
static bool fdsp_single(int op, float *dst, int dst_n, const float *s0, int n0, const float *s1, int n1, const float *s2, int n2, float mul,
                        int len)
{
    if (len <= 0)
        return false;
    const size_t bd = ((size_t)dst_n * 4 + 15) & ~(size_t)15, b0 = ((size_t)n0 * 4 + 15) & ~(size_t)15, b1 = ((size_t)n1 * 4 + 15) & ~(size_t)15,
                 b2 = ((size_t)n2 * 4 + 15) & ~(size_t)15;
    Arena A(bd + b0 + b1 + b2 + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    float *dd = (float *)buf, *d0 = (float *)(buf + bd), *d1 = (float *)(buf + bd + b0), *d2 = (float *)(buf + bd + b0 + b1);
    const bool dst_in = op == FFHIP_FDSP_FMAC_SCALAR || op == FFHIP_FDSP_BUTTERFLIES;
    if ((dst_in && hipMemcpy(dd, dst, (size_t)dst_n * 4, hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemcpy(d0, s0, (size_t)n0 * 4, hipMemcpyHostToDevice) != hipSuccess ||
        (s1 && hipMemcpy(d1, s1, (size_t)n1 * 4, hipMemcpyHostToDevice) != hipSuccess) ||
        (s2 && hipMemcpy(d2, s2, (size_t)n2 * 4, hipMemcpyHostToDevice) != hipSuccess))
        return false;
    if (ffhip_launch_fdsp(op, dd, 0, d0, 0, s1 ? d1 : nullptr, 0, s2 ? d2 : nullptr, 0, mul, len, 1, 0) < 0 || !A.down())
        return false;
    memcpy(dst, A.host(dd), (size_t)dst_n * 4);
    if (op == FFHIP_FDSP_BUTTERFLIES)
        memcpy(const_cast<float *>(s0), A.host(d0), (size_t)n0 * 4);
    return true;
}
static FFHipFloatDSPContext g_fb_fdsp;
static void s_fd_fmul(float *d, const float *a, const float *b, int n) { if (!fdsp_single(FFHIP_FDSP_FMUL, d, n, a, n, b, n, nullptr, 0, 0, n)) SHIM_FB(g_fb_fdsp, vector_fmul, d, a, b, n); }
static void s_fd_fmac(float *d, const float *a, float m, int n) { if (!fdsp_single(FFHIP_FDSP_FMAC_SCALAR, d, n, a, n, nullptr, 0, nullptr, 0, m, n)) SHIM_FB(g_fb_fdsp, vector_fmac_scalar, d, a, m, n); }
static void s_fd_fmuls(float *d, const float *a, float m, int n) { if (!fdsp_single(FFHIP_FDSP_FMUL_SCALAR, d, n, a, n, nullptr, 0, nullptr, 0, m, n)) SHIM_FB(g_fb_fdsp, vector_fmul_scalar, d, a, m, n); }
static void s_fd_window(float *d, const float *a, const float *b, const float *w, int n) { if (!fdsp_single(FFHIP_FDSP_FMUL_WINDOW, d, 2 * n, a, n, b, n, w, 2 * n, 0, n)) SHIM_FB(g_fb_fdsp, vector_fmul_window, d, a, b, w, n); }
static void s_fd_fmadd(float *d, const float *a, const float *b, const float *c, int n) { if (!fdsp_single(FFHIP_FDSP_FMUL_ADD, d, n, a, n, b, n, c, n, 0, n)) SHIM_FB(g_fb_fdsp, vector_fmul_add, d, a, b, c, n); }
static void s_fd_frev(float *d, const float *a, const float *b, int n) { if (!fdsp_single(FFHIP_FDSP_FMUL_REVERSE, d, n, a, n, b, n, nullptr, 0, 0, n)) SHIM_FB(g_fb_fdsp, vector_fmul_reverse, d, a, b, n); }
static void s_fd_bfly(float *a, float *b, int n) { if (!fdsp_single(FFHIP_FDSP_BUTTERFLIES, a, n, b, n, nullptr, 0, nullptr, 0, 0, n)) SHIM_FB(g_fb_fdsp, butterflies_float, a, b, n); }

extern "C" int ff_float_dsp_init_hip(FFHipFloatDSPContext *c)
{
    if (!c)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    FFHipFloatDSPContext o = *c;
    o.vector_fmul = s_fd_fmul; o.vector_fmac_scalar = s_fd_fmac; o.vector_fmul_scalar = s_fd_fmuls;
    o.vector_fmul_window = s_fd_window; o.vector_fmul_add = s_fd_fmadd; o.vector_fmul_reverse = s_fd_frev;
    o.butterflies_float = s_fd_bfly;
    fb_snapshot(g_fb_fdsp, *c, o);
    *c = o;
    return 0;
}

/* ---- me_cmp --------------------------------------------------------------------------------------------- */
static bool cmp_single(int kind, int width, const uint8_t *blk1, const uint8_t *blk2, ptrdiff_t stride, int h, int *result)
{
    const int rows = kind == FFHIP_ME_SATD ? (width == 16 ? (h == 16 ? 16 : 8) : 8) : h;
    if (rows <= 0) {
        *result = 0;
        return true;
    }
    /* what the C function of the slot reads: the half-pel forms one more column / row of blk2; sse and nsse stay inside the blocks */
    const int bx = kind == FFHIP_ME_SAD_X2 || kind == FFHIP_ME_SAD_XY2, by = kind == FFHIP_ME_SAD_Y2 || kind == FFHIP_ME_SAD_XY2;
    Rect a = { const_cast<uint8_t *>(blk1), stride, 0, rows - 1, 0, width - 1, nullptr };
    Rect b = { const_cast<uint8_t *>(blk2), stride, 0, rows - 1 + by, 0, width - 1 + bx, nullptr };
    Arena A(rect_bytes(a) + rect_bytes(b) + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    if (!rect_up(a, buf + 64) || !rect_up(b, buf + 64 + rect_bytes(a)))
        return false;
    const int32_t offs[2] = { (int32_t)(a.dev - buf), (int32_t)(b.dev - buf) };
    int32_t *d = (int32_t *)buf; /* [0] off1 [1] off2 [2] result */
    if (hipMemcpy(d, offs, 8, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_me_cmp(kind, width, kind == FFHIP_ME_SATD ? rows : h, buf, d, buf, d + 1, DP, d + 2, 1, 0) < 0 || !A.down())
        return false;
    *result = reinterpret_cast<const int32_t *>(A.host(d))[2];
    return true;
}
static FFHipMECmpContext g_fb_me;
/* an int-returning face: the displaced function's value, or 0 (and the recorded error) when there is none */
#define CMP_SHIM(name, kind, width, member)                                                                   \
    static int name(void *c, const uint8_t *a, const uint8_t *b, ptrdiff_t s, int h)                          \
    {                                                                                                         \
        int r = 0;                                                                                            \
        if (cmp_single(kind, width, a, b, s, h, &r))                                                          \
            return r;                                                                                         \
        const bool have_ = g_fb_me.member != nullptr;                                                         \
        shim_note(#member, have_);                                                                            \
        return have_ ? g_fb_me.member(c, a, b, s, h) : 0;                                                     \
    }
CMP_SHIM(s_sad16, FFHIP_ME_SAD, 16, sad[0])
CMP_SHIM(s_sad8, FFHIP_ME_SAD, 8, sad[1])
CMP_SHIM(s_pixabs16, FFHIP_ME_SAD, 16, pix_abs[0][0])
CMP_SHIM(s_pixabs8, FFHIP_ME_SAD, 8, pix_abs[1][0])
CMP_SHIM(s_satd16, FFHIP_ME_SATD, 16, hadamard8_diff[0])
CMP_SHIM(s_satd8, FFHIP_ME_SATD, 8, hadamard8_diff[1])
CMP_SHIM(s_pixabs16_x2, FFHIP_ME_SAD_X2, 16, pix_abs_hpel[0][0])
CMP_SHIM(s_pixabs16_y2, FFHIP_ME_SAD_Y2, 16, pix_abs_hpel[0][1])
CMP_SHIM(s_pixabs16_xy2, FFHIP_ME_SAD_XY2, 16, pix_abs_hpel[0][2])
CMP_SHIM(s_pixabs8_x2, FFHIP_ME_SAD_X2, 8, pix_abs_hpel[1][0])
CMP_SHIM(s_pixabs8_y2, FFHIP_ME_SAD_Y2, 8, pix_abs_hpel[1][1])
CMP_SHIM(s_pixabs8_xy2, FFHIP_ME_SAD_XY2, 8, pix_abs_hpel[1][2])
CMP_SHIM(s_sse16, FFHIP_ME_SSE, 16, sse[0])
CMP_SHIM(s_sse8, FFHIP_ME_SSE, 8, sse[1])
/* nsse: the second term's weight comes from the encoder context when there is one (me_cmp.c:404-407): those calls are the C function's */
#define NSSE_SHIM(name, width, member)                                                                        \
    static int name(void *c, const uint8_t *a, const uint8_t *b, ptrdiff_t s, int h)                          \
    {                                                                                                         \
        int r = 0;                                                                                            \
        if (!c && cmp_single(FFHIP_ME_NSSE, width, a, b, s, h, &r))                                           \
            return r;                                                                                         \
        const bool have_ = g_fb_me.member != nullptr;                                                         \
        if (!c || !have_)                                                                                     \
            shim_note(#member, have_);                                                                        \
        return have_ ? g_fb_me.member(c, a, b, s, h) : 0;                                                     \
    }
NSSE_SHIM(s_nsse16, 16, nsse[0])
NSSE_SHIM(s_nsse8, 8, nsse[1])

extern "C" int ff_me_cmp_init_hip(FFHipMECmpContext *c)
{
    if (!c)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    FFHipMECmpContext o = *c;
    o.sad[0] = s_sad16;              o.sad[1] = s_sad8;
    o.pix_abs[0][0] = s_pixabs16;    o.pix_abs[1][0] = s_pixabs8;    /* ff_me_cmp_init: me_cmp.c:989-1000 */
    o.hadamard8_diff[0] = s_satd16;  o.hadamard8_diff[1] = s_satd8;
    o.pix_abs_hpel[0][0] = s_pixabs16_x2; o.pix_abs_hpel[0][1] = s_pixabs16_y2; o.pix_abs_hpel[0][2] = s_pixabs16_xy2;
    o.pix_abs_hpel[1][0] = s_pixabs8_x2;  o.pix_abs_hpel[1][1] = s_pixabs8_y2;  o.pix_abs_hpel[1][2] = s_pixabs8_xy2;
    o.sse[0] = s_sse16;              o.sse[1] = s_sse8;
    o.nsse[0] = s_nsse16;            o.nsse[1] = s_nsse8;
    fb_snapshot(g_fb_me, *c, o);
    *c = o;
    return 0;
}

/* ---- vp9dsp: every face per bit depth (8: uint8_t samples, int16 coefficients; 10 / 12: uint16_t, int32 — vp9dsp_10bpp.c /
 * vp9dsp_12bpp.c), the depth baked into the function like the reference's instantiations.  ps = bytes per sample. ---- */
#define VP9_FB(tab, BD) tab[hevc_bdi(BD)]

/* itxfm_add: the block and the size x size picture rectangle travel through scratch */
static bool vp9_itxfm_single(int bd, int tx, int txtp, uint8_t *dst, ptrdiff_t stride, int16_t *block, int eob)
{
    const int n = tx == 4 ? 4 : 4 << tx, ps = bd > 8 ? 2 : 1, P = 64 * ps;
    const size_t cbytes = (size_t)n * n * (bd > 8 ? 4 : 2);
    Arena A(64 + cbytes + (size_t)n * P + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    uint8_t *dco = buf + 64;
    uint8_t *ddst = buf + 64 + cbytes;
    if (hipMemcpy(dco, block, cbytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy2D(ddst, P, dst, stride, (size_t)n * ps, n, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    FFHipVp9TU k = {};
    k.txtp = (uint8_t)txtp; k.dc_only = eob == 1;
    if (hipMemcpy(buf, &k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_vp9_itxfm_bd(bd, tx, dco, ddst, P, (const FFHipVp9TU *)buf, 1, 0) < 0 || !A.down())
        return false;
    memcpy(block, A.host(dco), cbytes);
    commit2d(A, dst, stride, ddst, P, (size_t)n * ps, n);
    return true;
}
static FFHipVP9ItxfmContext g_fb_vp9itx[3];
template <int BD, int TX, int TP>
static void s_vp9_itx(uint8_t *d, ptrdiff_t s, int16_t *b, int e)
{ if (!vp9_itxfm_single(BD, TX, TP, d, s, b, e)) SHIM_FB(VP9_FB(g_fb_vp9itx, BD), itxfm_add[TX][TP], d, s, b, e); }
template <int BD>
static void vp9_itx_fill(FFHipVP9ItxfmContext &o)
{
#define VP9_ROW(tx) o.itxfm_add[tx][0] = s_vp9_itx<BD, tx, 0>; o.itxfm_add[tx][1] = s_vp9_itx<BD, tx, 1>; \
                    o.itxfm_add[tx][2] = s_vp9_itx<BD, tx, 2>; o.itxfm_add[tx][3] = s_vp9_itx<BD, tx, 3>;
    VP9_ROW(0) VP9_ROW(1) VP9_ROW(2) VP9_ROW(3) VP9_ROW(4)
#undef VP9_ROW
}
#define VP9_INIT_BODY(CTX, FILL, FB)                                          \
    if (!c || (bpp != 8 && bpp != 10 && bpp != 12))                           \
        return FFHIP_EINVAL;                                                  \
    if (!ffhip_have_device())                                                 \
        return FFHIP_ENOSYS;                                                  \
    CTX o = *c;                                                               \
    if (bpp == 8) FILL(8) else if (bpp == 10) FILL(10) else FILL(12)          \
    fb_snapshot(FB[hevc_bdi(bpp)], *c, o);                                    \
    *c = o;                                                                   \
    return 0;

extern "C" int ff_vp9dsp_itxfm_init_hip(FFHipVP9ItxfmContext *c, int bpp)
{
#define F_(B) vp9_itx_fill<B>(o);
    VP9_INIT_BODY(FFHipVP9ItxfmContext, F_, g_fb_vp9itx)
#undef F_
}

/* mc: source rows -3..h+4 x columns -3..w+4 at a pitch of 128 samples, destination w x h at a pitch of 64 samples */
static bool vp9_mc_single(int bd, int width, int filter, int avg, uint8_t *dst, ptrdiff_t ds, const uint8_t *src, ptrdiff_t ss, int h, int mx,
                          int my)
{
    if (h <= 0 || h > 64)
        return false;
    const int ps = bd > 8 ? 2 : 1, P = 128 * ps, DPX = 64 * ps, before = 3, after = 4;
    const size_t sbytes = (size_t)(h + before + after) * P, dbytes = (size_t)h * DPX;
    Arena A(64 + sbytes + dbytes + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf, *dsrc = buf + 64, *ddst = dsrc + sbytes;
    /* only what the reference function of this slot reads: rows / columns beyond the block exist when that axis is filtered */
    const int ry0 = my ? -before : 0, ry1 = my ? h + after : h, cx0 = mx ? -before : 0, cx1 = mx ? width + after : width;
    if (hipMemcpy2D(dsrc + (size_t)(ry0 + before) * P + (size_t)(before + cx0) * ps, P, src + ry0 * ss + cx0 * ps, ss, (size_t)(cx1 - cx0) * ps,
                    ry1 - ry0, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy2D(ddst, DPX, dst, ds, (size_t)width * ps, h, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    FFHipVp9McBlock k = {};
    k.src_offset = before * P + before * ps;
    k.width = (uint8_t)width; k.height = (uint8_t)h; k.filter = (uint8_t)filter; k.mx = (uint8_t)mx; k.my = (uint8_t)my; k.avg = (uint8_t)avg;
    if (hipMemcpy(buf, &k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_vp9_mc_bd(bd, ddst, DPX, dsrc, P, (const FFHipVp9McBlock *)buf, 1, 0) < 0 || !A.down())
        return false;
    commit2d(A, dst, ds, ddst, DPX, (size_t)width * ps, h);
    return true;
}
static FFHipVP9McContext g_fb_vp9mc[3];
/* the [!!mx][!!my] slots differ only in which fractions are non-zero: a slot's function masks the other one as the reference's
 * dedicated h / v functions ignore it */
template <int BD, int W, int I, int F, int AVG, int HX, int VY>
static void s_vp9_mc(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int h, int mx, int my)
{
    if (!vp9_mc_single(BD, W, F, AVG, d, ds, s, ss, h, HX ? mx : 0, VY ? my : 0))
        SHIM_FB(VP9_FB(g_fb_vp9mc, BD), mc[I][F][AVG][HX][VY], d, ds, s, ss, h, mx, my);
}
template <int BD, int W, int I>
static void vp9_mc_fill(FFHipVP9McContext *c)
{
#define VP9_MC_F(F) \
    c->mc[I][F][0][0][0] = s_vp9_mc<BD, W, I, F, 0, 0, 0>; c->mc[I][F][0][0][1] = s_vp9_mc<BD, W, I, F, 0, 0, 1>; \
    c->mc[I][F][0][1][0] = s_vp9_mc<BD, W, I, F, 0, 1, 0>; c->mc[I][F][0][1][1] = s_vp9_mc<BD, W, I, F, 0, 1, 1>; \
    c->mc[I][F][1][0][0] = s_vp9_mc<BD, W, I, F, 1, 0, 0>; c->mc[I][F][1][0][1] = s_vp9_mc<BD, W, I, F, 1, 0, 1>; \
    c->mc[I][F][1][1][0] = s_vp9_mc<BD, W, I, F, 1, 1, 0>; c->mc[I][F][1][1][1] = s_vp9_mc<BD, W, I, F, 1, 1, 1>;
    VP9_MC_F(0) VP9_MC_F(1) VP9_MC_F(2) VP9_MC_F(3)
#undef VP9_MC_F
}

extern "C" int ff_vp9dsp_mc_init_hip(FFHipVP9McContext *c, int bpp)
{
#define F_(B) { vp9_mc_fill<B, 64, 0>(&o); vp9_mc_fill<B, 32, 1>(&o); vp9_mc_fill<B, 16, 2>(&o); vp9_mc_fill<B, 8, 3>(&o); vp9_mc_fill<B, 4, 4>(&o); }
    VP9_INIT_BODY(FFHipVP9McContext, F_, g_fb_vp9mc)
#undef F_
}

/* loop filter: 16 lines x 16 samples around the edge travel through scratch (pitch 32 samples) */
static bool vp9_lf_single(int bd, int nseg, const int wd_idx[2], int dir, uint8_t *dst, ptrdiff_t stride, const int E[2], const int I[2],
                          const int H[2])
{
    const int ps = bd > 8 ? 2 : 1, P = 32 * ps, lines = 8 * nseg;
    Arena A(64 + 32 * P + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf, *d = buf + 64;
    /* device layout: the edge at column 8 (dir 0: rows = lines) or row 8 (dir 1: columns = lines); only the samples the
     * reference function of this slot touches travel: 8 on either side for the 16-wide filter, 4 otherwise */
    const int r = (wd_idx[0] == 2 || (nseg == 2 && wd_idx[1] == 2)) ? 8 : 4;
    const int rows = dir ? 2 * r : lines, cols = dir ? lines : 2 * r;
    const uint8_t *h0 = dir ? dst - r * stride : dst - r * ps;
    uint8_t *dd = dir ? d + (8 - r) * P : d + (8 - r) * ps;
    if (hipMemcpy2D(dd, P, h0, stride, (size_t)cols * ps, rows, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    FFHipVp9Edge k[2] = {};
    for (int sgm = 0; sgm < nseg; sgm++) {
        k[sgm].offset = dir ? 8 * P + 8 * sgm * ps : 8 * sgm * P + 8 * ps;
        k[sgm].wd_idx = (uint8_t)wd_idx[sgm]; k[sgm].dir = (uint8_t)dir;
        k[sgm].E = (uint8_t)E[sgm]; k[sgm].I = (uint8_t)I[sgm]; k[sgm].H = (uint8_t)H[sgm];
    }
    if (hipMemcpy(buf, k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_vp9_loop_filter_bd(bd, d, P, (const FFHipVp9Edge *)buf, nseg, 0) < 0 || !A.down())
        return false;
    commit2d(A, (uint8_t *)h0, stride, dd, P, (size_t)cols * ps, rows);
    return true;
}
static FFHipVP9LoopFilterContext g_fb_vp9lf[3];
template <int BD, int WD, int DIR>
static void s_vp9_lf8(uint8_t *d, ptrdiff_t s, int E, int I, int H)
{
    const int w[2] = { WD, 0 }, e[2] = { E, 0 }, i[2] = { I, 0 }, h[2] = { H, 0 };
    if (!vp9_lf_single(BD, 1, w, DIR, d, s, e, i, h))
        SHIM_FB(VP9_FB(g_fb_vp9lf, BD), loop_filter_8[WD][DIR], d, s, E, I, H);
}
template <int BD, int DIR>
static void s_vp9_lf16(uint8_t *d, ptrdiff_t s, int E, int I, int H)
{
    const int w[2] = { 2, 2 }, e[2] = { E, E }, i[2] = { I, I }, h[2] = { H, H };
    if (!vp9_lf_single(BD, 2, w, DIR, d, s, e, i, h))
        SHIM_FB(VP9_FB(g_fb_vp9lf, BD), loop_filter_16[DIR], d, s, E, I, H);
}
template <int BD, int W1, int W2, int DIR>
static void s_vp9_lfmix(uint8_t *d, ptrdiff_t s, int E, int I, int H)
{
    const int w[2] = { W1, W2 }, e[2] = { E & 0xff, E >> 8 }, i[2] = { I & 0xff, I >> 8 }, h[2] = { H & 0xff, H >> 8 };
    if (!vp9_lf_single(BD, 2, w, DIR, d, s, e, i, h))
        SHIM_FB(VP9_FB(g_fb_vp9lf, BD), loop_filter_mix2[W1][W2][DIR], d, s, E, I, H);
}
template <int BD>
static void vp9_lf_fill(FFHipVP9LoopFilterContext &o)
{
    o.loop_filter_8[0][0] = s_vp9_lf8<BD, 0, 0>; o.loop_filter_8[0][1] = s_vp9_lf8<BD, 0, 1>;
    o.loop_filter_8[1][0] = s_vp9_lf8<BD, 1, 0>; o.loop_filter_8[1][1] = s_vp9_lf8<BD, 1, 1>;
    o.loop_filter_8[2][0] = s_vp9_lf8<BD, 2, 0>; o.loop_filter_8[2][1] = s_vp9_lf8<BD, 2, 1>;
    o.loop_filter_16[0] = s_vp9_lf16<BD, 0>; o.loop_filter_16[1] = s_vp9_lf16<BD, 1>;
    o.loop_filter_mix2[0][0][0] = s_vp9_lfmix<BD, 0, 0, 0>; o.loop_filter_mix2[0][0][1] = s_vp9_lfmix<BD, 0, 0, 1>;
    o.loop_filter_mix2[0][1][0] = s_vp9_lfmix<BD, 0, 1, 0>; o.loop_filter_mix2[0][1][1] = s_vp9_lfmix<BD, 0, 1, 1>;
    o.loop_filter_mix2[1][0][0] = s_vp9_lfmix<BD, 1, 0, 0>; o.loop_filter_mix2[1][0][1] = s_vp9_lfmix<BD, 1, 0, 1>;
    o.loop_filter_mix2[1][1][0] = s_vp9_lfmix<BD, 1, 1, 0>; o.loop_filter_mix2[1][1][1] = s_vp9_lfmix<BD, 1, 1, 1>;
}

extern "C" int ff_vp9dsp_loopfilter_init_hip(FFHipVP9LoopFilterContext *c, int bpp)
{
#define F_(B) vp9_lf_fill<B>(o);
    VP9_INIT_BODY(FFHipVP9LoopFilterContext, F_, g_fb_vp9lf)
#undef F_
}

/* intra_pred: the edge line is assembled from exactly the samples the mode reads */
static FFHipVP9IntraContext g_fb_vp9intra[3];
template <int BD, int TX, int MODE>
static bool vp9_intra_gpu(uint8_t *dst, ptrdiff_t stride, const uint8_t *left, const uint8_t *top)
{
    constexpr int N = 4 << TX, PS = BD > 8 ? 2 : 1;
    constexpr bool use_top = MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7 || MODE == 9 || MODE == 11;
    constexpr bool use_left = MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 8 || MODE == 9 || MODE == 10;
    constexpr bool use_tl = MODE == 4 || MODE == 5 || MODE == 6 || MODE == 9;
    constexpr int ntop = (TX == 0 && (MODE == 3 || MODE == 7)) ? 8 : N;
    uint8_t e[(32 + 1 + 32 + 8) * PS] = { 0 };
    if (use_left) memcpy(e, left, N * PS);
    if (use_tl) memcpy(e + N * PS, top - PS, PS);
    if (use_top) memcpy(e + (N + 1) * PS, top, ntop * PS);
    constexpr int P = 32 * PS;
    Arena A(64 + 256 + (size_t)N * P + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf, *de = buf + 64, *dd = de + 256;
    FFHipVp9Intra k = {};
    k.mode = MODE;
    if (hipMemcpy(buf, &k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(de, e, sizeof(e), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_vp9_intra_bd(BD, TX, dd, P, de, (const FFHipVp9Intra *)buf, 1, 0) < 0 || !A.down())
        return false;
    commit2d(A, dst, stride, dd, P, (size_t)N * PS, N);
    return true;
}
template <int BD, int TX, int MODE>
static void s_vp9_intra(uint8_t *dst, ptrdiff_t stride, const uint8_t *left, const uint8_t *top)
{
    if (!vp9_intra_gpu<BD, TX, MODE>(dst, stride, left, top))
        SHIM_FB(VP9_FB(g_fb_vp9intra, BD), intra_pred[TX][MODE], dst, stride, left, top);
}
template <int BD, int TX>
static void vp9_intra_fill(FFHipVP9IntraContext *c)
{
#define VI(M) c->intra_pred[TX][M] = s_vp9_intra<BD, TX, M>;
    VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) VI(8) VI(9) VI(10) VI(11) VI(12) VI(13) VI(14)
#undef VI
}

extern "C" int ff_vp9dsp_intrapred_init_hip(FFHipVP9IntraContext *c, int bpp)
{
#define F_(B) { vp9_intra_fill<B, 0>(&o); vp9_intra_fill<B, 1>(&o); vp9_intra_fill<B, 2>(&o); vp9_intra_fill<B, 3>(&o); }
    VP9_INIT_BODY(FFHipVP9IntraContext, F_, g_fb_vp9intra)
#undef F_
}

/* scaled mc: the source rectangle the call reads, at a pitch of 192 samples */
static FFHipVP9ScaledMcContext g_fb_vp9smc[3];
template <int BD, int W, int F, int AVG>
static bool vp9_smc_gpu(uint8_t *dst, ptrdiff_t ds, const uint8_t *src, ptrdiff_t ss, int h, int mx, int my, int dx, int dy)
{
    if (h <= 0 || h > 64 || dx < 1 || dx > 32 || dy < 1 || dy > 32)
        return false;
    constexpr int PS = BD > 8 ? 2 : 1;
    const int P = 192 * PS, DPX = 64 * PS, bil = F == 3, before = bil ? 0 : 3, after = bil ? 1 : 4;
    const int cols = ((mx + (W - 1) * dx) >> 4) + 1 + before + after, rows = ((my + (h - 1) * dy) >> 4) + 1 + before + after;
    const size_t sbytes = (size_t)rows * P, dbytes = (size_t)h * DPX;
    Arena A(64 + sbytes + dbytes + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf, *dsrc = buf + 64, *ddst = dsrc + sbytes;
    if (hipMemcpy2D(dsrc, P, src - before * ss - before * PS, ss, (size_t)cols * PS, rows, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy2D(ddst, DPX, dst, ds, (size_t)W * PS, h, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    FFHipVp9ScaledBlock k = {};
    k.src_offset = before * P + before * PS;
    k.width = W; k.height = (uint8_t)h; k.filter = F; k.mx = (uint8_t)mx; k.my = (uint8_t)my; k.avg = AVG; k.dx = (uint8_t)dx; k.dy = (uint8_t)dy;
    if (hipMemcpy(buf, &k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_vp9_smc_bd(BD, ddst, DPX, dsrc, P, (const FFHipVp9ScaledBlock *)buf, 1, 0) < 0 || !A.down())
        return false;
    commit2d(A, dst, ds, ddst, DPX, (size_t)W * PS, h);
    return true;
}
template <int BD, int W, int I, int F, int AVG>
static void s_vp9_smc(uint8_t *dst, ptrdiff_t ds, const uint8_t *src, ptrdiff_t ss, int h, int mx, int my, int dx, int dy)
{
    if (!vp9_smc_gpu<BD, W, F, AVG>(dst, ds, src, ss, h, mx, my, dx, dy))
        SHIM_FB(VP9_FB(g_fb_vp9smc, BD), smc[I][F][AVG], dst, ds, src, ss, h, mx, my, dx, dy);
}
template <int BD, int W, int I>
static void vp9_smc_fill(FFHipVP9ScaledMcContext *c)
{
    c->smc[I][0][0] = s_vp9_smc<BD, W, I, 0, 0>; c->smc[I][0][1] = s_vp9_smc<BD, W, I, 0, 1>;
    c->smc[I][1][0] = s_vp9_smc<BD, W, I, 1, 0>; c->smc[I][1][1] = s_vp9_smc<BD, W, I, 1, 1>;
    c->smc[I][2][0] = s_vp9_smc<BD, W, I, 2, 0>; c->smc[I][2][1] = s_vp9_smc<BD, W, I, 2, 1>;
    c->smc[I][3][0] = s_vp9_smc<BD, W, I, 3, 0>; c->smc[I][3][1] = s_vp9_smc<BD, W, I, 3, 1>;
}

extern "C" int ff_vp9dsp_scaled_mc_init_hip(FFHipVP9ScaledMcContext *c, int bpp)
{
#define F_(B) { vp9_smc_fill<B, 64, 0>(&o); vp9_smc_fill<B, 32, 1>(&o); vp9_smc_fill<B, 16, 2>(&o); vp9_smc_fill<B, 8, 3>(&o); vp9_smc_fill<B, 4, 4>(&o); }
    VP9_INIT_BODY(FFHipVP9ScaledMcContext, F_, g_fb_vp9smc)
#undef F_
}

/* ---- h264pred host faces: the picture patch is staged from exactly the neighbours the C member reads ---- */
/* need: bit0 left column, bit1 row above, bit2 corner, bit3 top-right (4x4: topright[0..3]; 8x8l: T9..15 if has_topright) */
#define HP_P 64 /* pitch of the staged patch; the block sits at row 1, column 16 */
/* bd: the depth the face was installed for (samples of 2 bytes and int32 coefficients above 8) */
static bool h264_pred_host(int bd, int kind, int mode, int n, unsigned need, int lrows, uint8_t *src, ptrdiff_t stride, const uint8_t *topright,
                           int has_tl, int has_tr, int16_t *block, int nh = 0)
{
    if (!nh)
        nh = n; /* rows of the block (8 x 16: n = 8 columns, nh = 16) */
    const int px = bd > 8 ? 2 : 1;
    uint8_t st[17 * HP_P] = { 0 };
    uint8_t *o = st + HP_P + 16;
    if (need & 1)
        for (int y = 0; y < lrows; y++)
            memcpy(o + y * HP_P - px, src + y * stride - px, px);
    if (need & 2) {
        memcpy(o - HP_P, src - stride, n * px);
        if (kind == FFHIP_H264_PRED8x8L || kind == FFHIP_H264_PRED8x8L_FILTER_ADD) {
            if (has_tr)
                memcpy(o + 8 * px - HP_P, src + 8 * px - stride, px);
            if (has_tr && (need & 8))
                memcpy(o - HP_P + 9 * px, src - stride + 9 * px, 7 * px);
        }
    }
    if (need & 4)
        memcpy(o - HP_P - px, src - stride - px, px);
    const bool k4 = kind == FFHIP_H264_PRED4x4 || (kind == FFHIP_H264_PRED_CODEC && n == 4); /* the 4x4 kinds take a topright pointer */
    if (k4 && (need & 8))
        memcpy(o - HP_P + 32, topright, 4 * px); /* wherever the caller's pointer leads, the record addresses the staged copy */
    const int ncoef = block ? n * n * px : 0; /* in int16 units: int32 coefficients above 8 bits */
    Arena A(64 + sizeof(st) + 256 + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf, *dp = buf + 64;
    int16_t *dc = (int16_t *)(dp + sizeof(st));
    FFHipH264Pred k = {};
    k.offset = HP_P + 16;
    k.aux = k4 ? 16 + 32 : 0; /* where topright[] was staged: row 0 of the patch */
    k.mode = (uint8_t)mode;
    k.flags = (uint8_t)((has_tl ? FFHIP_H264_PRED_TOPLEFT : 0) | (has_tr ? FFHIP_H264_PRED_TOPRIGHT : 0));
    if (hipMemcpy(buf, &k, sizeof(k), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dp, st, sizeof(st), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ncoef && hipMemcpy(dc, block, ncoef * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_pred_bd(bd, kind, dp, HP_P, dc, (const FFHipH264Pred *)buf, 1, 0) < 0 || !A.down())
        return false;
    commit2d(A, src, stride, dp + HP_P + 16, HP_P, (size_t)n * px, nh);
    if (ncoef)
        memcpy(block, A.host(dc), ncoef * sizeof(int16_t)); /* cleared by the kernel */
    return true;
}
/* the C functions our faces displaced, per depth; F = 1: the 4:2:2 forms of pred8x8[] / pred8x8_add[] (a 4:2:0 and a 4:2:2 context of
 * one depth hold different C functions there) */
template <int BD, int F = 0> struct PredFb { static FFHipH264PredContext t; };
template <int BD, int F> FFHipH264PredContext PredFb<BD, F>::t;
#define g_fb_pred PredFb<BD>::t
static constexpr unsigned hp_need4(int mode) { return (unsigned)(0x0211a777a312ull >> (4 * mode)) & 15u; } /* as kernels/h264_pred.hip */
/* pred8x8 / pred16x16: 0 DC 1 HOR 2 VERT 3 PLANE 4 LEFT_DC 5 TOP_DC 6 DC_128 7 L0T 8 0LT 9 L00 10 0L0 */
static constexpr unsigned hp_need_blk(int mode)
{
    return (mode == 0 || mode == 1 || mode == 3 || mode == 4 || mode >= 7 ? 1u : 0u) |
           (mode == 0 || mode == 2 || mode == 3 || mode == 5 || mode == 7 || mode == 8 ? 2u : 0u) | (mode == 3 ? 4u : 0u);
}
template <int BD, int MODE>
static void s_pred4x4(uint8_t *src, const uint8_t *topright, ptrdiff_t stride)
{
    if (!h264_pred_host(BD, FFHIP_H264_PRED4x4, MODE, 4, hp_need4(MODE), 4, src, stride, topright, 0, 0, nullptr))
        SHIM_FB(g_fb_pred, pred4x4[MODE], src, topright, stride);
}
template <int BD, int MODE>
static void s_pred8x8l(uint8_t *src, int has_topleft, int has_topright, ptrdiff_t stride)
{
    constexpr unsigned need = hp_need4(MODE);
    /* the corner is also read by the edge filter when has_topleft (PREDICT_8x8_LOAD_LEFT / _TOP) */
    if (!h264_pred_host(BD, FFHIP_H264_PRED8x8L, MODE, 8, need | ((has_topleft && (need & 3)) ? 4u : 0u), 8, src, stride, nullptr, has_topleft,
                        has_topright, nullptr))
        SHIM_FB(g_fb_pred, pred8x8l[MODE], src, has_topleft, has_topright, stride);
}
template <int BD, int MODE>
static void s_pred8x8(uint8_t *src, ptrdiff_t stride)
{
    if (!h264_pred_host(BD, FFHIP_H264_PRED8x8, MODE, 8, hp_need_blk(MODE), MODE == 7 ? 4 : 8, src, stride, nullptr, 0, 0, nullptr))
        SHIM_FB(g_fb_pred, pred8x8[MODE], src, stride);
}
template <int BD, int MODE> /* pred8x8[MODE] at chroma_format_idc >= 2 */
static void s_pred8x16(uint8_t *src, ptrdiff_t stride)
{
    if (!h264_pred_host(BD, FFHIP_H264_PRED8x16, MODE, 8, hp_need_blk(MODE), MODE == 7 ? 4 : 16, src, stride, nullptr, 0, 0, nullptr, 16))
        SHIM_FB((PredFb<BD, 1>::t), pred8x8[MODE], src, stride);
}
template <int BD, int MODE>
static void s_pred16x16(uint8_t *src, ptrdiff_t stride)
{
    if (!h264_pred_host(BD, FFHIP_H264_PRED16x16, MODE, 16, hp_need_blk(MODE), 16, src, stride, nullptr, 0, 0, nullptr))
        SHIM_FB(g_fb_pred, pred16x16[MODE], src, stride);
}
template <int BD, int KIND, int N, int MODE>
static void s_pred_add(uint8_t *pix, int16_t *block, ptrdiff_t stride)
{
    if (!h264_pred_host(BD, KIND, MODE, N, MODE == 0 ? 2u : 1u, N, pix, stride, nullptr, 0, 0, block)) {
        if (KIND == FFHIP_H264_PRED4x4_ADD) SHIM_FB(g_fb_pred, pred4x4_add[MODE], pix, block, stride);
        else                                SHIM_FB(g_fb_pred, pred8x8l_add[MODE], pix, block, stride);
    }
}
template <int BD, int MODE>
static void s_pred8x8l_filter_add(uint8_t *pix, int16_t *block, int has_topleft, int has_topright, ptrdiff_t stride)
{
    if (!h264_pred_host(BD, FFHIP_H264_PRED8x8L_FILTER_ADD, MODE, 8, (MODE == 0 ? 2u : 1u) | (has_topleft ? 4u : 0u), 8, pix, stride, nullptr,
                        has_topleft, MODE == 0 ? has_topright : 0, block))
        SHIM_FB(g_fb_pred, pred8x8l_filter_add[MODE], pix, block, has_topleft, has_topright, stride);
}
/* pred8x8_add / pred16x16_add walk block_offset[] in the C order, each 4x4 seeing what the previous ones wrote
 * (h264pred_template.c:1262-1330) */
template <int BD, int NB, int MODE8x8>
static void s_pred_mb_add(uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    for (int i = 0; i < NB; i++)
        s_pred_add<BD, FFHIP_H264_PRED4x4_ADD, 4, MODE8x8 == 2 ? 0 : 1>(pix + block_offset[i], block + i * 16 * (BD > 8 ? 2 : 1), stride);
}

/* pred8x16_vertical_add / _horizontal_add (h264pred_template.c:1302-1330): blocks 0..3 at block_offset[0..3], 4..7 at [8..11] */
template <int BD, int MODE8x8>
static void s_pred_8x16_add(uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    for (int i = 0; i < 8; i++)
        s_pred_add<BD, FFHIP_H264_PRED4x4_ADD, 4, MODE8x8 == 2 ? 0 : 1>(pix + block_offset[i < 4 ? i : i + 4], block + i * 16 * (BD > 8 ? 2 : 1), stride);
}

#undef g_fb_pred
/* the faces of the H.264 table at the depth */
template <int BD>
static void h264_pred_faces(FFHipH264PredContext &o, int chroma_format_idc)
{
#define HP(M) o.pred4x4[M] = s_pred4x4<BD, M>; o.pred8x8l[M] = s_pred8x8l<BD, M>;
    HP(0) HP(1) HP(2) HP(3) HP(4) HP(5) HP(6) HP(7) HP(8) HP(9) HP(10) HP(11)
#undef HP
    if (chroma_format_idc <= 1) {
#define HP(M) o.pred8x8[M] = s_pred8x8<BD, M>;
        HP(0) HP(1) HP(2) HP(3) HP(4) HP(5) HP(6) HP(7) HP(8) HP(9) HP(10)
#undef HP
    } else {
#define HP(M) o.pred8x8[M] = s_pred8x16<BD, M>;
        HP(0) HP(1) HP(2) HP(3) HP(4) HP(5) HP(6) HP(7) HP(8) HP(9) HP(10)
#undef HP
    }
#define HP(M) o.pred16x16[M] = s_pred16x16<BD, M>;
    HP(0) HP(1) HP(2) HP(3) HP(4) HP(5) HP(6)
#undef HP
    o.pred4x4_add[0] = s_pred_add<BD, FFHIP_H264_PRED4x4_ADD, 4, 0>;   o.pred4x4_add[1] = s_pred_add<BD, FFHIP_H264_PRED4x4_ADD, 4, 1>;
    o.pred8x8l_add[0] = s_pred_add<BD, FFHIP_H264_PRED8x8L_ADD, 8, 0>; o.pred8x8l_add[1] = s_pred_add<BD, FFHIP_H264_PRED8x8L_ADD, 8, 1>;
    o.pred8x8l_filter_add[0] = s_pred8x8l_filter_add<BD, 0>;           o.pred8x8l_filter_add[1] = s_pred8x8l_filter_add<BD, 1>;
    if (chroma_format_idc <= 1) {
        o.pred8x8_add[2] = s_pred_mb_add<BD, 4, 2>;    o.pred8x8_add[1] = s_pred_mb_add<BD, 4, 1>;
    } else {
        o.pred8x8_add[2] = s_pred_8x16_add<BD, 2>;     o.pred8x8_add[1] = s_pred_8x16_add<BD, 1>;
    }
    o.pred16x16_add[2] = s_pred_mb_add<BD, 16, 2>; o.pred16x16_add[1] = s_pred_mb_add<BD, 16, 1>;
}

template <int BD>
static int h264_pred_fill(FFHipH264PredContext *h, int chroma_format_idc)
{
    FFHipH264PredContext o = *h;
    h264_pred_faces<BD>(o, chroma_format_idc);
    if (chroma_format_idc <= 1) {
        fb_snapshot(PredFb<BD>::t, *h, o);
    } else {
        /* the 8 x 16 members' C functions go to the 4:2:2 table, everything else to the shared one */
        FFHipH264PredContext rest = *h;
        memcpy(rest.pred8x8, o.pred8x8, sizeof(rest.pred8x8));
        memcpy(rest.pred8x8_add, o.pred8x8_add, sizeof(rest.pred8x8_add));
        fb_snapshot(PredFb<BD>::t, rest, o);
        fb_snapshot(PredFb<BD, 1>::t, *h, o);
    }
    *h = o;
    return 0;
}


/* ---- the other codecs that share H264PredContext (h264pred.c:540-578; 8 bits, 4:2:0): SVQ3, RV40, VP7, VP8 ---- */
/* the C functions their own forms displaced: one table per codec (CI 0 SVQ3, 1 RV40, 2 VP7, 3 VP8); the members a codec shares with
 * H.264 answer through the 8-bit table PredFb<8>::t like every other 8-bit context's */
template <int CI> struct PredFbCodec { static FFHipH264PredContext t; };
template <int CI> FFHipH264PredContext PredFbCodec<CI>::t;
/* need: bit0 left (LROWS rows), bit1 row above, bit2 corner, bit3 topright[0..3] */
template <int CI, int IDX, int V, unsigned NEED, int LROWS>
static void s_predv4(uint8_t *src, const uint8_t *topright, ptrdiff_t stride)
{
    if (!h264_pred_host(8, FFHIP_H264_PRED_CODEC, V, 4, NEED, LROWS, src, stride, topright, 0, 0, nullptr))
        SHIM_FB(PredFbCodec<CI>::t, pred4x4[IDX], src, topright, stride);
}
/* an H.264 form in a slot of the codec's own (VP7 / VP8: the plain vertical / horizontal at VERT_VP8_PRED / HOR_VP8_PRED) */
template <int CI, int IDX, int MODE>
static void s_predv4_plain(uint8_t *src, const uint8_t *topright, ptrdiff_t stride)
{
    if (!h264_pred_host(8, FFHIP_H264_PRED4x4, MODE, 4, hp_need4(MODE), 4, src, stride, topright, 0, 0, nullptr))
        SHIM_FB(PredFbCodec<CI>::t, pred4x4[IDX], src, topright, stride);
}
template <int CI, int IDX, int V, unsigned NEED>
static void s_predv8(uint8_t *src, ptrdiff_t stride)
{
    if (!h264_pred_host(8, FFHIP_H264_PRED_CODEC, V, 8, NEED, 8, src, stride, nullptr, 0, 0, nullptr))
        SHIM_FB(PredFbCodec<CI>::t, pred8x8[IDX], src, stride);
}
template <int CI, int IDX, int V, unsigned NEED>
static void s_predv16(uint8_t *src, ptrdiff_t stride)
{
    if (!h264_pred_host(8, FFHIP_H264_PRED_CODEC, V, 16, NEED, 16, src, stride, nullptr, 0, 0, nullptr))
        SHIM_FB(PredFbCodec<CI>::t, pred16x16[IDX], src, stride);
}

template <int CI>
static int h264_pred_fill_codec(FFHipH264PredContext *h)
{
    const FFHipH264PredContext in = *h;
    FFHipH264PredContext o = in;
    h264_pred_faces<8>(o, 1);
    FFHipH264PredContext shared = o; /* what the codec takes from the H.264 table */
    if (CI == 0) { /* SVQ3 */
        o.pred4x4[3] = s_predv4<CI, 3, FFHIP_H264_PREDV_DL_SVQ3, 3u, 4>;
        o.pred16x16[3] = s_predv16<CI, 3, FFHIP_H264_PREDV16_PLANE_SVQ3, 7u>;
    } else if (CI == 1) { /* RV40 */
        o.pred4x4[3] = s_predv4<CI, 3, FFHIP_H264_PREDV_DL_RV40, 11u, 8>;
        o.pred4x4[7] = s_predv4<CI, 7, FFHIP_H264_PREDV_VL_RV40, 11u, 8>;
        o.pred4x4[8] = s_predv4<CI, 8, FFHIP_H264_PREDV_HU_RV40, 11u, 8>;
        o.pred4x4[12] = s_predv4<CI, 12, FFHIP_H264_PREDV_DL_RV40_NODOWN, 11u, 4>;
        o.pred4x4[13] = s_predv4<CI, 13, FFHIP_H264_PREDV_HU_RV40_NODOWN, 11u, 4>;
        o.pred4x4[14] = s_predv4<CI, 14, FFHIP_H264_PREDV_VL_RV40_NODOWN, 11u, 4>;
        o.pred16x16[3] = s_predv16<CI, 3, FFHIP_H264_PREDV16_PLANE_RV40, 7u>;
    } else { /* VP7 / VP8 */
        o.pred4x4[0] = s_predv4<CI, 0, FFHIP_H264_PREDV_VERT_VP8, 14u, 4>;
        o.pred4x4[1] = s_predv4<CI, 1, FFHIP_H264_PREDV_HOR_VP8, 5u, 4>;
        o.pred4x4[7] = s_predv4<CI, 7, FFHIP_H264_PREDV_VL_VP8, 10u, 4>;
        o.pred4x4[9] = s_predv4<CI, 9, FFHIP_H264_PREDV_TM_VP8, 7u, 4>;
        o.pred4x4[10] = s_predv4_plain<CI, 10, 0>;  /* VERT_VP8_PRED: the plain vertical / horizontal forms (h264pred.c:563,566) */
        o.pred4x4[14] = s_predv4_plain<CI, 14, 1>;  /* HOR_VP8_PRED */
        o.pred4x4[12] = s_predv4<CI, 12, FFHIP_H264_PREDV_127_DC, 0u, 4>;
        o.pred4x4[13] = s_predv4<CI, 13, FFHIP_H264_PREDV_129_DC, 0u, 4>;
        o.pred8x8[3] = s_predv8<CI, 3, FFHIP_H264_PREDV8_TM_VP8, 7u>;
        o.pred8x8[7] = s_predv8<CI, 7, FFHIP_H264_PREDV8_127_DC, 0u>;
        o.pred8x8[8] = s_predv8<CI, 8, FFHIP_H264_PREDV8_129_DC, 0u>;
        o.pred16x16[3] = s_predv16<CI, 3, FFHIP_H264_PREDV16_TM_VP8, 7u>;
        o.pred16x16[7] = s_predv16<CI, 7, FFHIP_H264_PREDV16_127_DC, 0u>;
        o.pred16x16[8] = s_predv16<CI, 8, FFHIP_H264_PREDV16_129_DC, 0u>;
        if (CI == 3)
            o.pred4x4[11] = in.pred4x4[11]; /* VP8: DC_128_PRED is not set (h264pred.c:466) */
    }
    if (CI >= 1) { /* RV40 / VP7 / VP8: the rv40 DCs, and no "mad cow" forms (h264pred.c:489-513) */
        o.pred8x8[0] = s_predv8<CI, 0, FFHIP_H264_PREDV8_DC_RV40, 3u>;
        o.pred8x8[4] = s_predv8<CI, 4, FFHIP_H264_PREDV8_LEFT_DC_RV40, 1u>;
        o.pred8x8[5] = s_predv8<CI, 5, FFHIP_H264_PREDV8_TOP_DC_RV40, 2u>;
        for (int m = 7; m <= 10; m++)
            if (!(CI >= 2 && m <= 8))
                o.pred8x8[m] = in.pred8x8[m];
    }
    /* the displaced C functions: a member whose face is the H.264 table's goes to that table, the codec's own forms to the codec's */
    FFHipH264PredContext in_shared = in, in_own = in;
    void **po = reinterpret_cast<void **>(&o), **ps = reinterpret_cast<void **>(&shared);
    void **pis = reinterpret_cast<void **>(&in_shared), **pio = reinterpret_cast<void **>(&in_own);
    for (size_t i = 0; i < sizeof(o) / sizeof(void *); i++) {
        if (po[i] == ps[i])
            pio[i] = po[i]; /* shared face: nothing for the codec's table */
        else
            pis[i] = po[i]; /* own form (or left alone): nothing for the shared table */
    }
    fb_snapshot(PredFb<8>::t, in_shared, o);
    fb_snapshot(PredFbCodec<CI>::t, in_own, o);
    *h = o;
    return 0;
}

extern "C" int ff_h264_pred_init_hip(FFHipH264PredContext *h, int codec_id, int bit_depth, int chroma_format_idc)
{
    /* AV_CODEC_ID_H264 at the depths it defines; chroma_format_idc >= 2 switches pred8x8 to the 8 x 16 forms (h264pred.c:478-535) */
    if (!h || chroma_format_idc < 0 || chroma_format_idc > 3)
        return FFHIP_EINVAL;
    if (codec_id == FFHIP_CODEC_ID_SVQ3 || codec_id == FFHIP_CODEC_ID_RV40 || codec_id == FFHIP_CODEC_ID_VP7 || codec_id == FFHIP_CODEC_ID_VP8) {
        if (bit_depth != 8 || chroma_format_idc > 1)
            return FFHIP_EINVAL; /* these codecs are 8 bits, 4:2:0 */
        if (!ffhip_have_device())
            return FFHIP_ENOSYS;
        return codec_id == FFHIP_CODEC_ID_SVQ3 ? h264_pred_fill_codec<0>(h) : codec_id == FFHIP_CODEC_ID_RV40 ? h264_pred_fill_codec<1>(h)
             : codec_id == FFHIP_CODEC_ID_VP7 ? h264_pred_fill_codec<2>(h) : h264_pred_fill_codec<3>(h);
    }
    if (codec_id != FFHIP_CODEC_ID_H264)
        return FFHIP_EINVAL;
    if (bit_depth != 8 && bit_depth != 9 && bit_depth != 10 && bit_depth != 12 && bit_depth != 14)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    switch (bit_depth) {
    case 8:  return h264_pred_fill<8>(h, chroma_format_idc);
    case 9:  return h264_pred_fill<9>(h, chroma_format_idc);
    case 10: return h264_pred_fill<10>(h, chroma_format_idc);
    case 12: return h264_pred_fill<12>(h, chroma_format_idc);
    default: return h264_pred_fill<14>(h, chroma_format_idc);
    }
}
