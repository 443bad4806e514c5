/*
 * shims_h264_hbd.hip — the signature-exact, HOST-pointer faces of the H.264 tables above 8 bits (9 / 10 / 12 / 14) and of the members the
 * 8-bit faces of shims.hip do not cover at any depth: the MBAFF loop filters and the 4:2:2 chroma forms.
 *
 * What ff_h264dsp_init() / ff_h264qpel_init() / ff_h264chroma_init() select per bit depth and chroma format (libavcodec/h264dsp.c:
 * 70-153, h264qpel.c:87-103, h264chroma.c:38-52).  The depth is baked into the installed function, as the reference's per-BIT_DEPTH
 * instantiations are: one instantiation of every face per depth (templates on BD).  Same contract as shims.hip: one call = one
 * launch of the batched kernel with n = 1 through the device scratch arena; results are committed to host memory only after the
 * device reported success; a call that cannot run on the device is answered by the C function the init displaced.
 * tests/checkasm/h264dsp.c, h264qpel.c and h264chroma.c exercise exactly these pointers at 8 / 9 / 10 / 12 / 14 bits.
 */
#include <string.h>

#include "kernels/common.h"
#include "kernels/h264_kernels.h"
#include "kernels/shim_rect.h"

namespace {

/* The C functions an init displaced, per bit depth — and, for the six members ff_h264dsp_init() picks by chroma_format_idc
 * (h264dsp.c:113-132: the horizontal chroma filters, idct_add8, chroma_dc_dequant_idct), per chroma format: a 4:2:0 and a 4:2:2 decoder
 * of one depth in one process each fall back to their own C function (round 4; one table per depth before). */
template <int BD> struct Fb {
    static FFHipH264DSPContext dsp;
    static FFHipH264DSPContext dsp422;
    static FFHipH264QpelContext qpel;
    static FFHipH264ChromaContext chroma;
    static FFHipH264WeightContext weight;
};
template <int BD> FFHipH264DSPContext Fb<BD>::dsp;
template <int BD> FFHipH264DSPContext Fb<BD>::dsp422;
template <int BD> FFHipH264QpelContext Fb<BD>::qpel;
template <int BD> FFHipH264ChromaContext Fb<BD>::chroma;
template <int BD> FFHipH264WeightContext Fb<BD>::weight;

constexpr int PX(int bd) { return bd > 8 ? 2 : 1; }

/* ---- single blocks ---- */
template <int BD>
bool idct_single(int kind, uint8_t *dst, int16_t *block, ptrdiff_t stride)
{
    const int size = (kind & 1) ? 8 : 4, cbytes = size * size * 2 * PX(BD);
    Rect d = { dst, stride, 0, size - 1, 0, size * PX(BD) - 1, nullptr };
    Arena A(rect_bytes(d) + 512 + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    int16_t *dblk = (int16_t *)buf;
    int32_t *doff = (int32_t *)(buf + 256);
    const int32_t zero = 0;
    if (!rect_up(d, buf + 512) || hipMemcpy(dblk, block, cbytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(doff, &zero, 4, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_idct_add_bd(BD, kind, d.dev, DP, doff, dblk, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, d, 0, size - 1, 0, size * PX(BD) - 1);
    memcpy(block, A.host(dblk), cbytes);
    return true;
}

/* ---- macroblock dispatchers: which 0 idct_add16, 1 idct8_add4, 2 idct_add16intra (one plane), 3 idct_add8, 4 idct_add8_422 (two planes) ---- */
template <int BD>
bool idct_mb(int which, uint8_t *const *planes, const int *blockoffset, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc)
{
    if (stride <= 0 || stride > (1 << 16))
        return false;
    const int bs = which == 1 ? 8 : 4, npl = which >= 3 ? 2 : 1;
    const int ncoef_bytes = (which >= 3 ? 768 : 256) * 2 * PX(BD), nnz_bytes = which >= 3 ? 120 : 40;
    int slots[2][16], nslot[2] = { 0, 0 };
    if (which == 1) { for (int i = 0; i < 16; i += 4) slots[0][nslot[0]++] = i; }
    else if (which < 3) { for (int i = 0; i < 16; i++) slots[0][nslot[0]++] = i; }
    else
        for (int j = 0; j < 2; j++)
            for (int r = 0; r < (which == 3 ? 4 : 8); r++)
                slots[j][nslot[j]++] = 16 * (j + 1) + r + (r >= 4 ? 4 : 0);
    int lo[2], hi[2];
    size_t span[2] = { 0, 0 }, sp[2] = { 0, 0 };
    for (int j = 0; j < npl; j++) {
        lo[j] = hi[j] = blockoffset[slots[j][0]];
        for (int k = 0; k < nslot[j]; k++) {
            const int o = blockoffset[slots[j][k]];
            if (o < lo[j]) lo[j] = o;
            if (o > hi[j]) hi[j] = o;
        }
        span[j] = (size_t)(hi[j] - lo[j]) + (size_t)(bs - 1) * stride + (size_t)bs * PX(BD);
        sp[j] = (span[j] + 63) & ~(size_t)63;
    }
    Arena A(3072 + 256 + 128 + 64 + sp[0] + sp[1] + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    int16_t *dblk = (int16_t *)buf;              /* <= 3072 B */
    int32_t *dbo = (int32_t *)(buf + 3072);      /* 192 B    */
    uint8_t *dnn = buf + 3072 + 256;             /* 120 B    */
    int32_t *dmb = (int32_t *)(buf + 3072 + 256 + 128);
    uint8_t *dpix[2] = { buf + 3072 + 256 + 128 + 64, buf + 3072 + 256 + 128 + 64 + sp[0] };
    const int32_t mboff = 0;
    int32_t bo[48] = { 0 };
    for (int j = 0; j < npl; j++)
        for (int k = 0; k < nslot[j]; k++)
            bo[slots[j][k]] = blockoffset[slots[j][k]] - lo[j];
    for (int j = 0; j < npl; j++)
        if (hipMemcpy(dpix[j], planes[j] + lo[j], span[j], hipMemcpyHostToDevice) != hipSuccess)
            return false;
    if (hipMemcpy(dblk, block, ncoef_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dbo, bo, 192, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dnn, nnzc, nnz_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dmb, &mboff, 4, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_idct_mb_bd(BD, which, dpix[0], dpix[1], stride, dmb, dbo, dblk, dnn, 1, 0) < 0 || !A.down())
        return false;
    for (int j = 0; j < npl; j++)
        for (int k = 0; k < nslot[j]; k++) /* only this macroblock's own blocks travel back */
            commit2d(A, planes[j] + blockoffset[slots[j][k]], stride, dpix[j] + bo[slots[j][k]], stride, (size_t)bs * PX(BD), bs);
    if (which >= 3)
        memcpy((uint8_t *)block + 256 * 2 * PX(BD), A.host((uint8_t *)dblk + 256 * 2 * PX(BD)), 512 * 2 * PX(BD)); /* the chroma planes' coefficients */
    else
        memcpy(block, A.host(dblk), ncoef_bytes);
    return true;
}

/* ---- DC transforms: which 0 luma, 1 chroma 4:2:0, 2 chroma 4:2:2 ---- */
template <int BD>
bool dc_dequant(int which, int16_t *output, int16_t *input, int qmul)
{
    const int cb = 2 * PX(BD); /* bytes per coefficient */
    Arena A(1024 + 128 + 64 + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    int16_t *dout = (int16_t *)buf, *din = (int16_t *)(buf + 1024);
    int32_t *dq = (int32_t *)(buf + 1152), *doff = (int32_t *)(buf + 1184);
    const int32_t q = qmul, zero = 0;
    if (hipMemcpy(dq, &q, 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(doff, &zero, 4, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    const int nout = which == 0 ? 256 : which == 1 ? 64 : 128; /* coefficients the function's writes span */
    if (which == 0 && hipMemcpy(din, input, 16 * cb, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (hipMemcpy(dout, output, (size_t)nout * cb, hipMemcpyHostToDevice) != hipSuccess ||
        ffhip_launch_h264_dc_dequant_bd(BD, which, dout, 256, din, 16, doff, dq, 1, 0) < 0 || !A.down())
        return false;
    const uint8_t *h = A.host(dout);
    /* the DC positions are all the function writes: 16 (luma), 4 (4:2:0: blocks 0..3), 8 (4:2:2: blocks 0..7) */
    const int nblk = which == 0 ? 16 : which == 1 ? 4 : 8;
    for (int i = 0; i < nblk; i++)
        memcpy((uint8_t *)output + (size_t)16 * i * cb, h + (size_t)16 * i * cb, cb);
    return true;
}

/* ---- loop filters: kind bit 0 h_, bit 1 chroma, bit 2 intra; inner = lines per tc0 entry ---- */
template <int BD>
bool lf_single(int kind, int inner, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, const int8_t *tc0)
{
    const bool chroma = kind & 2, vert_edge = kind & 1;
    const int along = 4 * inner, across = chroma ? 2 : 4; /* lines; samples read each side of the edge */
    const int px = PX(BD);
    Rect r = { pix, stride, vert_edge ? 0 : -across, vert_edge ? along - 1 : across - 1,
               vert_edge ? -across * px : 0, vert_edge ? across * px - 1 : along * px - 1, nullptr };
    Arena A(rect_bytes(r) + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    FFHipH264Edge e;
    memset(&e, 0, sizeof(e));
    e.kind = (uint8_t)kind;
    e.pad = (uint8_t)inner;
    const int32_t ab[2] = { alpha, beta }; /* the caller's ints as they are: the C function takes any (it scales them by the depth itself) */
    if (tc0)
        memcpy(e.tc0, tc0, 4);
    if (!rect_up(r, buf + 64) || hipMemcpy(buf, &e, sizeof(e), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(buf + 16, ab, sizeof(ab), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_loop_filter_bd(BD, r.dev, DP, (const FFHipH264Edge *)buf, 1, 0, (const int32_t *)(buf + 16)) < 0 || !A.down())
        return false;
    rect_commit(A, r, r.r0, r.r1, r.c0, r.c1);
    return true;
}

/* ---- luma qpel ---- */
template <int BD>
bool qpel_single(int avg, int size_idx, int mcxy, uint8_t *dst, const uint8_t *src, ptrdiff_t stride)
{
    const int n = 16 >> size_idx, px = PX(BD);
    Rect d = { dst, stride, 0, n - 1, 0, n * px - 1, nullptr };
    const bool fx = mcxy & 3, fy = mcxy >> 2; /* the 6-tap margin exists on an axis only when that axis is filtered */
    Rect s = { const_cast<uint8_t *>(src), stride, fy ? -2 : 0, fy ? n + 2 : n - 1, fx ? -2 * px : 0, fx ? (n + 3) * px - 1 : n * px - 1, nullptr };
    Rect full = s;
    full.r0 = -2; full.r1 = n + 2; full.c0 = -2 * px; full.c1 = (n + 3) * px - 1;
    Arena A(rect_bytes(d) + rect_bytes(full) + 64);
    if (!A.ok)
        return false;
    uint8_t *buf = A.buf;
    if (!rect_up(d, buf + 64) || !rect_up(s, buf + 64 + rect_bytes(d) + (size_t)(s.r0 + 2) * DP + (s.c0 + 2 * px)))
        return false;
    FFHipQpelBlock b;
    memset(&b, 0, sizeof(b));
    b.dst_offset = (int32_t)(d.dev - buf); b.src_offset = (int32_t)(s.dev - buf);
    b.mcxy = (uint8_t)mcxy; b.size_idx = (uint8_t)size_idx; b.avg = (uint8_t)avg;
    if (hipMemcpy(buf, &b, sizeof(b), hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_h264_qpel_bd(BD, buf, buf, DP, (const FFHipQpelBlock *)buf, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, d, 0, n - 1, 0, n * px - 1);
    return true;
}

/* ---- chroma MC: the template reads a neighbour column / row only when its weight is non-zero ---- */
template <int BD>
bool chroma_single(int avg, int w_idx, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h, int x, int y)
{
    const int w = 8 >> w_idx, px = PX(BD);
    if (h <= 0 || h > 16 || (unsigned)x > 7u || (unsigned)y > 7u)
        return false;
    Rect d = { dst, stride, 0, h - 1, 0, w * px - 1, nullptr };
    Rect s = { const_cast<uint8_t *>(src), stride, 0, h - 1 + (y ? 1 : 0), 0, (w + (x ? 1 : 0)) * px - 1, nullptr };
    Rect full = s;
    full.r1 = h; full.c1 = (w + 1) * px - 1;
    Arena A(rect_bytes(d) + rect_bytes(full) + 64);
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
    if (ffhip_launch_h264_chroma_mc_bd(BD, buf, buf, DP, (const FFHipChromaBlock *)buf, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, d, 0, h - 1, 0, w * px - 1);
    return true;
}

template <int BD>
bool weight_single(int bi, int w_idx, uint8_t *dst, uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd, int weights, int offset)
{
    const int w = 16 >> w_idx, px = PX(BD);
    if (height <= 0 || height > 16 || log2_denom < 0 || log2_denom > 7)
        return false;
    Rect d = { dst, stride, 0, height - 1, 0, w * px - 1, nullptr };
    Rect s = { bi ? src : dst, stride, 0, height - 1, 0, w * px - 1, nullptr };
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
    if (ffhip_launch_h264_weight_bd(BD, buf, buf, DP, (const FFHipWeightBlock *)buf, 1, 0) < 0 || !A.down())
        return false;
    rect_commit(A, d, 0, height - 1, 0, w * px - 1);
    return true;
}

/* ---- the faces, one set per depth ---- */
template <int BD> struct F {
    typedef Fb<BD> B;
#define IDCT1(name, kind) static void name(uint8_t *d, int16_t *b, ptrdiff_t s) { if (!idct_single<BD>(kind, d, b, s)) SHIM_FB(B::dsp, name, d, b, s); }
    IDCT1(idct_add, FFHIP_H264_IDCT4) IDCT1(idct8_add, FFHIP_H264_IDCT8) IDCT1(idct_dc_add, FFHIP_H264_IDCT4_DC) IDCT1(idct8_dc_add, FFHIP_H264_IDCT8_DC)
    IDCT1(add_pixels4_clear, FFHIP_H264_ADD_PIXELS4_CLEAR) IDCT1(add_pixels8_clear, FFHIP_H264_ADD_PIXELS8_CLEAR)
#undef IDCT1
#define IDCTM(name, which) static void name(uint8_t *d, const int *bo, int16_t *b, ptrdiff_t s, const uint8_t n[5 * 8]) \
    { uint8_t *pl[2] = { d, nullptr }; if (!idct_mb<BD>(which, pl, bo, b, s, n)) SHIM_FB(B::dsp, name, d, bo, b, s, n); }
    IDCTM(idct_add16, 0) IDCTM(idct8_add4, 1) IDCTM(idct_add16intra, 2)
#undef IDCTM
    static void idct_add8(uint8_t **d, const int *bo, int16_t *b, ptrdiff_t s, const uint8_t n[15 * 8])
    { if (!idct_mb<BD>(3, d, bo, b, s, n)) SHIM_FB(B::dsp, idct_add8, d, bo, b, s, n); }
    static void idct_add8_422(uint8_t **d, const int *bo, int16_t *b, ptrdiff_t s, const uint8_t n[15 * 8])
    { if (!idct_mb<BD>(4, d, bo, b, s, n)) SHIM_FB(B::dsp422, idct_add8, d, bo, b, s, n); }
    static void luma_dc_dequant_idct(int16_t *o, int16_t *i, int q) { if (!dc_dequant<BD>(0, o, i, q)) SHIM_FB(B::dsp, luma_dc_dequant_idct, o, i, q); }
    static void chroma_dc_dequant_idct(int16_t *b, int q) { if (!dc_dequant<BD>(1, b, nullptr, q)) SHIM_FB(B::dsp, chroma_dc_dequant_idct, b, q); }
    static void chroma422_dc_dequant_idct(int16_t *b, int q) { if (!dc_dequant<BD>(2, b, nullptr, q)) SHIM_FB(B::dsp422, chroma_dc_dequant_idct, b, q); }
#define LFT(name, member, kind, inner, tab) static void name(uint8_t *p, ptrdiff_t s, int a, int b, int8_t *t) \
    { if (!lf_single<BD>(kind, inner, p, s, a, b, t)) SHIM_FB(B::tab, member, p, s, a, b, t); }
#define LFIT(name, member, kind, inner, tab) static void name(uint8_t *p, ptrdiff_t s, int a, int b) \
    { if (!lf_single<BD>(kind, inner, p, s, a, b, nullptr)) SHIM_FB(B::tab, member, p, s, a, b); }
#define LF(name, member, kind, inner) LFT(name, member, kind, inner, dsp)
#define LFI(name, member, kind, inner) LFIT(name, member, kind, inner, dsp)
    LF(v_luma, v_loop_filter_luma, 0, 4) LF(h_luma, h_loop_filter_luma, 1, 4) LF(h_luma_mbaff, h_loop_filter_luma_mbaff, 1, 2)
    LF(v_chroma, v_loop_filter_chroma, 2, 2) LF(h_chroma, h_loop_filter_chroma, 3, 2) LF(h_chroma_mbaff, h_loop_filter_chroma_mbaff, 3, 1)
    LFT(h_chroma422, h_loop_filter_chroma, 3, 4, dsp422) LFT(h_chroma422_mbaff, h_loop_filter_chroma_mbaff, 3, 2, dsp422)
    LFI(v_luma_i, v_loop_filter_luma_intra, 4, 4) LFI(h_luma_i, h_loop_filter_luma_intra, 5, 4) LFI(h_luma_mbaff_i, h_loop_filter_luma_mbaff_intra, 5, 2)
    LFI(v_chroma_i, v_loop_filter_chroma_intra, 6, 2) LFI(h_chroma_i, h_loop_filter_chroma_intra, 7, 2)
    LFI(h_chroma_mbaff_i, h_loop_filter_chroma_mbaff_intra, 7, 1) LFIT(h_chroma422_i, h_loop_filter_chroma_intra, 7, 4, dsp422)
    LFIT(h_chroma422_mbaff_i, h_loop_filter_chroma_mbaff_intra, 7, 2, dsp422)
#undef LF
#undef LFI
#undef LFT
#undef LFIT
    template <int AVG, int IDX, int MC> static void qpel(uint8_t *d, const uint8_t *s, ptrdiff_t st)
    {
        if (!qpel_single<BD>(AVG, IDX, MC, d, s, st)) {
            if (AVG) SHIM_FB(B::qpel, avg_h264_qpel_pixels_tab[IDX][MC], d, s, st);
            else     SHIM_FB(B::qpel, put_h264_qpel_pixels_tab[IDX][MC], d, s, st);
        }
    }
    template <int AVG, int IDX> static void chroma(uint8_t *d, const uint8_t *s, ptrdiff_t st, int h, int x, int y)
    {
        if (!chroma_single<BD>(AVG, IDX, d, s, st, h, x, y)) {
            if (AVG) SHIM_FB(B::chroma, avg_h264_chroma_pixels_tab[IDX], d, s, st, h, x, y);
            else     SHIM_FB(B::chroma, put_h264_chroma_pixels_tab[IDX], d, s, st, h, x, y);
        }
    }
    template <int IDX> static void weight(uint8_t *b, ptrdiff_t st, int h, int ld, int w, int o)
    { if (!weight_single<BD>(0, IDX, b, nullptr, st, h, ld, w, 0, o)) SHIM_FB(B::weight, weight_pixels_tab[IDX], b, st, h, ld, w, o); }
    template <int IDX> static void biweight(uint8_t *d, uint8_t *s, ptrdiff_t st, int h, int ld, int wd, int ws, int o)
    { if (!weight_single<BD>(1, IDX, d, s, st, h, ld, wd, ws, o)) SHIM_FB(B::weight, biweight_pixels_tab[IDX], d, s, st, h, ld, wd, ws, o); }
};

template <int BD, int AVG, int IDX>
void fill_qpel_row(ffhip_qpel_mc_func (&row)[16])
{
    row[0] = F<BD>::template qpel<AVG, IDX, 0>;   row[1] = F<BD>::template qpel<AVG, IDX, 1>;   row[2] = F<BD>::template qpel<AVG, IDX, 2>;
    row[3] = F<BD>::template qpel<AVG, IDX, 3>;   row[4] = F<BD>::template qpel<AVG, IDX, 4>;   row[5] = F<BD>::template qpel<AVG, IDX, 5>;
    row[6] = F<BD>::template qpel<AVG, IDX, 6>;   row[7] = F<BD>::template qpel<AVG, IDX, 7>;   row[8] = F<BD>::template qpel<AVG, IDX, 8>;
    row[9] = F<BD>::template qpel<AVG, IDX, 9>;   row[10] = F<BD>::template qpel<AVG, IDX, 10>; row[11] = F<BD>::template qpel<AVG, IDX, 11>;
    row[12] = F<BD>::template qpel<AVG, IDX, 12>; row[13] = F<BD>::template qpel<AVG, IDX, 13>; row[14] = F<BD>::template qpel<AVG, IDX, 14>;
    row[15] = F<BD>::template qpel<AVG, IDX, 15>;
}

/* all: every member at this depth; else only what the 8-bit faces of shims.hip leave out (MBAFF, and for 4:2:2 the chroma forms) */
template <int BD>
void fill_dsp(FFHipH264DSPContext &o, int cfi, bool all)
{
    typedef F<BD> X;
    if (all) {
        o.v_loop_filter_luma = X::v_luma;               o.h_loop_filter_luma = X::h_luma;
        o.v_loop_filter_luma_intra = X::v_luma_i;       o.h_loop_filter_luma_intra = X::h_luma_i;
        o.v_loop_filter_chroma = X::v_chroma;           o.v_loop_filter_chroma_intra = X::v_chroma_i;
        o.idct_add = X::idct_add;                       o.idct8_add = X::idct8_add;
        o.idct_dc_add = X::idct_dc_add;                 o.idct8_dc_add = X::idct8_dc_add;
        o.idct_add16 = X::idct_add16;                   o.idct8_add4 = X::idct8_add4;
        o.idct_add16intra = X::idct_add16intra;         o.luma_dc_dequant_idct = X::luma_dc_dequant_idct;
        o.add_pixels4_clear = X::add_pixels4_clear;     o.add_pixels8_clear = X::add_pixels8_clear;
    }
    o.h_loop_filter_luma_mbaff = X::h_luma_mbaff;       o.h_loop_filter_luma_mbaff_intra = X::h_luma_mbaff_i;
    if (cfi <= 1) {
        if (all) {
            o.h_loop_filter_chroma = X::h_chroma;       o.h_loop_filter_chroma_intra = X::h_chroma_i;
            o.idct_add8 = X::idct_add8;                 o.chroma_dc_dequant_idct = X::chroma_dc_dequant_idct;
        }
        o.h_loop_filter_chroma_mbaff = X::h_chroma_mbaff; o.h_loop_filter_chroma_mbaff_intra = X::h_chroma_mbaff_i;
    } else {
        o.h_loop_filter_chroma = X::h_chroma422;        o.h_loop_filter_chroma_intra = X::h_chroma422_i;
        o.h_loop_filter_chroma_mbaff = X::h_chroma422_mbaff; o.h_loop_filter_chroma_mbaff_intra = X::h_chroma422_mbaff_i;
        o.idct_add8 = X::idct_add8_422;                 o.chroma_dc_dequant_idct = X::chroma422_dc_dequant_idct;
    }
}

template <int BD>
int init_dsp(FFHipH264DSPContext *c, FFHipH264DSPContext &o, int cfi, bool all)
{
    fill_dsp<BD>(o, cfi, all);
    if (cfi <= 1) {
        fb_snapshot(Fb<BD>::dsp, *c, o);
    } else {
        /* the six members picked by the chroma format go to the 4:2:2 table, the others (the same C functions either way) to the
         * depth's common one: equal words are skipped by fb_snapshot */
        FFHipH264DSPContext in = *c, ours = o;
        fb_snapshot(Fb<BD>::dsp422, in, ours);
        ours.h_loop_filter_chroma = in.h_loop_filter_chroma;               ours.h_loop_filter_chroma_intra = in.h_loop_filter_chroma_intra;
        ours.h_loop_filter_chroma_mbaff = in.h_loop_filter_chroma_mbaff;   ours.h_loop_filter_chroma_mbaff_intra = in.h_loop_filter_chroma_mbaff_intra;
        ours.idct_add8 = in.idct_add8;                                     ours.chroma_dc_dequant_idct = in.chroma_dc_dequant_idct;
        fb_snapshot(Fb<BD>::dsp, in, ours);
    }
    return 0;
}

template <int BD>
int init_qpel(FFHipH264QpelContext *c)
{
    FFHipH264QpelContext o = *c;
    fill_qpel_row<BD, 0, 0>(o.put_h264_qpel_pixels_tab[0]); fill_qpel_row<BD, 0, 1>(o.put_h264_qpel_pixels_tab[1]);
    fill_qpel_row<BD, 0, 2>(o.put_h264_qpel_pixels_tab[2]); fill_qpel_row<BD, 1, 0>(o.avg_h264_qpel_pixels_tab[0]);
    fill_qpel_row<BD, 1, 1>(o.avg_h264_qpel_pixels_tab[1]); fill_qpel_row<BD, 1, 2>(o.avg_h264_qpel_pixels_tab[2]);
    fb_snapshot(Fb<BD>::qpel, *c, o);
    *c = o;
    return 0;
}

template <int BD>
int init_chroma(FFHipH264ChromaContext *c)
{
    FFHipH264ChromaContext o = *c; /* slot 3 (one sample wide, h264chroma.c:47) stays with the caller's C function */
    o.put_h264_chroma_pixels_tab[0] = F<BD>::template chroma<0, 0>; o.put_h264_chroma_pixels_tab[1] = F<BD>::template chroma<0, 1>;
    o.put_h264_chroma_pixels_tab[2] = F<BD>::template chroma<0, 2>; o.avg_h264_chroma_pixels_tab[0] = F<BD>::template chroma<1, 0>;
    o.avg_h264_chroma_pixels_tab[1] = F<BD>::template chroma<1, 1>; o.avg_h264_chroma_pixels_tab[2] = F<BD>::template chroma<1, 2>;
    fb_snapshot(Fb<BD>::chroma, *c, o);
    *c = o;
    return 0;
}

template <int BD>
int init_weight(FFHipH264WeightContext *c)
{
    FFHipH264WeightContext o = *c;
    o.weight_pixels_tab[0] = F<BD>::template weight<0>; o.weight_pixels_tab[1] = F<BD>::template weight<1>;
    o.weight_pixels_tab[2] = F<BD>::template weight<2>; o.weight_pixels_tab[3] = F<BD>::template weight<3>;
    o.biweight_pixels_tab[0] = F<BD>::template biweight<0>; o.biweight_pixels_tab[1] = F<BD>::template biweight<1>;
    o.biweight_pixels_tab[2] = F<BD>::template biweight<2>; o.biweight_pixels_tab[3] = F<BD>::template biweight<3>;
    fb_snapshot(Fb<BD>::weight, *c, o);
    *c = o;
    return 0;
}

} // namespace

#define BY_DEPTH(bd, CALL8, CALL)                                   \
    switch (bd) {                                                   \
    case 8:  return CALL8;                                          \
    case 9:  { constexpr int D = 9;  return CALL; }                 \
    case 10: { constexpr int D = 10; return CALL; }                 \
    case 12: { constexpr int D = 12; return CALL; }                 \
    case 14: { constexpr int D = 14; return CALL; }                 \
    default: return FFHIP_EINVAL;                                   \
    }

/* shims.hip's ff_h264dsp_init_hip() calls this: at 8 bits to add the MBAFF / 4:2:2 members to its own faces, above for everything */
int ffhip_h264dsp_fill_generic(FFHipH264DSPContext *c, FFHipH264DSPContext *o, int bit_depth, int chroma_format_idc)
{
    BY_DEPTH(bit_depth, init_dsp<8>(c, *o, chroma_format_idc, false), init_dsp<D>(c, *o, chroma_format_idc, true))
}
int ffhip_h264qpel_init_generic(FFHipH264QpelContext *c, int bit_depth) { BY_DEPTH(bit_depth, FFHIP_EINVAL, init_qpel<D>(c)) }
int ffhip_h264chroma_init_generic(FFHipH264ChromaContext *c, int bit_depth) { BY_DEPTH(bit_depth, FFHIP_EINVAL, init_chroma<D>(c)) }
int ffhip_h264weight_init_generic(FFHipH264WeightContext *c, int bit_depth) { BY_DEPTH(bit_depth, FFHIP_EINVAL, init_weight<D>(c)) }
