/*
 * ffo.h — liboracle.so: plain-C CPU restatement of the reference's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker; never by the product.  Parity status: PINNED — every function
 * is checked bit-exact (float: stated tolerance) against the real reference compiled from
 * /root/reference (oracle/_ref/libffref.so) in tests/test_oracle_vs_ref.py, and against the
 * committed fixtures in tests/golden/ that were generated from it.
 */
#ifndef FFO_H
#define FFO_H
#include <stddef.h>
#include <stdint.h>

/* AVPixelFormat values (libavutil/pixfmt.h) */
#define FFO_PIX_FMT_YUV420P 0
#define FFO_PIX_FMT_RGB24   2
#define FFO_PIX_FMT_YUV422P 4
#define FFO_PIX_FMT_YUV444P 5
#define FFO_PIX_FMT_BGR24   3
#define FFO_PIX_FMT_NV12    23
#define FFO_PIX_FMT_ARGB    25
#define FFO_PIX_FMT_RGBA    26
#define FFO_PIX_FMT_ABGR    27
#define FFO_PIX_FMT_BGRA    28
#define FFO_PIX_FMT_NV21    24

/* ---- swscale (ffo_sws.c) ---- */
typedef struct FfoYuv2RgbCoeffs {
    int64_t cy, oy, crv, cbu, cgu, cgv; /* after the /cy scaling, yuv2rgb.c:793-797 */
    int     yoffs;
} FfoYuv2RgbCoeffs;
typedef struct FfoYuv2RgbLuts {
    uint8_t ramp[2048];
    int     rV[1280], gU[1280], gV[1280], bU[1280];
} FfoYuv2RgbLuts;
typedef struct FfoSwsFilter {
    const int16_t *filter;
    const int32_t *pos;
    int size, n;
} FfoSwsFilter;
typedef struct FfoSwsTables {
    int srcW, srcH, srcFormat, dstW, dstH, dstFormat, flags;
    FfoSwsFilter hLum, hChr, vLum, vChr;
    FfoYuv2RgbCoeffs k;
    /* range conversion between YUV formats: c->opts.src_range / dst_range and the constants of c->lum / chrConvertRange
     * (init_range_convert_constants, libswscale/swscale.c:591-624); all zero: none */
    int src_range, dst_range;
    uint32_t lum_rc_coeff, chr_rc_coeff;
    int64_t lum_rc_offset, chr_rc_offset;
    /* SWS_FULL_CHR_H_INT on a packed RGB target (utils.c:1270-1290: asked for, or forced by an odd width or a 4:4:4 source): chroma at
     * full horizontal resolution (hChr.n == dstW) and the yuv2rgb_full_{1,2,X} writers (output.c:1998-2310), whose six int16
     * coefficients are c->yuv2rgb_{y_coeff, y_offset, v2r_coeff, v2g_coeff, u2g_coeff, u2b_coeff} (yuv2rgb.c:786-791) */
    int full_chr;
    int full_coef[6];
} FfoSwsTables;
/* init_range_convert_constants() for a source of range src_range (1 = full) going to the other one at a target of dst_depth bits */
void ffo_sws_range_constants(int src_range, int dst_depth, uint32_t *lum_coeff, int64_t *lum_offset, uint32_t *chr_coeff, int64_t *chr_offset);

void ffo_yuv2rgb_luts_init(FfoYuv2RgbLuts *l, const FfoYuv2RgbCoeffs *k);
int  ffo_yuv420p_to_rgb24(const FfoYuv2RgbLuts *l, int width, const uint8_t *const src[3], const int srcStride[3],
                          int srcSliceY, int srcSliceH, uint8_t *dst, int dstStride, int bgr);
/* the converter's 4:2:2 / source-alpha / planar-gbrp forms (layout 6: dst = {G, B, R} planes) */
int  ffo_yuv2rgb_unscaled(const FfoYuv2RgbLuts *l, int width, const uint8_t *const src[4], const int srcStride[4], int srcSliceY,
                          int srcSliceH, uint8_t *const dst[3], const int dstStride[3], int layout, int c422, int alpha);
void ffo_hscale8to15(int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *pos, int fs);
void ffo_yuv2planeX8(const int16_t *filter, int fs, const int16_t *const *src, uint8_t *dest, int dstW,
                     const uint8_t *dither, int offset);
void ffo_yuv2plane1_8(const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset);
void ffo_yuv2nv12cX(int swap, const uint8_t *dither, const int16_t *filter, int fs, const int16_t *const *u,
                    const int16_t *const *v, uint8_t *dest, int chrDstW);
/* one packed RGB line: yuv2rgb_{X,2,1}_c_template (libswscale/output.c:1789-1939); layout 0 rgb24 1 bgr24 2 argb 3 rgba 4 abgr 5 bgra */
void ffo_yuv2rgb_X(const FfoYuv2RgbLuts *l, const int16_t *lf, const int16_t *const *lum, int lfs, const int16_t *cf,
                   const int16_t *const *cu, const int16_t *const *cv, int cfs, uint8_t *dest, int dstW, int layout);
void ffo_yuv2rgb_2(const FfoYuv2RgbLuts *l, const int16_t *const lum[2], const int16_t *const cu[2], const int16_t *const cv[2], uint8_t *dest,
                   int dstW, int yalpha, int uvalpha, int layout);
void ffo_yuv2rgb_1(const FfoYuv2RgbLuts *l, const int16_t *lum, const int16_t *const cu[2], const int16_t *const cv[2], uint8_t *dest, int dstW,
                   int uvalpha, int layout);
/* the alpha bytes of a 32-bit RGB picture from the source's alpha plane, after ffo_sws_scale_frame() (yuv2rgba32_{1,2,X}_c and the _full twins) */
int  ffo_sws_rgba_alpha(const FfoSwsTables *t, const uint8_t *alpha, int alphaStride, uint8_t *dst, int dstStride);
int  ffo_sws_scale_frame(const FfoSwsTables *t, const uint8_t *const src[3], const int srcStride[3],
                         uint8_t *const dst[3], const int dstStride[3]);
/* the same above 8 bits (ffo_sws_hbd.c): a format is (depth, layout): layout 0 planar LE samples in the low bits (8-bit planar when
 * depth == 8), 1 semi-planar with the samples in the high bits (p010le / p012le / p016le), 2 semi-planar 8-bit (nv12).  Scaled
 * contexts only; planes as the format has them (planar Y, U, V; semi-planar Y, UV). */
int  ffo_sws_scale_frame_hbd(const FfoSwsTables *t, int sdepth, int slayout, int ddepth, int dlayout, const uint8_t *const src[3],
                             const int srcStride[3], uint8_t *const dst[3], const int dstStride[3]);

/* ---- h264dsp / h264qpel (ffo_h264.c), 8-bit ---- */
void ffo_h264_idct_add(uint8_t *dst, int16_t *block, ptrdiff_t stride);
void ffo_h264_idct8_add(uint8_t *dst, int16_t *block, ptrdiff_t stride);
void ffo_h264_idct_dc_add(uint8_t *dst, int16_t *block, ptrdiff_t stride);
void ffo_h264_idct8_dc_add(uint8_t *dst, int16_t *block, ptrdiff_t stride);
void ffo_h264_idct_add16(uint8_t *dst, const int *block_offset, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc);
void ffo_h264_idct8_add4(uint8_t *dst, const int *block_offset, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc);
void ffo_h264_idct_add16intra(uint8_t *dst, const int *block_offset, int16_t *block, ptrdiff_t stride,
                              const uint8_t *nnzc);
void ffo_h264_idct_add8(uint8_t **dest, const int *block_offset, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc);
void ffo_h264_luma_dc_dequant_idct(int16_t *output, int16_t *input, int qmul);
void ffo_h264_chroma_dc_dequant_idct(int16_t *block, int qmul);
void ffo_h264_add_pixels_clear(int n, uint8_t *dst, int16_t *block, ptrdiff_t stride);
/* which: FFHIP_H264_LF_* numbering (0 v_luma 1 h_luma 2 v_chroma 3 h_chroma, +4 intra) */
void ffo_h264_loop_filter(int which, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, const int8_t *tc0);
void ffo_h264_qpel(int avg, int size_idx, int mcxy, uint8_t *dst, const uint8_t *src, ptrdiff_t stride);
/* H264ChromaContext (w = 8|4|2) and H264DSPContext.weight/biweight (w = 16|8|4|2) */
void ffo_h264_chroma_mc(int avg, int w, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h, int x, int y);
void ffo_h264_weight(int w, uint8_t *block, ptrdiff_t stride, int height, int log2_denom, int weight, int offset);
void ffo_h264_biweight(int w, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd,
                       int weights, int offset);

/* ---- the same tables at ANY bit depth (8 / 9 / 10 / 12 / 14) and the 4:2:2 / MBAFF members: ffo_h264_hbd.c.  Above 8 bits samples
 *      are uint16_t, coefficients int32_t; strides stay in bytes.  kind of ffo_h264_idct_bd = FFHIP_H264_IDCT4 .. ADD_PIXELS8_CLEAR ---- */
void ffo_h264_idct_bd(int bd, int kind, uint8_t *dst, int16_t *block, ptrdiff_t stride);
void ffo_h264_idct_mb_bd(int bd, int which, uint8_t *dst, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc);
void ffo_h264_idct_add8_bd(int bd, int is422, uint8_t **dest, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t *nnzc);
void ffo_h264_luma_dc_dequant_bd(int bd, int16_t *output, int16_t *input, int qmul);
void ffo_h264_chroma_dc_dequant_bd(int bd, int is422, int16_t *block, int qmul);
/* kind: bit 0 = h_ (vertical edge), bit 1 = chroma, bit 2 = intra; inner = lines per tc0 entry (luma 4, MBAFF 2; chroma 2, MBAFF 1, 4:2:2 4, 4:2:2 MBAFF 2) */
void ffo_h264_loop_filter_bd(int bd, int kind, int inner, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, const int8_t *tc0);
void ffo_h264_qpel_bd(int bd, int avg, int size_idx, int mcxy, uint8_t *dst, const uint8_t *src, ptrdiff_t stride);
void ffo_h264_chroma_mc_bd(int bd, int avg, int w, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h, int x, int y);
void ffo_h264_weight_bd(int bd, int w, uint8_t *block, ptrdiff_t stride, int height, int log2_denom, int weight, int offset);
void ffo_h264_biweight_bd(int bd, int w, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd,
                          int weights, int offset);
/* ---- AVFloatDSPContext vector operations (ffo_fdsp.c): op numbering = FFHIP_FDSP_* of include/ffhip.h ---- */
#define FFO_FDSP_FMUL          0   /* dst = src0 * src1                                   */
#define FFO_FDSP_FMAC_SCALAR   1   /* dst += src0 * mul                                   */
#define FFO_FDSP_FMUL_SCALAR   2   /* dst = src0 * mul                                    */
#define FFO_FDSP_FMUL_WINDOW   3   /* dst[2 len] = overlap window of src0, src1 with src2 */
#define FFO_FDSP_FMUL_ADD      4   /* dst = src0 * src1 + src2                            */
#define FFO_FDSP_FMUL_REVERSE  5   /* dst[i] = src0[i] * src1[len-1-i]                    */
#define FFO_FDSP_BUTTERFLIES   6   /* (dst, src0) = (dst + src0, dst - src0), both written */
void ffo_fdsp(int op, float *dst, const float *src0, const float *src1, const float *src2, float mul, int len);
/* ---- hevcdsp at a given bit depth (8, 10, 12): pixels are uint16_t above 8 bits, strides in BYTES (ffo_hevc.c) ---- */
void ffo_hevc_idct_bd(int bd, int log2_size, int16_t *coeffs, int col_limit);
void ffo_hevc_idct_dc_bd(int bd, int log2_size, int16_t *coeffs);
void ffo_hevc_transform_4x4_luma_bd(int bd, int16_t *coeffs);
void ffo_hevc_add_residual_bd(int bd, int log2_size, uint8_t *dst, const int16_t *res, ptrdiff_t stride);
void ffo_hevc_dequant_bd(int bd, int16_t *coeffs, int log2_size);
void ffo_hevc_loop_filter_bd(int bd, int chroma, int vertical, uint8_t *pix, ptrdiff_t stride, int beta, const int32_t *tc,
                             const uint8_t *no_p, const uint8_t *no_q);
void ffo_hevc_sao_band_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *offset_val,
                          int left_class, int width, int height);
void ffo_hevc_sao_edge_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *offset_val,
                          int eo, int width, int height);
void ffo_hevc_sao_edge_restore_bd(int bd, int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t sd, ptrdiff_t ss, int eo, int offset0,
                                  const int *borders, int width, int height, const uint8_t *vert_edge, const uint8_t *horiz_edge,
                                  const uint8_t *diag_edge);
void ffo_hevc_mc_bd(int bd, int chroma, int uni, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, int mx,
                    int my, int width);
void ffo_hevc_mc_w_bd(int bd, int chroma, int mode, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                      const int16_t *src2, int height, int denom, int wx0, int wx1, int ox, int mx, int my, int width);
/* ---- HEVC inverse transforms, 8-bit (ffo_hevc.c): HEVCDSPContext.idct / idct_dc / transform_4x4_luma / add_residual ---- */
int  ffo_hevc_coef(int k, int i);                                   /* the 32-point core matrix */
void ffo_hevc_idct(int log2_size, int16_t *coeffs, int col_limit);  /* log2_size 2..5 */
void ffo_hevc_idct_dc(int log2_size, int16_t *coeffs);
void ffo_hevc_transform_4x4_luma(int16_t *coeffs);
void ffo_hevc_add_residual(int log2_size, uint8_t *dst, const int16_t *res, ptrdiff_t stride);
/* put_hevc_{qpel,epel}[..][!!my][!!mx] (uni = 0: int16 dst, row stride 64) and put_hevc_{qpel,epel}_uni (uni = 1: pixels) */
void ffo_hevc_mc(int chroma, int uni, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, int mx,
                 int my, int width);
void ffo_hevc_mc_w(int chroma, int mode, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                   int height, int denom, int wx0, int wx1, int ox, int mx, int my, int width);
/* VP9DSPContext.itxfm_add[tx][txtp], 8 bits (ffo_vp9.c): tx 0..3 = 4x4..32x32, 4 = WHT; consumes the block */
/* vp9dsp above 8 bits (10, 12): uint16_t pixels, int32 coefficients, strides in bytes (ffo_vp9.c) */
void ffo_vp9_loopfilter_sb(int bd, int ss_h, int ss_v, const uint8_t *lflvl_level, const uint8_t *lflvl_mask, int row, int col, uint8_t *y,
                           uint8_t *u, uint8_t *v, ptrdiff_t ls_y, ptrdiff_t ls_uv, const uint8_t *lim_lut, const uint8_t *mblim_lut);
void ffo_vp9_itxfm_add_bd(int bd, int tx, int txtp, uint8_t *dst, ptrdiff_t stride, int32_t *block, int eob);
void ffo_vp9_mc_bd(int bd, int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width,
                   int height, int mx, int my);
void ffo_vp9_smc_bd(int bd, int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width,
                    int height, int mx, int my, int dx, int dy);
void ffo_vp9_loop_filter_bd(int bd, int wd, int dir, uint8_t *dst, ptrdiff_t stride, int E, int I, int H);
void ffo_vp9_intra_pred_bd(int bd, int tx, int mode, uint8_t *dst, ptrdiff_t stride, const uint8_t *left, const uint8_t *top);
void ffo_vp9_itxfm_add(int tx, int txtp, uint8_t *dst, ptrdiff_t stride, int16_t *block, int eob);
/* VP9DSPContext.mc[..][filter][avg][!!mx][!!my], 8 bits: filter 0 smooth, 1 regular, 2 sharp, 3 bilinear */
void ffo_vp9_mc(int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width, int height,
                int mx, int my);
/* one 8-sample segment of a VP9 edge: wd 4 | 8 | 16, dir 0 column edge ("h"), 1 row edge ("v") */
void ffo_vp9_loop_filter(int wd, int dir, uint8_t *dst, ptrdiff_t stride, int E, int I, int H);
/* VP9DSPContext.intra_pred[tx][mode]: tx 0..3, mode = enum IntraPredMode 0..14; top[-1] is the corner, left[] bottom to top */
void ffo_vp9_intra_pred(int tx, int mode, uint8_t *dst, ptrdiff_t stride, const uint8_t *left, const uint8_t *top);
/* VP9DSPContext.smc[..][filter][avg]: scaled motion compensation, dx / dy = step in sixteenths of a reference sample */
void ffo_vp9_smc(int filter, int avg, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int width, int height,
                 int mx, int my, int dx, int dy);
void ffo_hevc_dequant(int16_t *coeffs, int log2_size);
void ffo_hevc_transform_rdpcm(int16_t *coeffs, int log2_size, int mode);
void ffo_hevc_sao_edge_restore(int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t sd, ptrdiff_t ss, int eo, int offset0,
                               const int *borders, int width, int height, const uint8_t *vert_edge, const uint8_t *horiz_edge,
                               const uint8_t *diag_edge);
/* sao_band_filter / sao_edge_filter (eo 0..3); the reference's edge filter uses stride_src = 192 */
void ffo_hevc_sao_band(uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *offset_val,
                       int left_class, int width, int height);
void ffo_hevc_sao_edge(uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *offset_val, int eo,
                       int width, int height);
/* hevc_{h,v}_loop_filter_{luma,chroma}: vertical = 1 for hevc_v_* (the edge is vertical); beta unused for chroma */
void ffo_hevc_loop_filter(int chroma, int vertical, uint8_t *pix, ptrdiff_t stride, int beta, const int32_t *tc,
                          const uint8_t *no_p, const uint8_t *no_q);
/* frame-order luma deblock, same edge array layout as ffhip_h264_deblock_frame_dev (include/ffhip.h) */
typedef struct FfoH264Edge {
    int32_t offset;
    uint8_t kind, alpha, beta, pad;
    int8_t  tc0[4];
} FfoH264Edge;
/* frame order at 8 .. 14 bits (ffo_h264_hbd.c): uint16_t samples above 8 bits, stride in bytes */
void ffo_h264_deblock_frame_bd(int bd, int chroma, uint8_t *plane, ptrdiff_t stride, int mb_w, int mb_h, const FfoH264Edge *edges);
void ffo_h264_deblock_frame_c422_bd(int bd, uint8_t *plane, ptrdiff_t stride, int mb_w, int mb_h, const FfoH264Edge *edges);
void ffo_h264_deblock_frame(uint8_t *luma, ptrdiff_t stride, int mb_w, int mb_h, const FfoH264Edge *edges);
/* one 4:2:0 chroma plane in frame order: edges[(mb * 2 + dir) * 2 + e], edges at 0 and 4 */
void ffo_h264_deblock_frame_chroma(uint8_t *plane, ptrdiff_t stride, int mb_w, int mb_h, const FfoH264Edge *edges);

/* ---- me_cmp + ESA (ffo_mecmp.c) ---- */
int      ffo_sad(int width, const uint8_t *a, const uint8_t *b, ptrdiff_t stride, int h);
int      ffo_hadamard8_diff8x8(const uint8_t *dst, const uint8_t *src, ptrdiff_t stride);
int      ffo_hadamard8_diff16(const uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h);
/* kind 2 pix_abs_x2, 3 _y2, 4 _xy2, 5 sse, 6 nsse (context-free weight 8): me_cmp.c:53-104,184-440 */
int ffo_me_cmp_other(int kind, int width, const uint8_t *a, const uint8_t *b, ptrdiff_t stride, int h);
uint64_t ffo_me_search_esa(const uint8_t *cur, const uint8_t *ref, int linesize, int width, int height, int mb_size,
                           int search_param, int cost_kind, int x_mb, int y_mb, int *mv);
void     ffo_me_esa_frame(const uint8_t *cur, const uint8_t *ref, int linesize, int width, int height, int mb_size,
                          int search_param, int cost_kind, int16_t *mv_out, uint32_t *cost_out);

/* ---- av_tx float MDCT (ffo_tx.c) ---- */
typedef struct FfoTx FfoTx;
FfoTx *ffo_mdct_create(int inv, int len, float scale);
int    ffo_mdct_pfa_factor(int len); /* 0: power-of-two codelet; else the N of ff_tx_mdct_pfa_NxM av_tx picks */
void   ffo_mdct_run(const FfoTx *s, float *out, const float *in, ptrdiff_t stride);
void   ffo_mdct_free(FfoTx *s);
/* AV_TX_FLOAT_FFT, power-of-two len: complex (re, im) floats in and out */
void   ffo_imdct_full_run(const FfoTx *s, float *out, const float *in); /* AV_TX_FULL_IMDCT: 2 * len outputs */
void   ffo_fft_run(int inv, int len, float *out, const float *in);       /* ... or F * 2^k, F = 3 / 5 / 7 / 9 / 15 (ff_tx_fft_pfa) */
int    ffo_fft_pfa_factor(int len);                                      /* the F of such a length, 0 for none */
/* AV_TX_FLOAT_RDFT, power-of-two: inv == 0: in = len reals, out = len/2 + 1 complex; inv == 1: the other way round */
void   ffo_rdft_run(int inv, int len, float scale, float *out, const float *in);
/* ffo_tx_wide.c: AV_TX_DOUBLE_* / AV_TX_INT32_* FFT and MDCT, powers of two; is_int: int32_t samples, else double; contiguous rows */
void   ffo_txw_fft_run(int is_int, int inv, int len, void *out, const void *in);
void   ffo_txw_mdct_run(int is_int, int inv, int len, double scale, void *out, const void *in);
/* mode 1: AV_TX_REAL_TO_REAL (len/2 + 1 floats out), 2: AV_TX_REAL_TO_IMAGINARY (len/2 floats out); forward, len a power of two >= 8 */
void   ffo_rdft_half_run(int mode, int len, float scale, float *out, const float *in);
/* AV_TX_FLOAT_DCT_I / _DST_I forward, n even (tx_template.c:2006-2075); the inputs `stride` bytes apart */
void   ffo_dcst1_run(int is_dst, int n, float scale, float *out, const float *in, ptrdiff_t stride);
/* AV_TX_FLOAT_DCT: DCT-II (inv 0) / DCT-III (inv 1) of n real samples, n a power of two (tx_template.c:1832-2002) */
void   ffo_dct_run(int inv, int n, float scale, float *out, const float *in);
/* double-precision cosine-sum definition (ff_tx_mdct_naive_fwd/_inv, tx_template.c:1144-1193) */
void   ffo_mdct_naive_fwd(int len, double scale, double *out, const float *in);
void   ffo_mdct_naive_inv(int len, double scale, double *out, const float *in);

/* ---- ffo_h264pred.c: H264PredContext, H.264 codec, 8 bits, chroma_format_idc <= 1 (libavcodec/h264pred.h:92-116) ---- */
void ffo_h264_pred4x4(int mode, uint8_t *src, const uint8_t *topright, ptrdiff_t stride);
void ffo_h264_pred8x8l(int mode, uint8_t *src, int has_topleft, int has_topright, ptrdiff_t stride);
void ffo_h264_pred8x8(int mode, uint8_t *src, ptrdiff_t stride);
void ffo_h264_pred16x16(int mode, uint8_t *src, ptrdiff_t stride);
void ffo_h264_pred8x16(int mode, uint8_t *src, ptrdiff_t stride);   /* pred8x8[] at chroma_format_idc 2 */
void ffo_h264_pred4x4_add(int mode, uint8_t *pix, int16_t *block, ptrdiff_t stride);
void ffo_h264_pred8x8l_add(int mode, uint8_t *pix, int16_t *block, ptrdiff_t stride);
void ffo_h264_pred8x8l_filter_add(int mode, uint8_t *pix, int16_t *block, int has_topleft, int has_topright, ptrdiff_t stride);
void ffo_h264_pred8x8_add(int mode, uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride);
void ffo_h264_pred8x16_add(int mode, uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride);
void ffo_h264_pred16x16_add(int mode, uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride);

/* one TNS filter of a channel-frame: `size` coefficients from `start` (index into the frame's 1024) stepping by `inc` */
typedef struct FfoAacTnsFilter { int start, size, inc, order; float coef[20]; } FfoAacTnsFilter;
int  ffo_aac_tns_filters(FfoAacTnsFilter *out, const int n_filt[8], const int length[8][4], const int direction[8][4], const int order[8][4],
                         const float coef[8][4][20], int num_windows, int num_swb, const uint16_t *swb_offset, int tns_max_bands, int max_sfb);
void ffo_aac_tns_run(float *coef, const FfoAacTnsFilter *r, int decode);
/* ---- ffo_aac.c: the stereo tools and long-term prediction of AACDecDSP, float (aacdec_dsp_template.c:83-160,225-320) ---- */
void ffo_aac_apply_mid_side_stereo(float *ch0, float *ch1, int num_window_groups, const uint8_t *group_len, int max_sfb_ste,
                                   const uint8_t *ms_mask, const int *band_type0, const int *band_type1, const uint16_t *swb_offset);
void ffo_aac_apply_intensity_stereo(const float *coef0, float *coef1, int num_window_groups, const uint8_t *group_len, int max_sfb,
                                    int ms_present, const uint8_t *ms_mask, const int *band_type1, const float *sf1, const uint16_t *swb_offset);
void ffo_aac_apply_ltp(const FfoTx *mdct_ltp, const float *const windows[4], float *coeffs, const float *ltp_state, int lag, float coef,
                       const int8_t *used, const int seq[2], const int kb[2], int max_sfb, const uint16_t *swb_offset,
                       const FfoAacTnsFilter *tns, int ntns, float *predFreq);
void ffo_aac_update_ltp(const float *const windows[4], float *ltp_state, const float *buf_mdct, const float *saved, const float *output,
                        int seq0, int kb0);
/* ---- ffo_aac.c: AACDecDSP.imdct_and_windowing, float, 1024-sample frames (libavcodec/aac/aacdec_dsp_template.c:325-387) ---- */
void ffo_aac_sine_window(float *w, int n);
void ffo_aac_kbd_window(float *w, float alpha, int n);
/* windows[]: sine_1024, sine_128, kbd_long_1024, kbd_short_128; seq / kb = { this frame, previous frame }; saved[512] in / out */
void ffo_aac_apply_prediction(float *ps, float *coef, int is_long, int *initialized, int predictor_present, const uint8_t *prediction_used,
                              int pred_sfb_max, const uint16_t *swb_offset, int reset_group);
void ffo_aac_apply_dependent_coupling(float *dest, const float *src, int num_window_groups, const uint8_t *group_len, int max_sfb,
                                      const int *band_type, const float *gain, const uint16_t *swb_offset);
void ffo_aac_apply_independent_coupling(float *dest, const float *src, float gain, int len);
void ffo_aac_imdct_and_windowing_ld(const FfoTx *mdct512, const float *sine_512, const float *sine_128, const float *coeffs, int kb_prev,
                                    float *saved, float *out);
void ffo_aac_imdct_and_windowing_eld(int n, const FfoTx *mdct, const float *window, const float *coeffs, float *saved, float *out);
void ffo_aac_imdct_and_windowing_len(int L, int in_short_stride, const FfoTx *mdct_long, const FfoTx *mdct_short, const float *const windows[4],
                                     const float *coeffs, const int seq[2], const int kb[2], float *saved, float *out);
void ffo_aac_imdct_and_windowing(const FfoTx *mdct1024, const FfoTx *mdct128, const float *const windows[4], const float *coeffs,
                                 const int seq[2], const int kb[2], float *saved, float *out);

/* ---- swscale micro-op lists (ffo_sws_uops.c): the five entry points of include/ffhip.h's SwsOpBackend section, on the CPU ---- */
struct FFHipSwsUOp;
struct FFHipSwsOpExec;
typedef struct FfoSwsUOps FfoSwsUOps;
int  ffo_sws_uops_compile(const struct FFHipSwsUOp *uops, int n, FfoSwsUOps **out);
void ffo_sws_uops_free(FfoSwsUOps **p);
int  ffo_sws_uops_block_size(const FfoSwsUOps *p);
void ffo_sws_uops_func(const struct FFHipSwsOpExec *e, const void *priv, int bx_start, int y_start, int bx_end, int y_end);

/* ffo_sws_rgbin.c: the input stage of packed 8-bit RGB sources (lines for ffo_sws_scale_frame_hbd at sdepth 14 | 0x100) */
void ffo_sws_rgb2yuv_default(int32_t t[9]);
int  ffo_sws_rgb_half(int srcW, int dstW, int chrDstHSubSample, int flags);
void ffo_sws_rgb_in(const uint8_t *src, ptrdiff_t stride, int w, int h, int bpp, int ro, int go, int bo, int half, const int32_t t[9],
                    uint16_t *Y, ptrdiff_t ys, uint16_t *U, uint16_t *V, ptrdiff_t cs);

#endif
