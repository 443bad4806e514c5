/*
 * ffref.h — flat C accessors onto the REAL reference (FFmpeg) compiled by oracle/refbuild/Makefile.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, tools/make_golden.py and bench.py's cpu_baseline leg.
 * Never linked or loaded by the product (libffhip.so / ffmpeg_amd).
 *
 * The shim exists because the reference's DSP tables are structs of function pointers with
 * internal layouts; ctypes callers need flat symbols.  Every function forwards to the pointer the
 * reference's own ff_*_init()/sws_getContext()/av_tx_init() installed with av_force_cpu_flags(0).
 */
#ifndef FFREF_H
#define FFREF_H
#include <stddef.h>
#include <stdint.h>

/* ---- libswscale ---- */
void *ffref_sws_create(int srcW, int srcH, int srcFmt, int dstW, int dstH, int dstFmt, int flags, int threads);
void *ffref_sws_create_ranges(int srcW, int srcH, int srcFmt, int dstW, int dstH, int dstFmt, int flags, int threads, int src_range, int dst_range);
void  ffref_sws_free(void *ctx);
int   ffref_sws_set_colorspace(void *ctx, int cs, int src_range, int brightness, int contrast, int saturation);
void  ffref_sws_coefficients(int cs, int out[4]);
int   ffref_sws_scale(void *ctx, const uint8_t *const src[], const int srcStride[], int y, int h,
                      uint8_t *const dst[], const int dstStride[]);
/* which: 0 hLum 1 hChr 2 vLum 3 vChr.  Returns filter size; *n = number of output samples */
int   ffref_sws_filter(void *ctx, int which, const int16_t **filter, const int32_t **pos, int *n);
/* 1 when the context installed a convert_unscaled special converter */
int   ffref_sws_is_unscaled(void *ctx);
int   ffref_sws_flags(void *ctx);
void  ffref_sws_full_coeffs(void *ctx, int out[6]);
/* the four (already divided) yuv2rgb table coefficients + y terms the context derived */
void  ffref_sws_yuv2rgb_tables(void *ctx, const uint8_t **rV, const int **gU, const int **gV, const uint8_t **bU);
int   ffref_pix_fmt(const char *name);
/* per-line function pointers of a context */
void  ffref_sws_hyscale(void *ctx, int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter,
                        const int32_t *pos, int fs);
void  ffref_sws_yuv2planeX(void *ctx, const int16_t *filter, int fs, const int16_t **src, uint8_t *dest,
                           int dstW, const uint8_t *dither, int offset);
void  ffref_sws_yuv2plane1(void *ctx, const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset);
void  ffref_sws_yuv2nv12cX(void *ctx, int dstFormat, const uint8_t *chrDither, const int16_t *chrFilter, int fs,
                           const int16_t **chrU, const int16_t **chrV, uint8_t *dest, int dstW);

/* ---- libavcodec h264dsp / h264qpel / me_cmp (8-bit) ---- */
/* which: 0 idct_add 1 idct8_add 2 idct_dc_add 3 idct8_dc_add */
void ffref_sws_yuv2packedX(void *ctx, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize, const int16_t *chrFilter,
                           const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize, uint8_t *dest, int dstW, int y);
void ffref_sws_yuv2packed2(void *ctx, const int16_t *lumSrc[2], const int16_t *chrUSrc[2], const int16_t *chrVSrc[2], uint8_t *dest, int dstW,
                           int yalpha, int uvalpha, int y);
void ffref_sws_yuv2packed1(void *ctx, const int16_t *lumSrc, const int16_t *chrUSrc[2], const int16_t *chrVSrc[2], uint8_t *dest, int dstW,
                           int uvalpha, int y);
void ffref_h264_idct(int which, uint8_t *dst, int16_t *block, ptrdiff_t stride);
void ffref_h264_luma_dc_dequant_idct(int16_t *output, int16_t *input, int qmul);
void ffref_h264_chroma_dc_dequant_idct(int16_t *block, int qmul);
void ffref_h264_add_pixels_clear(int n, uint8_t *dst, int16_t *block, ptrdiff_t stride);
/* CPU-baseline runners: a batch split statically over pthreads (disjoint blocks / independent frames) */
int  ffref_h264_idct_batch(int which, uint8_t *dst, ptrdiff_t stride, const int32_t *off, int16_t *blk, int n, int threads);
int  ffref_h264_idct_batch_timed(int which, uint8_t *dst, ptrdiff_t stride, const int32_t *off, int16_t *blk, int n, int threads,
                                 double min_seconds, double *seconds, int *passes);
int  ffref_sws_scale_frames_mt(void *const *ctxs, const uint8_t *const *const *srcs, const int *ss, uint8_t *const *const *dsts,
                               const int *ds, int srcH, int threads, int reps);
/* which: 0 idct_add16 1 idct8_add4 2 idct_add16intra */
void ffref_h264_idct_multi(int which, uint8_t *dst, const int *blockoffset, int16_t *block, ptrdiff_t stride,
                           const uint8_t *nnzc);
void ffref_h264_idct_add8(uint8_t **dst, const int *blockoffset, int16_t *block, ptrdiff_t stride,
                          const uint8_t *nnzc, int chroma_format_idc);
/* which: 0 v_luma 1 h_luma 2 v_chroma 3 h_chroma (tc0 used); 4 v_luma_intra 5 h_luma_intra 6 v_chroma_intra 7 h_chroma_intra */
void ffref_h264_loop_filter(int which, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0);
/* avg: 0 put 1 avg; size_idx: 0 16x16 1 8x8 2 4x4; mcxy = x + 4*y */
void ffref_h264_qpel(int avg, int size_idx, int mcxy, uint8_t *dst, const uint8_t *src, ptrdiff_t stride);
/* re-initialises the h264dsp / h264qpel / h264chroma tables the calls above use at 8 / 9 / 10 / 12 / 14 bits */
void ffref_h264_set_bit_depth(int bit_depth);
void ffref_h264_loop_filter_variant(int kind, int variant, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0);
void ffref_h264_chroma_dc_dequant_idct_422(int16_t *block, int qmul);
/* H264ChromaContext.{put,avg}_h264_chroma_pixels_tab[idx]: idx 0 = 8 wide, 1 = 4, 2 = 2; x,y in 1/8 pel */
void ffref_h264_chroma(int avg, int idx, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h, int x, int y);
/* H264DSPContext.weight_h264_pixels_tab[idx] / biweight_h264_pixels_tab[idx]: idx 0 = 16 wide, 1 = 8, 2 = 4, 3 = 2 */
void ffref_h264_weight(int idx, uint8_t *block, ptrdiff_t stride, int height, int log2_denom, int weight, int offset);
void ffref_h264_biweight(int idx, uint8_t *dst, uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd,
                         int weights, int offset);
/* kind: 0 sad 1 hadamard8_diff 2 sse ; idx: 0 = 16 wide, 1 = 8 wide */
int  ffref_me_cmp(int kind, int idx, const uint8_t *blk1, const uint8_t *blk2, ptrdiff_t stride, int h);
/* libavfilter ESA search (vf_mestimate semantics); returns cost, mv[2] = absolute best position */
uint64_t ffref_me_search_esa(const uint8_t *cur, const uint8_t *ref, int linesize, int width, int height,
                             int mb_size, int search_param, int x_mb, int y_mb, int *mv);

/* ---- libavutil av_tx ---- */
/* type: AVTXType value (0 FLOAT_FFT, 1 FLOAT_MDCT); returns ctx or NULL */
void *ffref_tx_create(int type, int inv, int len, float scale, uint64_t flags);
void  ffref_tx_run(void *ctx, void *out, void *in, ptrdiff_t stride);
void  ffref_tx_free(void *ctx);

/* ---- libswscale op backends (ffref_shim_ops.c): the `hip` SwsOpBackend inside the reference's own dispatch ---- */
struct FFHipSwsUOp;
struct FFHipSwsOpExec;
#define FFREF_SWS_BACKEND_C      (1 << 1)   /* SWS_BACKEND_C */
#define FFREF_SWS_BACKEND_MEMCPY (1 << 2)   /* SWS_BACKEND_MEMCPY */
#define FFREF_SWS_BACKEND_HIP    (1 << 6)   /* the bit ffref_shim_ops.c gives backend_hip */
/* the five functions behind backend_hip: ffhip_sws_uops_{compile,free,block_size,func,set_fallback} or the oracle's */
int  ffref_sws_hip_bind(void *compile, void *free_, void *block_size, void *func, void *set_fallback);
long ffref_sws_hip_count(int what);   /* 0: lists the bound backend compiled, 1: lists it declined; < 0: reset */
int  ffref_sws_frame_convert(int backends, int flags, int scaler, int dither, int threads,
                             int sw, int sh, int sfmt, const uint8_t *const src[4], const int sstride[4],
                             int dw, int dh, int dfmt, uint8_t *const dst[4], const int dstride[4]);
int  ffref_sws_uops_run_c(const struct FFHipSwsUOp *uops, int n, const struct FFHipSwsOpExec *exec, int x_start, int y_start, int x_end,
                          int y_end);
int  ffref_sws_filter_generate(int scaler, int src_size, int dst_size, int *filter_size, int *weights, int weights_cap, int *offsets);
/* instance idx of the table of micro-ops backend_c implements (uops_macros.h); returns the number of instances */
int  ffref_sws_uop_instance(int idx, struct FFHipSwsUOp *out, char *name, int cap);
int  ffref_image_layout(int fmt, int w, int h, int align, int linesize[4], int lines[4]);
int  ffref_sws_describe_uops(int flags, int scaler, int sw, int sh, int sfmt, int dw, int dh, int dfmt, char *buf, int cap);

#endif
