/* h264_kernels.h — launchers of the h264dsp / h264qpel kernels (internal to libffhip). */
#ifndef FFHIP_H264_KERNELS_H
#define FFHIP_H264_KERNELS_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "ffhip.h"
#include "progress_pool.h"

/* the block lists of up to three planes in one launch (chroma MC, weighted prediction); empty segments are dropped */
struct FFHipPlaneSeg { uint8_t *dst; const uint8_t *src; const void *blocks; int stride, n, first; };
struct FFHipPlaneMulti { FFHipPlaneSeg seg[3]; int nseg; int pic_w = 0, pic_h = 0; /* chroma MC: FFHIP_MC_EMU clamps to these */ };
int ffhip_launch_h264_chroma_mc_multi(FFHipPlaneMulti &M, hipStream_t stream);
int ffhip_launch_h264_weight_multi(FFHipPlaneMulti &M, hipStream_t stream);
/* several idct_add lists in one launch (kinds FFHIP_H264_IDCT4 .. FFHIP_H264_IDCT8_DC); empty segments are dropped */
struct FFHipIdctSeg { uint8_t *dst; const int32_t *offs; int16_t *coef; int stride, n, kind, first; };
struct FFHipIdctMulti { FFHipIdctSeg seg[12]; int nseg; };
int ffhip_launch_h264_idct_multi(FFHipIdctMulti &M, hipStream_t stream);
int ffhip_launch_h264_idct_add(int kind, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset,
                               int16_t *blocks, int n, hipStream_t stream);
int ffhip_launch_h264_idct_add_mb(int which, uint8_t *dst_base, ptrdiff_t stride, const int32_t *mb_offset,
                                  const int32_t *blockoffset16, int16_t *blocks, const uint8_t *nnzc, int nmb,
                                  hipStream_t stream);
int ffhip_launch_h264_idct_add8(uint8_t *cb_base, uint8_t *cr_base, ptrdiff_t stride, const int32_t *mb_offset, const int32_t *blockoffset48,
                                int16_t *blocks, const uint8_t *nnzc, int nmb, hipStream_t stream);
int ffhip_launch_h264_luma_dc_dequant(int16_t *output, size_t out_pitch, const int16_t *input, size_t in_pitch, const int32_t *qmul, int n,
                                      hipStream_t stream);
int ffhip_launch_h264_chroma_dc_dequant(int16_t *blocks, const int32_t *block_offset, const int32_t *qmul, int n, hipStream_t stream);
int ffhip_launch_h264_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n,
                                  hipStream_t stream);
/* any bit depth + the MBAFF / 4:2:2 members (h264_hbd.hip) */
int ffhip_launch_h264_idct_add_bd(int bd, int kind, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset, int16_t *blocks, int n,
                                  hipStream_t stream);
int ffhip_launch_h264_idct_mb_bd(int bd, int which, uint8_t *dst_base, uint8_t *dst2, ptrdiff_t stride, const int32_t *mb_offset,
                                 const int32_t *blockoffset, int16_t *blocks, const uint8_t *nnzc, int nmb, hipStream_t stream);
int ffhip_launch_h264_dc_dequant_bd(int bd, int which, int16_t *output, size_t out_pitch, const int16_t *input, size_t in_pitch,
                                    const int32_t *block_offset, const int32_t *qmul, int n, hipStream_t stream);
/* alpha_beta: NULL (the records' bytes) or n x { alpha, beta } ints overriding them */
int ffhip_launch_h264_loop_filter_bd(int bd, uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n, hipStream_t stream,
                                     const int32_t *alpha_beta = nullptr);
/* pic_w / pic_h (samples of the plane; 0: no record is read as FFHIP_MC_EMU) */
int ffhip_launch_h264_qpel_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n, hipStream_t stream,
                              int pic_w = 0, int pic_h = 0);
int ffhip_launch_h264_chroma_mc_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks, int n,
                                   hipStream_t stream, int pic_w = 0, int pic_h = 0);
int ffhip_launch_h264_weight_bd(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks, int n,
                                hipStream_t stream);
int ffhip_launch_h264_deblock_frames_bd(int bd, int chroma, uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                        const FFHipH264Edge *edges, hipStream_t stream);
int ffhip_launch_h264_pred_bd(int bd, int kind, uint8_t *plane, ptrdiff_t stride, int16_t *coeffs, const FFHipH264Pred *blocks, int n,
                              hipStream_t stream);
/* host faces of those (shims_h264_hbd.hip) */
int ffhip_h264dsp_fill_generic(FFHipH264DSPContext *c, FFHipH264DSPContext *o, int bit_depth, int chroma_format_idc);
int ffhip_h264qpel_init_generic(FFHipH264QpelContext *c, int bit_depth);
int ffhip_h264chroma_init_generic(FFHipH264ChromaContext *c, int bit_depth);
int ffhip_h264weight_init_generic(FFHipH264WeightContext *c, int bit_depth);
/* up to FFHIP_INTRA_PICS pictures of one geometry in one launch of the intra wavefront (== FFHipH264IntraPic of include/ffhip.h) */
#define FFHIP_INTRA_PICS 32
typedef FFHipH264IntraPic FFHipIntraPic;
struct FFHipIntraPics { int n; int pad; FFHipIntraPic pic[FFHIP_INTRA_PICS]; };
/* luma_only: every entry is one plane of a 4:4:4 picture (y = the plane; cb / cr unused but non-null), its records that plane's */
int ffhip_launch_h264_intra_frames_bd(int bd, int npics, const FFHipIntraPic *pics, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                                      hipStream_t stream, int luma_only = 0);
int ffhip_launch_h264_intra_frame(uint8_t *y, uint8_t *cb, uint8_t *cr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                                  const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs, hipStream_t stream);
int ffhip_launch_h264_intra_frame_bd(int bd, uint8_t *y, uint8_t *cb, uint8_t *cr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                                     const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs, hipStream_t stream);
/* 4:2:2 (kernels/h264_c422.hip): the chroma planes' intra wavefront and their frame-order in-loop filter, one picture per launch */
int ffhip_launch_h264_intra_c422(int bd, uint8_t *cb, uint8_t *cr, ptrdiff_t sc, int mb_w, int mb_h, const FFHipH264IntraC422 *recs,
                                 const int32_t *row_start, const int16_t *coefs, hipStream_t stream);
int ffhip_launch_h264_deblock_c422(int bd, uint8_t *plane, ptrdiff_t stride, int mb_w, int mb_h, const FFHipH264Edge *edges, hipStream_t stream);
/* ... of several pictures of one geometry side by side (blockIdx.y = the picture / the plane) */
#define FFHIP_C422_PICS 32
struct FFHipH264C422Pic { uint8_t *cb, *cr; const FFHipH264IntraC422 *recs; const int32_t *row_start; const int16_t *coefs; };
int ffhip_launch_h264_intra_c422_pics(int bd, int npics, const FFHipH264C422Pic *pics, ptrdiff_t sc, int mb_w, int mb_h, hipStream_t stream);
int ffhip_launch_h264_deblock_c422_planes(int bd, int nplanes, uint8_t *const *planes, const FFHipH264Edge *const *edges, ptrdiff_t stride, int mb_w,
                                          int mb_h, hipStream_t stream);
#define FFHIP_DB_PTRS 32
int ffhip_launch_h264_deblock_pictures_bd(int bd, int chroma, uint8_t *const *planes, const FFHipH264Edge *const *edges, int nframes, ptrdiff_t stride,
                                          int mb_w, int mb_h, hipStream_t stream);
int ffhip_launch_h264_deblock_frame(uint8_t *luma, ptrdiff_t stride, int mb_w, int mb_h, const FFHipH264Edge *edges,
                                    hipStream_t stream);
int ffhip_launch_h264_deblock_frames_chroma(uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                            const FFHipH264Edge *edges, hipStream_t stream);
int ffhip_launch_h264_deblock_frames(uint8_t *luma, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                     const FFHipH264Edge *edges, hipStream_t stream);
int ffhip_launch_h264_qpel(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                           hipStream_t stream, int pic_w = 0, int pic_h = 0);
int ffhip_launch_h264_chroma_mc(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks, int n,
                                hipStream_t stream, int pic_w = 0, int pic_h = 0);
int ffhip_launch_h264_weight(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks, int n,
                             hipStream_t stream);
int ffhip_launch_h264_pred(int kind, uint8_t *plane, ptrdiff_t stride, int16_t *coeffs, const FFHipH264Pred *blocks, int n, hipStream_t stream);
int ffhip_launch_hevc_idct(int kind, int log2_size, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipHevcTU *tus, int n,
                           hipStream_t stream);
/* the hevcdsp kernels at bit depth 8 / 10 / 12 (16-bit samples above 8, strides and offsets in bytes) */
int ffhip_launch_hevc_idct_bd(int bd, int kind, int log2_size, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipHevcTU *tus, int n,
                              hipStream_t stream);
int ffhip_launch_hevc_loop_filter_bd(int bd, uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, hipStream_t stream);
int ffhip_launch_hevc_sao_bd(int bd, uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSao *blocks, int n,
                             hipStream_t stream);
int ffhip_launch_hevc_sao_restore_bd(int bd, uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSaoRestore *blocks,
                                     int n, hipStream_t stream);
int ffhip_launch_hevc_mc_bd(int bd, int chroma, int mode, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                            const int16_t *src2, const void *blocks, int n, hipStream_t stream);
int ffhip_launch_hevc_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, hipStream_t stream);
int ffhip_launch_hevc_sao(uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSao *blocks, int n, hipStream_t stream);
int ffhip_launch_vp9_smc(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const FFHipVp9ScaledBlock *blocks, int n,
                         hipStream_t stream);
int ffhip_launch_vp9_intra(int tx, uint8_t *dst, ptrdiff_t stride, const uint8_t *edges, const FFHipVp9Intra *blocks, int n, hipStream_t stream);
int ffhip_launch_vp9_loop_filter(uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, hipStream_t stream);
/* the vp9dsp kernels at bit depth 8 / 10 / 12 (uint16_t samples and int32 coefficients above 8; strides and offsets in bytes) */
/* up to FFHIP_VP9_LF_PICS pictures of one geometry per launch of the superblock-order loop filter (== FFHipVp9LfPic of include/ffhip.h) */
#define FFHIP_VP9_LF_PICS 32
struct FFHipVp9LfPics { int n; int pad; FFHipVp9LfPic pic[FFHIP_VP9_LF_PICS]; };
int ffhip_launch_vp9_lf_frames(int bd, int npics, const FFHipVp9LfPic *pics, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows, hipStream_t stream,
                               int planes444 = 0);
/* ... with the chroma tables of the rectangular-superblock formats (== FFHipVp9LfPicC) */
struct FFHipVp9LfPicsC { int n; int pad; FFHipVp9LfPicC pic[FFHIP_VP9_LF_PICS]; };
int ffhip_launch_vp9_lf_frames_ssc(int bd, int ss_h, int ss_v, int npics, const FFHipVp9LfPicC *pics, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                                   hipStream_t stream);
/* 4:2:2 / 4:4:0: luma by `tabs`' y part, the rectangular chroma superblocks by `ctabs` */
int ffhip_launch_vp9_lf_frame_ssc(int bd, int ss_h, int ss_v, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                                  const FFHipVp9LfSb *tabs, const FFHipVp9LfSbC *ctabs, hipStream_t stream);
int ffhip_launch_vp9_lf_frame(int bd, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t sy, ptrdiff_t suv, int cols, int rows,
                              const FFHipVp9LfSb *tabs, hipStream_t stream, int planes444 = 0);
int ffhip_launch_vp9_itxfm_bd(int bd, int tx, void *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n, hipStream_t stream);
int ffhip_launch_vp9_loop_filter_bd(int bd, uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, hipStream_t stream);
int ffhip_launch_vp9_intra_bd(int bd, int tx, uint8_t *dst, ptrdiff_t stride, const uint8_t *edges, const FFHipVp9Intra *blocks, int n,
                              hipStream_t stream);
int ffhip_launch_vp9_smc_bd(int bd, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const FFHipVp9ScaledBlock *blocks,
                            int n, hipStream_t stream);
int ffhip_launch_vp9_mc_bd(int bd, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const FFHipVp9McBlock *blocks, int n,
                           hipStream_t stream);
int ffhip_launch_vp9_mc(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const FFHipVp9McBlock *blocks, int n,
                        hipStream_t stream);
int ffhip_launch_vp9_itxfm(int tx, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n, hipStream_t stream);
int ffhip_launch_hevc_sao_restore(uint8_t *dst, ptrdiff_t sd, const uint8_t *src, ptrdiff_t ss, const FFHipHevcSaoRestore *blocks, int n,
                                  hipStream_t stream);
/* kernels/hevc_qpel_m.hip: the 16 x 16 luma blocks of a put_hevc_qpel* batch on the matrix cores (mode as ffhip_launch_hevc_mc) */
bool ffhip_hevc_qpel_m_ok(int mode, ptrdiff_t dststride, ptrdiff_t srcstride);
void ffhip_launch_hevc_qpel_m(int mode, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                              const void *blocks, int n, hipStream_t stream);
int ffhip_launch_hevc_mc(int chroma, int mode, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                         const void *blocks, int n, hipStream_t stream);
int ffhip_launch_fdsp(int op, float *dst, size_t pd, const float *s0, size_t p0, const float *s1, size_t p1, const float *s2, size_t p2,
                      float mul, int len, int nvec, hipStream_t stream);
#endif
