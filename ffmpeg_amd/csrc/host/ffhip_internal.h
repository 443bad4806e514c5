/* ffhip_internal.h — declarations shared between the plain-C host side and the HIP side of libffhip. */
#ifndef FFHIP_INTERNAL_H
#define FFHIP_INTERNAL_H
#include <stddef.h>
#include <stdint.h>
#include "ffhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* printf-style; stores the message returned by ffhip_last_error() (thread-local) */
void ffhip_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

/* host/sws_tables.c */
int  ffhip_host_init_filter(int16_t **out_filter, int32_t **out_pos, int xInc, int srcW, int dstW, int one,
                            int scaler, int flags);
void ffhip_host_yuv2rgb_coeffs(FFHipSwsTables *t, int fullRange);
/* the YUV formats above 8 bits: sample depth, layout (0 planar, samples in the low bits; 1 semi-planar, samples in the high bits),
 * chroma subsampling shifts.  Returns 0 for any other format (outputs untouched). */
int  ffhip_pixfmt_hbd(int fmt, int *depth, int *layout, int *hsub, int *vsub);
/* init_range_convert_constants() (libswscale/swscale.c:591-624) for a source of range `src_range` (1 full) going to the other one */
int ffhip_sws_rgb_source_plan(int srcW, int srcH, int srcFormat, int dstW, int dstH, int dstFormat, int flags, int *half, int32_t table[9],
                              int *bpp, int ofs[3]);
void ffhip_sws_range_constants(int src_range, int dst_depth, uint32_t *lum_coeff, int64_t *lum_offset, uint32_t *chr_coeff, int64_t *chr_offset);

#ifdef __cplusplus
}
#endif
#endif
