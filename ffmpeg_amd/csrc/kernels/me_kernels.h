/* me_kernels.h — launchers of the me_cmp / exhaustive-search kernels (internal to libffhip). */
#ifndef FFHIP_ME_KERNELS_H
#define FFHIP_ME_KERNELS_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "ffhip.h"

int ffhip_launch_me_cmp(int kind, int width, int h, const uint8_t *blk1, const int32_t *off1, const uint8_t *blk2,
                        const int32_t *off2, ptrdiff_t stride, int32_t *out, int n, hipStream_t stream);
int ffhip_launch_me_esa(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride, size_t frame_pitch,
                        int nframes, int mb_size, int R, int cost_kind, int16_t *mv_out, uint32_t *cost_out,
                        hipStream_t stream);
/* SATD search on the matrix cores (me_satd.hip): 1 = launched, 0 = not its case */
int ffhip_launch_me_esa_satd_mx(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride, size_t frame_pitch, int nframes,
                                int mb_size, int R, int16_t *mv_out, uint32_t *cost_out, hipStream_t stream);
#endif
