/*
 * me_api.hip — C-ABI entry points of the me_cmp / motion-search part of libffhip (include/ffhip.h).
 */
#include "kernels/common.h"
#include "kernels/me_kernels.h"

extern "C" int ffhip_me_cmp_batch_dev(int kind, int width, int h, const uint8_t *blk1, const int32_t *off1,
                                      const uint8_t *blk2, const int32_t *off2, ptrdiff_t stride, int32_t *out, int n,
                                      void *stream)
{
    if (!blk1 || !blk2 || !off1 || !off2 || !out || n < 0 || (width != 16 && width != 8) || h <= 0 ||
        kind < FFHIP_ME_SAD || kind > FFHIP_ME_NSSE)
        return FFHIP_EINVAL;
    if (kind == FFHIP_ME_SATD && h != 8 && h != 16) /* hadamard8_diff16_c handles h 8|16 only (me_cmp.c:933-950) */
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_me_cmp(kind, width, h, blk1, off1, blk2, off2, stride, out, n, (hipStream_t)stream);
}

extern "C" int ffhip_me_esa_batch_dev(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride,
                                      size_t frame_pitch, int nframes, int mb_size, int search_param, int cost_kind,
                                      int16_t *mv_out, uint32_t *cost_out, void *stream)
{
    if (!cur || !ref || !mv_out || !cost_out || width <= 0 || height <= 0 || nframes < 0 || search_param < 0 ||
        (mb_size != 16 && mb_size != 8) || (cost_kind != FFHIP_ME_SAD && cost_kind != FFHIP_ME_SATD))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_me_esa(cur, ref, width, height, stride, frame_pitch, nframes, mb_size, search_param, cost_kind,
                               mv_out, cost_out, (hipStream_t)stream);
}
