/* shim_rect.h — staging of host rectangles for the host-pointer faces (shims.hip, shims_h264_hbd.hip).  Internal to libffhip. */
#ifndef FFHIP_SHIM_RECT_H
#define FFHIP_SHIM_RECT_H

#include "shim_arena.h"

#define DP 64 /* device row pitch of a staged rectangle */

/* a rectangle rows r0..r1 x columns c0..c1 around host pointer p (row step = stride, may be negative) */
struct Rect {
    uint8_t *host;
    ptrdiff_t stride;
    int r0, r1, c0, c1;
    uint8_t *dev; /* device address corresponding to `host` */
};

static inline size_t rect_bytes(const Rect &r) { return (size_t)(r.r1 - r.r0 + 1) * DP + 2 * DP; }

static inline bool rect_up(Rect &r, uint8_t *buf)
{
    r.dev = buf + DP - (ptrdiff_t)r.r0 * DP - r.c0; /* row r0 col c0 lands at buf + DP */
    const int w = r.c1 - r.c0 + 1, h = r.r1 - r.r0 + 1;
    if (r.stride >= w) /* one 2-D copy; bottom-up pictures (negative strides) go row by row */
        return hipMemcpy2D(r.dev + (ptrdiff_t)r.r0 * DP + r.c0, DP, r.host + r.r0 * r.stride + r.c0, r.stride, w, h, hipMemcpyHostToDevice) ==
               hipSuccess;
    for (int y = r.r0; y <= r.r1; y++)
        if (hipMemcpy(r.dev + (ptrdiff_t)y * DP + r.c0, r.host + y * r.stride + r.c0, w, hipMemcpyHostToDevice) != hipSuccess)
            return false;
    return true;
}

/* commit rows r0..r1 x columns c0..c1 of a staged rectangle from the bounce buffer (after Arena::down()) */
static inline void rect_commit(const Arena &A, const Rect &r, int r0, int r1, int c0, int c1)
{
    for (int y = r0; y <= r1; y++)
        memcpy(r.host + y * r.stride + c0, A.host(r.dev + (ptrdiff_t)y * DP + c0), c1 - c0 + 1);
}
static inline void commit2d(const Arena &A, void *dst, ptrdiff_t dstride, const void *dev, ptrdiff_t dpitch, size_t wbytes, int rows)
{
    for (int y = 0; y < rows; y++)
        memcpy(static_cast<uint8_t *>(dst) + y * dstride, A.host(static_cast<const uint8_t *>(dev) + y * dpitch), wbytes);
}


#endif
