/*
 * h264_lf_line.h — one sample line of the H.264 in-loop filters (h264_{v,h}_loop_filter_{luma,chroma}[_intra]_<depth>_c,
 * libavcodec/h264dsp_template.c:104-330), shared by the frame-order kernels of h264_deblock.hip (8 bits) and the MBAFF kernels of
 * h264_mbaff.hip (any depth: alpha, beta and tc0 arrive scaled to the depth — alpha << (depth - 8), beta likewise, luma tc0 * (1 << (depth - 8)),
 * chroma ((tc0 - 1) << (depth - 8)) + 1 — and maxv = (1 << depth) - 1).
 */
#ifndef FFHIP_H264_LF_LINE_H
#define FFHIP_H264_LF_LINE_H
#include "common.h"

struct LfLine { int p3, p2, p1, p0, q0, q1, q2, q3; };

/* Filters one sample line in place; returns the mask of changed taps: bit0 p2, bit1 p1, bit2 p0, bit3 q0,
 * bit4 q1, bit5 q2.  cls: 0 luma, 1 chroma, 2 luma intra, 3 chroma intra. */
__device__ __forceinline__ int lf_line(LfLine &v, int cls, int alpha, int beta, int tc0, int maxv = 255)
{
    const int p0 = v.p0, p1 = v.p1, p2 = v.p2, q0 = v.q0, q1 = v.q1, q2 = v.q2;
    if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta)
        return 0;
    if (cls == 0) {
        if (tc0 < 0)
            return 0;
        int tc = tc0, m = 4 | 8;
        if (abs(p2 - p0) < beta) {
            if (tc0) {
                v.p1 = p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0);
                m |= 2;
            }
            tc++;
        }
        if (abs(q2 - q0) < beta) {
            if (tc0) {
                v.q1 = q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0);
                m |= 16;
            }
            tc++;
        }
        const int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
        v.p0 = min(max(p0 + delta, 0), maxv);
        v.q0 = min(max(q0 - delta, 0), maxv);
        return m;
    }
    if (cls == 1) {
        if (tc0 <= 0)
            return 0;
        const int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc0, tc0);
        v.p0 = min(max(p0 + delta, 0), maxv);
        v.q0 = min(max(q0 - delta, 0), maxv);
        return 4 | 8;
    }
    if (cls == 3) {
        v.p0 = (2 * p1 + p0 + q1 + 2) >> 2;
        v.q0 = (2 * q1 + q0 + p1 + 2) >> 2;
        return 4 | 8;
    }
    /* luma intra */
    int m = 4 | 8;
    if (abs(p0 - q0) < ((alpha >> 2) + 2)) {
        if (abs(p2 - p0) < beta) {
            v.p0 = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
            v.p1 = (p2 + p1 + p0 + q0 + 2) >> 2;
            v.p2 = (2 * v.p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
            m |= 1 | 2;
        } else {
            v.p0 = (2 * p1 + p0 + q1 + 2) >> 2;
        }
        if (abs(q2 - q0) < beta) {
            v.q0 = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
            v.q1 = (p0 + q0 + q1 + q2 + 2) >> 2;
            v.q2 = (2 * v.q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
            m |= 16 | 32;
        } else {
            v.q0 = (2 * q1 + q0 + p1 + 2) >> 2;
        }
    } else {
        v.p0 = (2 * p1 + p0 + q1 + 2) >> 2;
        v.q0 = (2 * q1 + q0 + p1 + 2) >> 2;
    }
    return m;
}

#endif
