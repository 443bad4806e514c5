/*
 * ffo_fdsp.c — CPU restatement of the AVFloatDSPContext vector operations that sit on either side of the MDCT
 * (SURVEY.md §8 f-4).  TEST INFRASTRUCTURE ONLY.
 *
 * Follows libavutil/float_dsp.c: vector_fmul_c :27, vector_fmac_scalar_c :43, vector_fmul_scalar_c :59,
 * vector_fmul_window_c :75, vector_fmul_add_c :95, vector_fmul_reverse_c :103, butterflies_float_c :113.
 * Each output is one or two IEEE single-precision multiplies and at most one add/sub, in the order written there;
 * built with -ffp-contract=off, so there is no fused multiply-add anywhere.
 */
#include "ffo.h"

void ffo_fdsp(int op, float *dst, const float *src0, const float *src1, const float *src2, float mul, int len)
{
    switch (op) {
    case FFO_FDSP_FMUL:
        for (int i = 0; i < len; i++)
            dst[i] = src0[i] * src1[i];
        break;
    case FFO_FDSP_FMAC_SCALAR:
        for (int i = 0; i < len; i++)
            dst[i] += src0[i] * mul;
        break;
    case FFO_FDSP_FMUL_SCALAR:
        for (int i = 0; i < len; i++)
            dst[i] = src0[i] * mul;
        break;
    case FFO_FDSP_FMUL_WINDOW: /* dst[2 len] from src0[len], src1[len], win = src2[2 len] */
        for (int i = 0; i < len; i++) {
            const int j = 2 * len - 1 - i;
            const float s0 = src0[i], s1 = src1[len - 1 - i], wi = src2[i], wj = src2[j];
            dst[i] = s0 * wj - s1 * wi;
            dst[j] = s0 * wi + s1 * wj;
        }
        break;
    case FFO_FDSP_FMUL_ADD:
        for (int i = 0; i < len; i++)
            dst[i] = src0[i] * src1[i] + src2[i];
        break;
    case FFO_FDSP_FMUL_REVERSE:
        for (int i = 0; i < len; i++)
            dst[i] = src0[i] * src1[len - 1 - i];
        break;
    case FFO_FDSP_BUTTERFLIES: /* in place on (dst, src0 treated as the second vector, written) */
        {
            float *v1 = dst, *v2 = (float *)src0;
            for (int i = 0; i < len; i++) {
                const float t = v1[i] - v2[i];
                v1[i] += v2[i];
                v2[i] = t;
            }
        }
        break;
    }
}
