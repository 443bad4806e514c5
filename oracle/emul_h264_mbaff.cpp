/*
 * emul_h264_mbaff.cpp — TEST INFRASTRUCTURE ONLY.  The CPU list executor of an MBAFF frame's two chains (ffmpeg_amd/csrc/h264_mbaff.hip):
 * the intra macroblocks through the SAME phase bodies the kernel runs (kernels/h264_intra_mb.h), each on a tile filled and written back
 * at the macroblock's own line step, in decoding order; then the recorded loop-filter calls through the oracle's restatement of the
 * h264dsp members (oracle/ffo_h264.c, pinned to the reference), in the order ff_h264_filter_mb() issued them.  tests/ compare the result
 * with the reference's own decode of the stream (tests/test_h264_stream_cpu.py), so the host side of the MBAFF path — the recorder's four
 * objects, the lists, the tile's geometry — is pinned where no GPU is present.
 */
#include <stdint.h>
#include <string.h>

#include "../ffmpeg_amd/csrc/kernels/h264_intra_mb.h"
extern "C" {
#include "ffo.h"
}

namespace {
struct EmulWave {
    template <class F>
    void run(F body)
    {
        for (int lane = 0; lane < 64; lane++)
            body(lane);
    }
};
} // namespace

/* the intra chain at one sample type (uint8_t, or uint16_t above 8 bits with int32 coefficients) */
template <typename PIX>
static int mbaff_intra(const FFHipH264MbaffLists *L, uint8_t *const dst[3], const int stride[3], int maxv)
{
    typedef typename ImbCoef<PIX>::T CF;
    constexpr int PS = (int)sizeof(PIX);
    EmulWave X;
    static uint32_t p4tab[IMB_TABS];
    for (int i = 0; i < IMB_TABS; i++)
        p4tab[i] = imb_tab(i);
    const ptrdiff_t sy = stride[0], sc = stride[1];
    const int mb_w = L->mb_w;
    long prev = -1;
    for (int k = 0; k < L->nrecs; k++) {
        const FFHipH264IntraMB &R = L->recs[k];
        const uint32_t g = L->geo[k];
        const int mx = (int)(g & 0xFFF), my = (int)((g >> 12) & 0xFFF), field = (int)(g >> 24) & 1, p = my >> 1;
        const long order = ((long)p * mb_w + mx) * 2 + (my & 1);
        if (mx >= mb_w || my >= L->mb_h || order <= prev)
            return -1;
        prev = order;
        ImbTileT<PIX> T;
        memset(&T, 0xA5, sizeof(T));
        memset(T.zero, 0, sizeof(T.zero));
        const int step = field ? 2 : 1, line0 = field ? 32 * p + (my & 1) : 16 * my, cline0 = field ? 16 * p + (my & 1) : 8 * my;
        const ptrdiff_t ysy = sy * step, csc = sc * step;
        uint8_t *ymb = dst[0] + (ptrdiff_t)line0 * sy + mx * 16 * PS;
        uint8_t *cmb[2] = { dst[1] + (ptrdiff_t)cline0 * sc + mx * 8 * PS, dst[2] + (ptrdiff_t)cline0 * sc + mx * 8 * PS };
        const bool has_l = mx > 0, has_t = line0 - step >= 0, has_r = mx + 1 < mb_w;
        for (int lane = 0; lane < 46; lane++) { /* the kernel's tile fill, quad by quad */
            PIX v[4] = { 0, 0, 0, 0 };
            if (lane < 8) {
                const int c = 4 * lane - 4;
                if (has_t && (c >= 0 || has_l) && (c < 16 || has_r))
                    memcpy(v, ymb - ysy + c * PS, 4 * PS);
                memcpy(&T.y[imb_yi(-1, c)], v, 4 * PS);
            } else if (lane < 24) {
                const int r = lane - 8;
                if (has_l)
                    memcpy(v, ymb + (ptrdiff_t)r * ysy - 4 * PS, 4 * PS);
                memcpy(&T.y[imb_yi(r, -4)], v, 4 * PS);
                memset(&T.y[imb_yi(r, 16)], 0, 8 * PS);
            } else if (lane < 30) {
                const int pl = (lane - 24) / 3, c = 4 * ((lane - 24) % 3) - 4;
                if (has_t && (c >= 0 || has_l))
                    memcpy(v, cmb[pl] - csc + c * PS, 4 * PS);
                memcpy(&T.c[pl][imb_ci(-1, c)], v, 4 * PS);
            } else {
                const int pl = (lane - 30) >> 3, r = (lane - 30) & 7;
                if (has_l)
                    memcpy(v, cmb[pl] + (ptrdiff_t)r * csc - 4 * PS, 4 * PS);
                memcpy(&T.c[pl][imb_ci(r, -4)], v, 4 * PS);
            }
        }
        imb_reconstruct<PIX>(X, T, R, reinterpret_cast<const CF *>(L->coefs + R.coef), p4tab, maxv, 3);
        for (int r = 0; r < 16; r++)
            memcpy(ymb + (ptrdiff_t)r * ysy, &T.y[imb_yi(r, 0)], 16 * PS);
        for (int pl = 0; pl < 2; pl++)
            for (int r = 0; r < 8; r++)
                memcpy(cmb[pl] + (ptrdiff_t)r * csc, &T.c[pl][imb_ci(r, 0)], 8 * PS);
    }
    return 0;
}

extern "C" int ffemul_h264_mbaff_flush(const FFHipH264MbaffLists *L, uint8_t *const dst[3], const int stride[3])
{
    if (!L || !dst || !stride)
        return -1;
    const int bd = L->bit_depth ? L->bit_depth : 8;
    const int r = bd > 8 ? mbaff_intra<uint16_t>(L, dst, stride, (1 << bd) - 1) : mbaff_intra<uint8_t>(L, dst, stride, 255);
    if (r < 0)
        return r;
    /* the in-loop filter: every plane's calls in order (planes are independent of each other) */
    for (int pl = 0; pl < 3; pl++)
        for (int i = 0; i < L->ncalls[pl]; i++) {
            const FFHipH264Edge &e = L->calls[pl][i];
            const int kind = e.kind & 7, chroma = (kind & 2) != 0, mbaff = (e.pad & FFHIP_H264_LF_CALL_MBAFF) != 0;
            const ptrdiff_t st = (e.pad & FFHIP_H264_LF_CALL_FIELD) ? 2 * (ptrdiff_t)stride[pl] : (ptrdiff_t)stride[pl];
            /* inner iterations per tc0 entry: luma 4, chroma (4:2:0) 2; the _mbaff members half of that (h264dsp_template.c:127-133,262-272) */
            const int inner = (chroma ? 2 : 4) >> (mbaff ? 1 : 0);
            ffo_h264_loop_filter_bd(bd, kind, inner, dst[pl] + e.offset, st, e.alpha, e.beta, e.tc0);
        }
    return 0;
}
