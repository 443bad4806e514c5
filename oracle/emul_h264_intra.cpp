/*
 * emul_h264_intra.cpp — TEST INFRASTRUCTURE ONLY.  Runs the phase bodies of ffmpeg_amd/csrc/kernels/h264_intra_mb.h — the
 * code k_h264_intra_frame executes per macroblock on the GPU — on the CPU: a phase is a loop over the 64 lanes, the tile is
 * plain memory, macroblocks are walked in raster order (a valid serialisation of the kernel's wavefront).  tests/ compare it
 * with oracle/ffo_h264.c's restatement of hl_decode_mb() (itself pinned to the reference's ff_h264_hl_decode_mb), so the
 * kernel's per-macroblock logic is pinned where no GPU is present; the GPU tests then cover what is left — the hand-off between
 * rows and the tile's loads and stores.
 */
#include <stdint.h>
#include <string.h>

#include "../ffmpeg_amd/csrc/kernels/h264_intra_mb.h"

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

/* 1: the kernel's two-wavefront form: the whole picture's luma, then the whole picture's chroma; 2: the luma-only form (a plane of a
 * 4:4:4 picture: py is that plane, the records its luma-only ones, pcb / pcr are not touched) */
static int g_split;
extern "C" void ffemul_h264_intra_set_split(int on) { g_split = on; }

template <typename PIX>
static int intra_frame(uint8_t *py, uint8_t *pcb, uint8_t *pcr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h, const FFHipH264IntraMB *recs,
                       const int32_t *row_start, const int16_t *coefs, int maxv)
{
    typedef typename ImbCoef<PIX>::T CF;
    constexpr int PS = (int)sizeof(PIX), QB = 4 * PS; /* the kernel moves four samples per lane */
    EmulWave X;
    static uint32_t p4tab[IMB_TABS]; /* the kernel's copy lives in LDS */
    for (int i = 0; i < IMB_TABS; i++)
        p4tab[i] = imb_tab(i);
    for (int pass = 0; pass < (g_split == 1 ? 2 : 1); pass++)
    for (int my = 0, parts = g_split ? 1 + pass : 3; my < mb_h; my++)
        for (int k = row_start[my]; k < row_start[my + 1]; k++) {
            const FFHipH264IntraMB &R = recs[k];
            const int mx = R.mb_x;
            if (R.mb_y != my || mx < 0 || mx >= mb_w || (k > row_start[my] && recs[k - 1].mb_x >= mx))
                return -1;
            ImbTileT<PIX> T;
            memset(&T, 0xA5, sizeof(T)); /* whatever the phases do not write first must not matter */
            memset(T.zero, 0, sizeof(T.zero)); /* (the kernel: once per wave) */
            uint8_t *ymb = py + (ptrdiff_t)my * 16 * sy + mx * 16 * PS;
            uint8_t *cmb[2] = { pcb + (ptrdiff_t)my * 8 * sc + mx * 8 * PS, pcr + (ptrdiff_t)my * 8 * sc + mx * 8 * PS };
            const bool has_l = mx > 0, has_t = my > 0, has_r = mx + 1 < mb_w;
            /* the kernel's tile fill, quad by quad */
            for (int lane = 0; lane < 46; lane++) {
                uint64_t v = 0;
                if (lane < 8) {
                    const int c = 4 * lane - 4;
                    if (has_t && (c >= 0 || has_l) && (c < 16 || has_r))
                        memcpy(&v, ymb - sy + c * PS, QB);
                    memcpy(&T.y[imb_yi(-1, c)], &v, QB);
                } else if (lane < 24) {
                    const int r = lane - 8;
                    if (has_l)
                        memcpy(&v, ymb + (ptrdiff_t)r * sy - 4 * PS, QB);
                    memcpy(&T.y[imb_yi(r, -4)], &v, QB);
                    memset(&T.y[imb_yi(r, 16)], 0, 2 * QB);
                } else if (lane < 30) {
                    const int p = (lane - 24) / 3, c = 4 * ((lane - 24) % 3) - 4;
                    if (has_t && (c >= 0 || has_l))
                        memcpy(&v, cmb[p] - sc + c * PS, QB);
                    memcpy(&T.c[p][imb_ci(-1, c)], &v, QB);
                } else {
                    const int p = (lane - 30) >> 3, r = (lane - 30) & 7;
                    if (has_l)
                        memcpy(&v, cmb[p] + (ptrdiff_t)r * sc - 4 * PS, QB);
                    memcpy(&T.c[p][imb_ci(r, -4)], &v, QB);
                }
            }
            imb_reconstruct<PIX>(X, T, R, reinterpret_cast<const CF *>(coefs + R.coef), p4tab, maxv, parts);
            for (int r = 0; r < 16 && (parts & 1); r++)
                memcpy(ymb + (ptrdiff_t)r * sy, &T.y[imb_yi(r, 0)], 16 * PS);
            for (int p = 0; p < 2 && (parts & 2); p++)
                for (int r = 0; r < 8; r++)
                    memcpy(cmb[p] + (ptrdiff_t)r * sc, &T.c[p][imb_ci(r, 0)], 8 * PS);
        }
    return 0;
}

extern "C" int ffemul_h264_intra_frame(uint8_t *py, uint8_t *pcb, uint8_t *pcr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                                       const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs)
{
    return intra_frame<uint8_t>(py, pcb, pcr, sy, sc, mb_w, mb_h, recs, row_start, coefs, 255);
}

/* bit_depth 9 / 10 / 12 / 14: uint16_t samples, strides in bytes, runs of int32 coefficients */
extern "C" int ffemul_h264_intra_frame_bd(int bd, uint8_t *py, uint8_t *pcb, uint8_t *pcr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                                          const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs)
{
    if (bd == 8)
        return intra_frame<uint8_t>(py, pcb, pcr, sy, sc, mb_w, mb_h, recs, row_start, coefs, 255);
    return intra_frame<uint16_t>(py, pcb, pcr, sy, sc, mb_w, mb_h, recs, row_start, coefs, (1 << bd) - 1);
}

/* ---- 4:2:2: the chroma planes' wavefront (k_h264_intra_c422), macroblocks in raster order ---- */
template <typename PIX>
static int intra_c422_frame(uint8_t *pcb, uint8_t *pcr, ptrdiff_t sc, int mb_w, int mb_h, const FFHipH264IntraC422 *recs, const int32_t *row_start,
                            const int16_t *coefs, int maxv)
{
    typedef typename ImbCoef<PIX>::T CF;
    constexpr int PS = (int)sizeof(PIX);
    EmulWave X;
    for (int my = 0; my < mb_h; my++)
        for (int k = row_start[my]; k < row_start[my + 1]; k++) {
            const FFHipH264IntraC422 &R = recs[k];
            const int mx = R.mb_x;
            if (R.mb_y != my || mx < 0 || mx >= mb_w || (k > row_start[my] && recs[k - 1].mb_x >= mx))
                return -1;
            ImbTileC422<PIX> T;
            memset(&T, 0xA5, sizeof(T));
            memset(T.zero, 0, sizeof(T.zero));
            uint8_t *cmb[2] = { pcb + (ptrdiff_t)my * 16 * sc + mx * 8 * PS, pcr + (ptrdiff_t)my * 16 * sc + mx * 8 * PS };
            const bool has_l = mx > 0, has_t = my > 0;
            for (int p = 0; p < 2; p++) {
                for (int c = -4; c < 8; c++) { /* the row above, corner included; what lies outside the picture reads as 0 */
                    PIX v = 0;
                    if (has_t && (c >= 0 || has_l))
                        memcpy(&v, cmb[p] - sc + c * PS, PS);
                    T.c[p][imb_ci(-1, c)] = v;
                }
                for (int r = 0; r < 16; r++)
                    for (int c = -4; c < 0; c++) {
                        PIX v = 0;
                        if (has_l)
                            memcpy(&v, cmb[p] + (ptrdiff_t)r * sc + c * PS, PS);
                        T.c[p][imb_ci(r, c)] = v;
                    }
            }
            imb_c422_reconstruct<PIX>(X, T, R, reinterpret_cast<const CF *>(coefs + R.coef), maxv);
            for (int p = 0; p < 2; p++)
                for (int r = 0; r < 16; r++)
                    memcpy(cmb[p] + (ptrdiff_t)r * sc, &T.c[p][imb_ci(r, 0)], 8 * PS);
        }
    return 0;
}

extern "C" int ffemul_h264_intra_c422_frame_bd(int bd, uint8_t *pcb, uint8_t *pcr, ptrdiff_t sc, int mb_w, int mb_h, const FFHipH264IntraC422 *recs,
                                               const int32_t *row_start, const int16_t *coefs)
{
    if (bd == 8)
        return intra_c422_frame<uint8_t>(pcb, pcr, sc, mb_w, mb_h, recs, row_start, coefs, 255);
    return intra_c422_frame<uint16_t>(pcb, pcr, sc, mb_w, mb_h, recs, row_start, coefs, (1 << bd) - 1);
}
