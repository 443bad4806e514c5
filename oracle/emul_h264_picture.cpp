/*
 * emul_h264_picture.cpp — TEST INFRASTRUCTURE ONLY.  A CPU "flush" of an FFHipH264Picture: the lists ffhip_h264_picture_lists() exports
 * (what the FFmpeg-side recorder, integration/avcodec_h264_picture_hip.c, left in the object while the reference's own
 * ff_h264_hl_decode_mb() / ff_h264_filter_mb() ran over its recording members) are executed in ffhip_h264_picture_flush()'s stage order
 *
 *     per plane:  MC put -> picture | MC put -> bi-prediction scratch | MC avg -> picture | weight / biweight | IDCT + add |
 *                 intra macroblocks (the kernel's per-macroblock phases, emul_h264_intra.cpp) | deblock (frame order)
 *
 * with the ORACLE's dsp functions (oracle/ffo_h264*.c, pinned to the reference) on host planes.  tests/test_h264_picture_cpu.py compares
 * the result with the reference's own decode of the same decoder state: that pins, where no GPU is present, everything of the picture
 * layer that is host logic — which member the recorder turns into which record, offsets, the FFHIP_MC_EMU coordinates, scratchpad
 * bookkeeping, the residual dispatch, the intra packing, the edge tables — at 4:2:0 and 4:4:4, 8 bits and above.  What is left to the
 * GPU tests is the kernels' arithmetic on those lists (which the function-level GPU tests pin separately).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "ffhip.h"
extern "C" {
#include "ffo.h"
void ffemul_h264_intra_set_split(int on);
int ffemul_h264_intra_frame_bd(int bd, uint8_t *py, uint8_t *pcb, uint8_t *pcr, ptrdiff_t sy, ptrdiff_t sc, int mb_w, int mb_h,
                               const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs);
int ffemul_h264_intra_c422_frame_bd(int bd, uint8_t *pcb, uint8_t *pcr, ptrdiff_t sc, int mb_w, int mb_h, const FFHipH264IntraC422 *recs,
                                    const int32_t *row_start, const int16_t *coefs);
}

static_assert(sizeof(FfoH264Edge) == sizeof(FFHipH264Edge), "one edge record layout");

namespace {
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* FFHIP_MC_EMU (include/ffhip.h): footprint sample (x, y) of the window whose first sample is (x0, y0) of the reference picture is read
 * at row clamp(y), column clamp(x).  The window lands in `tmp` at the DESTINATION's row pitch (the dsp functions take one stride). */
const uint8_t *emu_window(std::vector<uint8_t> &tmp, const uint8_t *pic00, ptrdiff_t stride, int px, int x0, int y0, int w, int h, int pic_w, int pic_h)
{
    tmp.assign((size_t)h * (size_t)stride + 64, 0);
    for (int j = 0; j < h; j++) {
        const uint8_t *row = pic00 + (ptrdiff_t)clampi(y0 + j, 0, pic_h - 1) * stride;
        for (int i = 0; i < w; i++)
            memcpy(&tmp[(size_t)j * stride + (size_t)i * px], row + (size_t)clampi(x0 + i, 0, pic_w - 1) * px, (size_t)px);
    }
    return tmp.data();
}
} // namespace

/* dst / ref: host planes (ref[pl] = the base the records' src_offset counts from); stride in bytes.  Returns 0, or -1 for lists this
 * executor does not know how to run. */
extern "C" int ffemul_h264_picture_flush(const FFHipH264PictureLists *L, uint8_t *const dst[3], const int stride[3], const uint8_t *const ref[3])
{
    const int bd = L->bit_depth, px = bd > 8 ? 2 : 1, wide = px, c444 = L->chroma_format_idc == 3, c422 = L->chroma_format_idc == 2;
    if (L->chroma_format_idc < 1 || L->chroma_format_idc > 3)
        return -1;
    const int pw[3] = { 16 * L->mb_w, (c444 ? 16 : 8) * L->mb_w, (c444 ? 16 : 8) * L->mb_w };
    const int ph[3] = { 16 * L->mb_h, (c444 || c422 ? 16 : 8) * L->mb_h, (c444 || c422 ? 16 : 8) * L->mb_h };
    std::vector<uint8_t> scratch[3], win;
    for (int pl = 0; pl < 3; pl++)
        scratch[pl].assign((size_t)ph[pl] * (size_t)stride[pl] + 64, 0xCD);
    /* ---- prediction ---- */
    for (int st = 0; st < 3; st++)
        for (int pl = 0; pl < 3; pl++) {
            uint8_t *target = st == FFHIP_H264_MC_TMP ? scratch[pl].data() : dst[pl];
            for (int i = 0; i < L->nqpel[pl][st]; i++) {
                const FFHipQpelBlock &q = L->qpel[pl][st][i];
                const int n = 16 >> q.size_idx;
                const uint8_t *src = ref[pl] + q.src_offset;
                if (q.flags & FFHIP_MC_EMU)
                    src = emu_window(win, src, stride[pl], px, q.src_x - 2, q.src_y - 2, n + 5, n + 5, pw[pl], ph[pl]) + 2 * stride[pl] + 2 * px;
                if ((q.avg != 0) != (st == FFHIP_H264_MC_AVG))
                    return -1;
                ffo_h264_qpel_bd(bd, q.avg, q.size_idx, q.mcxy, target + q.dst_offset, src, stride[pl]);
            }
            if (!pl)
                continue;
            for (int i = 0; i < L->ncmc[pl - 1][st]; i++) {
                const FFHipChromaBlock &c = L->cmc[pl - 1][st][i];
                const int w = 8 >> c.w_idx;
                const uint8_t *src = ref[pl] + c.src_offset;
                if (c.flags & FFHIP_MC_EMU)
                    src = emu_window(win, src, stride[pl], px, c.src_x, c.src_y, w + 1, c.h + 1, pw[pl], ph[pl]);
                if ((c.avg != 0) != (st == FFHIP_H264_MC_AVG))
                    return -1;
                ffo_h264_chroma_mc_bd(bd, c.avg, w, target + c.dst_offset, src, stride[pl], c.h, c.x, c.y);
            }
        }
    for (int pl = 0; pl < 3; pl++)
        for (int i = 0; i < L->nwt[pl]; i++) {
            const FFHipWeightBlock &w = L->wt[pl][i];
            const int width = 16 >> w.w_idx;
            if (w.bi)
                ffo_h264_biweight_bd(bd, width, dst[pl] + w.dst_offset, scratch[pl].data() + w.src_offset, stride[pl], w.height, w.log2_denom, w.weightd,
                                     w.weights, w.offset);
            else
                ffo_h264_weight_bd(bd, width, dst[pl] + w.dst_offset, stride[pl], w.height, w.log2_denom, w.weightd, w.offset);
        }
    /* ---- residual ---- */
    for (int pl = 0; pl < 3; pl++)
        for (int k = 0; k < 4; k++) {
            const int nc = (k == FFHIP_H264_IDCT8 || k == FFHIP_H264_IDCT8_DC ? 64 : 16) * wide;
            for (int i = 0; i < L->nidct[pl][k]; i++) {
                int16_t blk[128];
                memcpy(blk, L->idct_coef[pl][k] + (size_t)i * nc, sizeof(int16_t) * nc);
                ffo_h264_idct_bd(bd, k, dst[pl] + L->idct_off[pl][k][i], blk, stride[pl]);
            }
        }
    /* (the lossless bypass of inter macroblocks: add_pixels4 / 8_clear) */
    for (int pl = 0; pl < 3; pl++)
        for (int k = 0; k < 2; k++) {
            const int nc = (k ? 64 : 16) * wide;
            for (int i = 0; i < L->naddpx[pl][k]; i++) {
                int16_t blk[128];
                memcpy(blk, L->addpx_coef[pl][k] + (size_t)i * nc, sizeof(int16_t) * nc);
                ffo_h264_idct_bd(bd, FFHIP_H264_ADD_PIXELS4_CLEAR + k, dst[pl] + L->addpx_off[pl][k][i], blk, stride[pl]);
            }
        }
    /* ---- intra macroblocks: sorted by (mb_y, mb_x) as flush() sorts them, through the kernel's per-macroblock phases ---- */
    for (int q = 0; q < (c444 ? 3 : 1); q++) {
        if (!L->nintra[q])
            continue;
        std::vector<FFHipH264IntraMB> recs(L->intra[q], L->intra[q] + L->nintra[q]);
        std::stable_sort(recs.begin(), recs.end(), [](const FFHipH264IntraMB &a, const FFHipH264IntraMB &b) {
            return a.mb_y != b.mb_y ? a.mb_y < b.mb_y : a.mb_x < b.mb_x;
        });
        std::vector<int32_t> rows((size_t)L->mb_h + 1, 0);
        for (const FFHipH264IntraMB &r : recs)
            rows[(size_t)r.mb_y + 1]++;
        for (int r = 0; r < L->mb_h; r++)
            rows[(size_t)r + 1] += rows[r];
        std::vector<int16_t> coefs(L->intra_coef[q], L->intra_coef[q] + L->nintra_coef[q]);
        coefs.resize(coefs.size() + 1024, 0); /* the kernel's run fetch is sized by the macroblock type, not by the run */
        int r;
        if (c444 || c422) { /* luma-only records: 4:4:4 every plane, 4:2:2 the luma plane */
            ffemul_h264_intra_set_split(2);
            r = ffemul_h264_intra_frame_bd(bd, dst[q], dst[q], dst[q], stride[q], stride[q], L->mb_w, L->mb_h, recs.data(), rows.data(), coefs.data());
            ffemul_h264_intra_set_split(0);
        } else {
            r = ffemul_h264_intra_frame_bd(bd, dst[0], dst[1], dst[2], stride[0], stride[1], L->mb_w, L->mb_h, recs.data(), rows.data(), coefs.data());
        }
        if (r)
            return -1;
    }
    if (L->nintra_c422) { /* 4:2:2: the chroma planes' records through k_h264_intra_c422's phase body */
        std::vector<FFHipH264IntraC422> recs(L->intra_c422, L->intra_c422 + L->nintra_c422);
        std::stable_sort(recs.begin(), recs.end(), [](const FFHipH264IntraC422 &a, const FFHipH264IntraC422 &b) {
            return a.mb_y != b.mb_y ? a.mb_y < b.mb_y : a.mb_x < b.mb_x;
        });
        std::vector<int32_t> rows((size_t)L->mb_h + 1, 0);
        for (const FFHipH264IntraC422 &r : recs)
            rows[(size_t)r.mb_y + 1]++;
        for (int r = 0; r < L->mb_h; r++)
            rows[(size_t)r + 1] += rows[r];
        if (!c422 || stride[1] != stride[2] ||
            ffemul_h264_intra_c422_frame_bd(bd, dst[1], dst[2], stride[1], L->mb_w, L->mb_h, recs.data(), rows.data(), L->intra_c422_coef))
            return -1;
    }
    /* ---- in-loop filter, decoder order ---- */
    for (int pl = 0; pl < 3; pl++)
        if (L->edges[pl]) {
            if (pl && c422)
                ffo_h264_deblock_frame_c422_bd(bd, dst[pl], stride[pl], L->mb_w, L->mb_h, reinterpret_cast<const FfoH264Edge *>(L->edges[pl]));
            else
                ffo_h264_deblock_frame_bd(bd, pl && !c444, dst[pl], stride[pl], L->mb_w, L->mb_h, reinterpret_cast<const FfoH264Edge *>(L->edges[pl]));
        }
    return 0;
}
