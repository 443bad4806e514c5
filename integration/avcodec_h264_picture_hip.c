/*
 * integration/avcodec_h264_picture_hip.c — libavcodec/hip/h264_picture.c of the FFmpeg-side patch: the H.264 macroblock loop
 * recorded into a libffhip picture object.
 *
 * How.  hl_decode_mb() (libavcodec/h264_mb_template.c:41-270) and ff_h264_filter_mb() (libavcodec/h264_loopfilter.c:716) reach the
 * pixels only through function pointers: h->h264qpel / h->h264chroma / h->h264dsp / h->vdsp.  The `hip` arch therefore does not
 * restate hl_motion() / mc_part() / mc_dir_part() / filter_mb_dir(): it installs members that RECORD their operands, and the
 * decoder's own code runs unchanged on top of them — which partition calls which table entry with which motion vector, reference,
 * weight, bS, alpha, beta and tc0 is decided by the reference's code, not by a copy of it.  What a member gets:
 *
 *   qpel_mc_func(dst, src, stride)                   -> FFHipQpelBlock   (table index = size, mcXY)
 *   h264_chroma_mc_func(dst, src, stride, h, x, y)   -> FFHipChromaBlock
 *   weight / biweight(dst[, src], stride, h, ...)    -> FFHipWeightBlock
 *   idct_add16 / idct8_add4 / idct_add8(dst, block_offset, block, stride, nnzc) -> ffhip_h264_picture_idct_mb()
 *   vdsp.emulated_edge_mc(buf, src, ..., src_x, src_y, w, h)  -> remembered; the qpel / chroma call that then reads `buf`
 *                                                       (sl->edge_emu_buffer) becomes a record flagged FFHIP_MC_EMU that carries
 *                                                       the block's position in the reference picture: no border, no copy
 *   h264_{v,h}_loop_filter_{luma,chroma}[_intra](pix, stride, alpha, beta[, tc0]) -> the macroblock's FFHipH264Edge records
 *
 * Addresses are never dereferenced here: the current picture and the references are hip frames, h->cur_pic.f->data[] and
 * H264Ref.data[] hold DEVICE addresses, and a member turns them into offsets from the plane bases.  A prediction put into
 * sl->bipred_scratchpad (mc_part_weighted(), h264_mb.c:407-419) is held back until the biweight call that follows names the block it
 * belongs to: the picture object keeps a picture-sized scratch plane per plane and addresses it with the destination's offset.
 *
 * Intra macroblocks predict from reconstructed neighbours and cannot be a list of independent calls: they go to libffhip as ONE
 * record each (ffhip_h264_picture_intra_mb(), the reconstruction wavefront), built from the H264SliceContext fields hl_decode_mb()
 * would read.  4:2:0 and 4:4:4 (hl_decode_mb_444: the luma members on all three planes — the recorder only has to accept them there),
 * frame macroblocks — or the field macroblocks of a FIELD picture (PAFF): a field is every second line of the frame buffer, i.e. a
 * picture of half the height at twice the line size, which is how hl_decode_mb() itself addresses it (mb_linesize = 2 * linesize,
 * block_offset[48..], odd rows one line down: h264_mb_template.c:61-78) and how the references' fields arrive (pic_as_field(),
 * h264_refs.c:39-48); the libffhip picture object is made for the field and never learns the difference.  CAVLC / CABAC alike (entropy
 * decoding stays on the CPU and fills sl-> as ever).
 *
 * MBAFF frames (frame and field macroblock pairs mixed in one picture; round 6; 4:2:0, 8 - 14 bits).  A field macroblock of a pair IS the field
 * case, per macroblock: hl_decode_mb() addresses it at twice the line size from the pair's first or second line (h264_mb_template.c:65-73)
 * and reads its references through the field entries of the list (ref_list[l][16 + 2 i + parity], h264_refs.c h264_fill_mbaff_ref_list; the
 * cache rewritten at h264_mb_template.c:74-91).  The recorder therefore keeps THREE picture objects over the same planes — the frame
 * macroblocks, the top-field ones, the bottom-field ones — and switches between them per macroblock; nothing below the switch knows.  What
 * does not split that way goes into a fourth object (FFHipH264Mbaff): the intra macroblocks (they read their neighbours at the macroblock's
 * own line step) and the loop filter, whose dsp calls in an MBAFF frame (the left edge in two halves through the _mbaff members, a frame
 * macroblock's top edge once per field at twice the line size: h264_loopfilter.c:494-560,728-830) are recorded as what they are: calls.
 */
#include <string.h>

#include "libavutil/attributes.h"
#include "libavutil/common.h"
#include "libavcodec/h264dec.h"
#include "libavcodec/h264_ps.h"
#include "libavcodec/mpegutils.h"

#include "avcodec_h264_picture_hip.h"

static _Thread_local FFHipH264Recorder *cur_rec;   /* function pointers carry no user data: the recorder of the running call */

#define REC FFHipH264Recorder *r = cur_rec
#define FAIL(e) do { if (r->error >= 0) r->error = (e); return; } while (0)

/* which plane of the current picture / of the scratchpad an address lies in: 0..2, 16 + plane for the scratchpad, -1 neither */
/* bytes from a plane's first sample to the end of its last row (a field's last row ends one frame line before rows * linesize would) */
static size_t plane_span(const FFHipH264Recorder *r, int pl)
{
    return (size_t)(r->rows[pl] - 1) * r->linesize[pl] + ((size_t)r->pic_w[pl] << r->pixel_shift);
}

/* MBAFF: the object the macroblock at hand is recorded into (0 the frame macroblocks, 1 / 2 the top- / bottom-field ones) */
static void select_view(FFHipH264Recorder *r, int v)
{
    if (r->active == v)
        return;
    r->active = v;
    r->pic = r->view[v].pic;
    r->field = v > 0;
    for (int pl = 0; pl < 3; pl++) {
        r->cur[pl] = r->view[v].cur[pl];
        r->linesize[pl] = r->view[v].linesize[pl];
        r->rows[pl] = r->view[v].rows[pl];
        r->last_ref[pl] = NULL;   /* (an origin found at one line size says nothing at the other) */
    }
    r->scratch_size = (size_t)16 * r->linesize[1] + (size_t)16 * r->linesize[0];
    r->emu_size = (size_t)21 * r->linesize[0] + ((size_t)21 << r->pixel_shift);
}

static int classify_dst(const FFHipH264Recorder *r, const uint8_t *p)
{
    for (int pl = 0; pl < 3; pl++)
        if (p >= r->cur[pl] && p < r->cur[pl] + plane_span(r, pl))
            return pl;
    if (p >= r->scratch && p < r->scratch + r->scratch_size) {
        /* tmp_cb = scratch, tmp_cr = scratch + (8 << pixel_shift + (chroma_idc == 3)), tmp_y = scratch + 16 * mb_uvlinesize
         * (h264_mb.c:388-390) */
        const size_t o = (size_t)(p - r->scratch);
        if (o >= (size_t)16 * r->linesize[1])
            return 16;
        return (o % (size_t)r->linesize[1]) >= ((size_t)(r->cfmt == 3 ? 16 : 8) << r->pixel_shift) ? 18 : 17; /* (4:2:2: 8 wide, 16 rows) */
    }
    return -1;
}

/* src of a qpel / chroma call -> (src_offset, flags, src_x, src_y).  Luma (need > 0): the batch kernels fetch a block's whole
 * (size + 5)^2 footprint whatever the quarter-sample position, while mc_dir_part() asks for emulation only where the position's own
 * taps leave the picture (`if (mx & 7) extra_width -= 3`, h264_mb.c:229-236) — an integer-position block may sit flush against the
 * picture's edge unemulated.  Such a block is recorded as FFHIP_MC_EMU as well: the clamped fetch returns the same samples inside
 * the picture, and the ones outside carry no weight at that position. */
/* The records address the references by a 32-bit offset from ONE base per plane (ffhip.h: FFHipQpelBlock.src_offset; flush()'s ref[]): the
 * current picture and every reference it uses must lie within 2 GiB of that base — one decoded-picture-buffer allocation (or a pool
 * carved from one), which is how integration/avutil_hwcontext_hip.c's frame pool is meant to be set up for a decoder.  A reference
 * outside that reach (separate far-apart allocations, a DPB beyond 2 GiB) is REFUSED here, not wrapped: the picture then fails with
 * FFHIP_EINVAL at its first such block instead of reading a wild device address. */
static int fits32(ptrdiff_t d, int32_t *out)
{
    if (d < INT32_MIN || d > INT32_MAX)
        return 0;
    *out = (int32_t)d;
    return 1;
}

static int locate_src(FFHipH264Recorder *r, int pl, const uint8_t *src, int need, int32_t *off, uint8_t *flags, int16_t *sx, int16_t *sy)
{
    if (r->emu.valid && src >= r->emu_buf && src < r->emu_buf + r->emu_size) {
        /* the block sits at (col, row) of the window emulated_edge_mc() was asked to copy; the window's first sample is
         * (emu.src_x, emu.src_y) of the reference picture, and emu.src the address that sample would have */
        const size_t o = (size_t)(src - r->emu_buf);
        const int row = (int)(o / (size_t)r->emu.linesize), col = (int)(o % (size_t)r->emu.linesize) >> r->pixel_shift;
        const uint8_t *origin = r->emu.src - ((ptrdiff_t)r->emu.src_y * r->emu.linesize + ((ptrdiff_t)r->emu.src_x << r->pixel_shift));
        /* (the block's own coordinates travel as int16: motion vectors are bounded well inside that, h264_mvpred.h / level limits) */
        if (!fits32(origin - r->ref_base[pl], off) || r->emu.src_x + col < INT16_MIN || r->emu.src_x + col > INT16_MAX ||
            r->emu.src_y + row < INT16_MIN || r->emu.src_y + row > INT16_MAX)
            return FFHIP_EINVAL;
        *flags = FFHIP_MC_EMU;
        *sx    = (int16_t)(r->emu.src_x + col);
        *sy    = (int16_t)(r->emu.src_y + row);
        return 0;
    }
    if (!fits32(src - r->ref_base[pl], off))
        return FFHIP_EINVAL;
    *flags = 0;
    *sx = *sy = 0;
    if (need > 0) {
        /* which reference picture is it?  the one looked at last, nearly always */
        const ptrdiff_t ls = r->linesize[pl], span = (ptrdiff_t)plane_span(r, pl);
        /* (the two fields of one frame interleave in memory: a sample belongs to the field in whose lines it lies — its column, taken at
         * the field's line size, falls inside the picture's width) */
        const ptrdiff_t wbytes = (ptrdiff_t)r->pic_w[pl] << r->pixel_shift;
        const uint8_t *origin = r->last_ref[pl];
        if (!origin || src < origin || src >= origin + span || (src - origin) % ls >= wbytes) {
            /* a field macroblock of an MBAFF frame reads the list's field entries: 16 + 2 i + parity (h264_mb_template.c:74-91) */
            const int first = r->mbaff && r->field ? 16 : 0;
            origin = NULL;
            for (int l = 0; l < (int)r->sl->list_count && !origin; l++) {
                const int n = first ? FFMIN(2 * (int)r->sl->ref_count[l], 32) : (int)r->sl->ref_count[l];
                for (int i = 0; i < n && !origin; i++) {
                    const uint8_t *d = r->sl->ref_list[l][first + i].data[pl];
                    if (d && src >= d && src < d + span && (src - d) % ls < wbytes)
                        origin = d;
                }
            }
            if (!origin)
                return FFHIP_EINVAL;
            r->last_ref[pl] = origin;
        }
        {
            const ptrdiff_t o = src - origin;
            const int y = (int)(o / ls), x = (int)(o % ls) >> r->pixel_shift;
            if (x < 2 || y < 2 || x + need + 3 > r->pic_w[pl] || y + need + 3 > r->rows[pl]) {
                if (!fits32(origin - r->ref_base[pl], off))
                    return FFHIP_EINVAL;
                *flags = FFHIP_MC_EMU;
                *sx    = (int16_t)x;
                *sy    = (int16_t)y;
            }
        }
    }
    return 0;
}

static void rec_qpel(int avg, int size_idx, int mcxy, uint8_t *dst, const uint8_t *src, ptrdiff_t stride)
{
    REC;
    FFHipQpelBlock q = { 0 };
    int where = classify_dst(r, dst), pl = where & 15, rc;
    /* the luma tables reach Cb / Cr in a 4:4:4 picture only (mc_dir_part(), h264_mb.c:262-288); field macroblocks (doubled stride) stay on
     * the C path */
    if (where < 0 || (pl != 0 && r->cfmt != 3) || stride != r->linesize[pl])
        FAIL(FFHIP_EINVAL);
    rc = locate_src(r, pl, src, 16 >> size_idx, &q.src_offset, &q.flags, &q.src_x, &q.src_y);
    if (rc < 0)
        FAIL(rc);
    q.mcxy = (uint8_t)mcxy;
    q.size_idx = (uint8_t)size_idx;
    if (where >= 16) {
        if (r->npend >= 8)
            FAIL(FFHIP_EINVAL);
        r->pend[r->npend].plane = pl;
        r->pend[r->npend].luma_tab = 1;
        r->pend[r->npend].tmp = dst;
        r->pend[r->npend].q = q;
        r->npend++;
        return;
    }
    q.dst_offset = (int32_t)(dst - r->cur[pl]);
    rc = ffhip_h264_picture_mc_luma_plane(r->pic, pl, avg ? FFHIP_H264_MC_AVG : FFHIP_H264_MC_PUT, &q);
    if (rc < 0)
        FAIL(rc);
}

static void rec_chroma(int avg, int w_idx, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int h, int x, int y)
{
    REC;
    FFHipChromaBlock c = { 0 };
    int where = classify_dst(r, dst), pl = where & 15, rc;
    if (where < 0 || pl < 1 || r->cfmt == 3 || stride != r->linesize[pl])
        FAIL(FFHIP_EINVAL);
    rc = locate_src(r, pl, src, 0, &c.src_offset, &c.flags, &c.src_x, &c.src_y);
    if (rc < 0)
        FAIL(rc);
    c.w_idx = (uint8_t)w_idx;
    c.h = (uint8_t)h;
    c.x = (uint8_t)x;
    c.y = (uint8_t)y;
    if (where >= 16) {
        if (r->npend >= 8)
            FAIL(FFHIP_EINVAL);
        r->pend[r->npend].plane = pl;
        r->pend[r->npend].luma_tab = 0;
        r->pend[r->npend].tmp = dst;
        r->pend[r->npend].c = c;
        r->npend++;
        return;
    }
    c.dst_offset = (int32_t)(dst - r->cur[pl]);
    rc = ffhip_h264_picture_mc_chroma(r->pic, pl, avg ? FFHIP_H264_MC_AVG : FFHIP_H264_MC_PUT, &c);
    if (rc < 0)
        FAIL(rc);
}

static void rec_weight(int w_idx, int bi, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd, int weights,
                       int offset)
{
    REC;
    FFHipWeightBlock w = { 0 };
    int pl = classify_dst(r, dst), rc;
    if (pl < 0 || pl > 2 || stride != r->linesize[pl])
        FAIL(FFHIP_EINVAL);
    w.dst_offset = w.src_offset = (int32_t)(dst - r->cur[pl]);
    w.w_idx = (uint8_t)w_idx;
    w.height = (uint8_t)height;
    w.log2_denom = (uint8_t)log2_denom;
    w.bi = (uint8_t)bi;
    w.weightd = (int16_t)weightd;
    w.weights = (int16_t)weights;
    w.offset = (int16_t)offset;
    if (bi) {
        /* the predictions waiting in the scratchpad belong here: same row pitch, so an address difference is an offset difference */
        int k = 0;
        for (int i = 0; i < r->npend; i++) {
            struct FFHipH264Pending *p = &r->pend[i];
            if (p->plane != pl) {
                r->pend[k++] = *p;
                continue;
            }
            if (p->luma_tab) {
                p->q.dst_offset = w.dst_offset + (int32_t)(p->tmp - src);
                rc = ffhip_h264_picture_mc_luma_plane(r->pic, pl, FFHIP_H264_MC_TMP, &p->q);
            } else {
                p->c.dst_offset = w.dst_offset + (int32_t)(p->tmp - src);
                rc = ffhip_h264_picture_mc_chroma(r->pic, pl, FFHIP_H264_MC_TMP, &p->c);
            }
            if (rc < 0)
                FAIL(rc);
        }
        r->npend = k;
    }
    rc = ffhip_h264_picture_weight(r->pic, pl, &w);
    if (rc < 0)
        FAIL(rc);
}

/* ---- the table members ------------------------------------------------------------------------------------------------------- */
#define QP1(op, avg, sz, si, mc) static void op##_qpel##sz##_mc##mc(uint8_t *d, const uint8_t *s, ptrdiff_t st) { rec_qpel(avg, si, mc, d, s, st); }
#define QP16(op, avg, sz, si) QP1(op, avg, sz, si, 0) QP1(op, avg, sz, si, 1) QP1(op, avg, sz, si, 2) QP1(op, avg, sz, si, 3) \
    QP1(op, avg, sz, si, 4) QP1(op, avg, sz, si, 5) QP1(op, avg, sz, si, 6) QP1(op, avg, sz, si, 7) QP1(op, avg, sz, si, 8) QP1(op, avg, sz, si, 9) \
    QP1(op, avg, sz, si, 10) QP1(op, avg, sz, si, 11) QP1(op, avg, sz, si, 12) QP1(op, avg, sz, si, 13) QP1(op, avg, sz, si, 14) QP1(op, avg, sz, si, 15)
QP16(put, 0, 16, 0) QP16(put, 0, 8, 1) QP16(put, 0, 4, 2) QP16(avg, 1, 16, 0) QP16(avg, 1, 8, 1) QP16(avg, 1, 4, 2)
#define QT(op, sz) { op##_qpel##sz##_mc0, op##_qpel##sz##_mc1, op##_qpel##sz##_mc2, op##_qpel##sz##_mc3, op##_qpel##sz##_mc4, op##_qpel##sz##_mc5, \
    op##_qpel##sz##_mc6, op##_qpel##sz##_mc7, op##_qpel##sz##_mc8, op##_qpel##sz##_mc9, op##_qpel##sz##_mc10, op##_qpel##sz##_mc11, \
    op##_qpel##sz##_mc12, op##_qpel##sz##_mc13, op##_qpel##sz##_mc14, op##_qpel##sz##_mc15 }
static const qpel_mc_func rec_put_qpel[3][16] = { QT(put, 16), QT(put, 8), QT(put, 4) };
static const qpel_mc_func rec_avg_qpel[3][16] = { QT(avg, 16), QT(avg, 8), QT(avg, 4) };

#define CH(op, avg, w, wi) static void op##_chroma##w(uint8_t *d, const uint8_t *s, ptrdiff_t st, int h, int x, int y) { rec_chroma(avg, wi, d, s, st, h, x, y); }
CH(put, 0, 8, 0) CH(put, 0, 4, 1) CH(put, 0, 2, 2) CH(avg, 1, 8, 0) CH(avg, 1, 4, 1) CH(avg, 1, 2, 2)

#define WT(w, wi) \
    static void weight##w(uint8_t *b, ptrdiff_t st, int h, int ld, int wt, int of) { rec_weight(wi, 0, b, b, st, h, ld, wt, 0, of); } \
    static void biweight##w(uint8_t *d, uint8_t *s, ptrdiff_t st, int h, int ld, int wd, int ws, int of) { rec_weight(wi, 1, d, s, st, h, ld, wd, ws, of); }
WT(16, 0) WT(8, 1) WT(4, 2) WT(2, 3)

/* idct_add16 / idct8_add4 / idct_add8: libffhip expands them (h264idct_template.c:168-228) and consumes sl->mb as they do */
static void rec_idct_mb(int which, uint8_t *const dst[2], const int *block_offset, int16_t *block, ptrdiff_t stride, const uint8_t nnzc[15 * 8])
{
    REC;
    int32_t off[2] = { 0, 0 };
    int pl = classify_dst(r, dst[0]), rc;
    if (pl < 0 || pl > 2 || stride != r->linesize[pl])
        FAIL(FFHIP_EINVAL);
    off[0] = (int32_t)(dst[0] - r->cur[pl]);
    if (which == 3) {
        if (pl != 1 || classify_dst(r, dst[1]) != 2)
            FAIL(FFHIP_EINVAL);
        off[1] = (int32_t)(dst[1] - r->cur[2]);
    }
    rc = ffhip_h264_picture_idct_mb(r->pic, which, pl, off, block_offset, block, nnzc);
    if (rc < 0)
        FAIL(rc);
}
static void rec_idct_add16(uint8_t *dst, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t nnzc[5 * 8])
{
    uint8_t *d[2] = { dst, NULL };
    rec_idct_mb(0, d, bo, block, stride, nnzc);
}
static void rec_idct8_add4(uint8_t *dst, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t nnzc[5 * 8])
{
    uint8_t *d[2] = { dst, NULL };
    rec_idct_mb(1, d, bo, block, stride, nnzc);
}
static void rec_idct_add8(uint8_t **dest, const int *bo, int16_t *block, ptrdiff_t stride, const uint8_t nnzc[15 * 8])
{
    rec_idct_mb(3, dest, bo, block, stride, nnzc);
}

/* the lossless bypass of an inter macroblock (hl_decode_mb_idct_luma() / hl_decode_mb() with transform_bypass: h264_mb.c:762-772,
 * h264_mb_template.c:208-221): add_pixels4_clear / add_pixels8_clear per coded block — recorded as they come */
static void rec_add_pixels(int kind, uint8_t *dst, int16_t *block, ptrdiff_t stride)
{
    REC;
    int pl = classify_dst(r, dst), rc;
    if (pl < 0 || pl > 2 || stride != r->linesize[pl])
        FAIL(FFHIP_EINVAL);
    rc = ffhip_h264_picture_idct_add(r->pic, pl, kind, (int32_t)(dst - r->cur[pl]), block);
    if (rc < 0)
        FAIL(rc);
}
static void rec_add_pixels4_clear(uint8_t *dst, int16_t *block, ptrdiff_t stride) { rec_add_pixels(FFHIP_H264_ADD_PIXELS4_CLEAR, dst, block, stride); }
static void rec_add_pixels8_clear(uint8_t *dst, int16_t *block, ptrdiff_t stride) { rec_add_pixels(FFHIP_H264_ADD_PIXELS8_CLEAR, dst, block, stride); }

static void rec_emulated_edge_mc(uint8_t *buf, const uint8_t *src, ptrdiff_t buf_linesize, ptrdiff_t src_linesize, int block_w, int block_h,
                                 int src_x, int src_y, int w, int h)
{
    REC;
    (void)block_w; (void)block_h; (void)w; (void)h; /* the window and the picture's size: the kernels clamp per sample instead */
    if (buf != r->emu_buf || buf_linesize != src_linesize)
        FAIL(FFHIP_EINVAL);
    r->emu.src = src;
    r->emu.linesize = src_linesize;
    r->emu.src_x = src_x;
    r->emu.src_y = src_y;
    r->emu.valid = 1;
}
static void rec_prefetch(const uint8_t *buf, ptrdiff_t stride, int h) { (void)buf; (void)stride; (void)h; }

/* ---- ff_h264_filter_mb(): the loop-filter members collect the macroblock's edges ------------------------------------------------ */
static _Thread_local struct {
    FFHipH264Edge e[3][8];
    int any[3], mb_x, mb_y;
} cur_edges;

static void rec_edge(int kind, uint8_t *pix, ptrdiff_t stride, int alpha, int beta, const int8_t *tc0, int mbaff_member)
{
    REC;
    const int chroma = (kind & 2) != 0, dir = !(kind & 1); /* FFHIP_H264_LF_V_* filter across a horizontal edge: dir 1 */
    int pl = classify_dst(r, pix), x, y, e;
    ptrdiff_t o;
    FFHipH264Edge *E;
    if (r->mbaff) {
        /* an MBAFF frame: the call as it is — where, which member, at the frame's line size or twice it (a field macroblock's lines, or a
         * frame macroblock's edge towards a field pair: h264_loopfilter.c:520-545,812-830) */
        FFHipH264Edge c = { 0 };
        int rc;
        if (pl < 0 || pl > 2 || (pl != 0) != chroma || (stride != r->linesize[pl] && stride != 2 * r->linesize[pl]))
            FAIL(FFHIP_EINVAL);
        c.offset = (int32_t)(pix - r->cur[pl]);
        c.kind = (uint8_t)kind;
        c.alpha = (uint8_t)alpha;
        c.beta = (uint8_t)beta;
        c.pad = (uint8_t)((stride != r->linesize[pl] ? FFHIP_H264_LF_CALL_FIELD : 0) | (mbaff_member ? FFHIP_H264_LF_CALL_MBAFF : 0));
        if (tc0)
            memcpy(c.tc0, tc0, 4);
        rc = ffhip_h264_mbaff_filter_call(r->chains, pl, cur_edges.mb_x, cur_edges.mb_y, &c);
        if (rc < 0)
            FAIL(rc);
        return;
    }
    if (mbaff_member)
        FAIL(FFHIP_EINVAL);
    /* 4:4:4: the luma members on every plane (filter_mb_edgev / edgeh on img_cb / img_cr, h264_loopfilter.c:601-703) */
    if (pl < 0 || pl > 2 || (r->cfmt == 3 ? chroma : (pl != 0) != chroma) || stride != r->linesize[pl])
        FAIL(FFHIP_EINVAL);
    o = pix - r->cur[pl];
    y = (int)(o / stride);
    x = (int)(o % stride) >> r->pixel_shift;
    if (!chroma) {
        if ((x >> 4) != cur_edges.mb_x || (y >> 4) != cur_edges.mb_y)
            FAIL(FFHIP_EINVAL);
        e = dir ? (y & 15) >> 2 : (x & 15) >> 2;
    } else if (r->cfmt == 2) {
        /* 4:2:2: an 8 x 16 chroma macroblock — the vertical edges x = 0, 4 (h_loop_filter_chroma422, 16 lines), then the horizontal ones
         * y = 0, 4, 8, 12 (filter_mb_dir(), h264_loopfilter.c:601-703 with chroma422): six records, [0..1] then [2..5] */
        if ((x >> 3) != cur_edges.mb_x || (y >> 4) != cur_edges.mb_y)
            FAIL(FFHIP_EINVAL);
        E = &cur_edges.e[pl][dir ? 2 + ((y & 15) >> 2) : (x & 7) >> 2];
        goto fill;
    } else {
        if ((x >> 3) != cur_edges.mb_x || (y >> 3) != cur_edges.mb_y)
            FAIL(FFHIP_EINVAL);
        e = dir ? (y & 7) >> 2 : (x & 7) >> 2;
    }
    E = &cur_edges.e[pl][dir * (chroma ? 2 : 4) + e];
fill:
    E->offset = (int32_t)o;
    E->kind = (uint8_t)kind;
    E->alpha = (uint8_t)alpha;
    E->beta = (uint8_t)beta;
    if (tc0)
        memcpy(E->tc0, tc0, 4);
    cur_edges.any[pl] = 1;
}
#define LF(name, kind) static void rec_##name(uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0) { rec_edge(kind, pix, stride, alpha, beta, tc0, 0); }
#define LFI(name, kind) static void rec_##name(uint8_t *pix, ptrdiff_t stride, int alpha, int beta) { rec_edge(kind, pix, stride, alpha, beta, NULL, 0); }
#define LFM(name, kind) static void rec_##name(uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0) { rec_edge(kind, pix, stride, alpha, beta, tc0, 1); }
#define LFMI(name, kind) static void rec_##name(uint8_t *pix, ptrdiff_t stride, int alpha, int beta) { rec_edge(kind, pix, stride, alpha, beta, NULL, 1); }
LF(v_loop_filter_luma, FFHIP_H264_LF_V_LUMA) LF(h_loop_filter_luma, FFHIP_H264_LF_H_LUMA)
LF(v_loop_filter_chroma, FFHIP_H264_LF_V_CHROMA) LF(h_loop_filter_chroma, FFHIP_H264_LF_H_CHROMA)
LFI(v_loop_filter_luma_intra, FFHIP_H264_LF_V_LUMA_INTRA) LFI(h_loop_filter_luma_intra, FFHIP_H264_LF_H_LUMA_INTRA)
LFI(v_loop_filter_chroma_intra, FFHIP_H264_LF_V_CHROMA_INTRA) LFI(h_loop_filter_chroma_intra, FFHIP_H264_LF_H_CHROMA_INTRA)
/* the left edge of a macroblock whose left pair is of the other kind: half as many lines per call (h264dsp_template.c:127-133,262-272) */
LFM(h_loop_filter_luma_mbaff, FFHIP_H264_LF_H_LUMA) LFMI(h_loop_filter_luma_mbaff_intra, FFHIP_H264_LF_H_LUMA_INTRA)
LFM(h_loop_filter_chroma_mbaff, FFHIP_H264_LF_H_CHROMA) LFMI(h_loop_filter_chroma_mbaff_intra, FFHIP_H264_LF_H_CHROMA_INTRA)

av_cold void ff_h264_hip_recorder_install(H264Context *h)
{
    for (int s = 0; s < 3; s++)
        for (int i = 0; i < 16; i++) {
            h->h264qpel.put_h264_qpel_pixels_tab[s][i] = rec_put_qpel[s][i];
            h->h264qpel.avg_h264_qpel_pixels_tab[s][i] = rec_avg_qpel[s][i];
        }
    h->h264chroma.put_h264_chroma_pixels_tab[0] = put_chroma8;
    h->h264chroma.put_h264_chroma_pixels_tab[1] = put_chroma4;
    h->h264chroma.put_h264_chroma_pixels_tab[2] = put_chroma2;
    h->h264chroma.avg_h264_chroma_pixels_tab[0] = avg_chroma8;
    h->h264chroma.avg_h264_chroma_pixels_tab[1] = avg_chroma4;
    h->h264chroma.avg_h264_chroma_pixels_tab[2] = avg_chroma2;
    h->h264dsp.weight_pixels_tab[0] = weight16;     h->h264dsp.biweight_pixels_tab[0] = biweight16;
    h->h264dsp.weight_pixels_tab[1] = weight8;      h->h264dsp.biweight_pixels_tab[1] = biweight8;
    h->h264dsp.weight_pixels_tab[2] = weight4;      h->h264dsp.biweight_pixels_tab[2] = biweight4;
    h->h264dsp.weight_pixels_tab[3] = weight2;      h->h264dsp.biweight_pixels_tab[3] = biweight2;
    h->h264dsp.idct_add16  = rec_idct_add16;
    h->h264dsp.idct8_add4  = rec_idct8_add4;
    h->h264dsp.idct_add8   = rec_idct_add8;
    h->h264dsp.add_pixels4_clear = rec_add_pixels4_clear;
    h->h264dsp.add_pixels8_clear = rec_add_pixels8_clear;
    h->h264dsp.v_loop_filter_luma         = rec_v_loop_filter_luma;
    h->h264dsp.h_loop_filter_luma         = rec_h_loop_filter_luma;
    h->h264dsp.v_loop_filter_luma_intra   = rec_v_loop_filter_luma_intra;
    h->h264dsp.h_loop_filter_luma_intra   = rec_h_loop_filter_luma_intra;
    h->h264dsp.v_loop_filter_chroma       = rec_v_loop_filter_chroma;
    h->h264dsp.h_loop_filter_chroma       = rec_h_loop_filter_chroma;
    h->h264dsp.v_loop_filter_chroma_intra = rec_v_loop_filter_chroma_intra;
    h->h264dsp.h_loop_filter_chroma_intra = rec_h_loop_filter_chroma_intra;
    h->h264dsp.h_loop_filter_luma_mbaff         = rec_h_loop_filter_luma_mbaff;
    h->h264dsp.h_loop_filter_luma_mbaff_intra   = rec_h_loop_filter_luma_mbaff_intra;
    h->h264dsp.h_loop_filter_chroma_mbaff       = rec_h_loop_filter_chroma_mbaff;
    h->h264dsp.h_loop_filter_chroma_mbaff_intra = rec_h_loop_filter_chroma_mbaff_intra;
    h->vdsp.emulated_edge_mc = rec_emulated_edge_mc;
    h->vdsp.prefetch = rec_prefetch;
}

/* What keeps a whole picture on the C path, known before its first macroblock (MBAFF frames that are not 4:2:0; lossless streams —
 * a qpprime_y_zero_transform_bypass macroblock may turn up at any macroblock — above 8 bits): a caller asks this BEFORE it
 * begins to record — once macroblocks have been recorded their coefficients are consumed and the pixels exist only as records, there is no
 * way back. */
int ff_h264_hip_picture_supported(const H264Context *h)
{
    if (h->ps.sps->transform_bypass && h->ps.sps->bit_depth_luma != 8)
        return 0;
    /* (streams of x264 before build 151 predict Intra8x8 DPCM blocks from the UNFILTERED edge: h264_mb.c:641-643; libffhip has the filtered form) */
    if (h->ps.sps->transform_bypass && h->ps.sps->profile_idc == 244 && (unsigned)h->x264_build < 151U)
        return 0;
    return !FRAME_MBAFF(h) || (h->ps.sps->chroma_format_idc == 1 && !(h->mb_height & 1));
}

void ff_h264_hip_recorder_begin(FFHipH264Recorder *r, FFHipH264Picture *pic, const H264Context *h, const H264SliceContext *sl,
                                const uint8_t *const ref_base[3])
{
    memset(r, 0, sizeof(*r));
    r->pic = pic;
    r->pixel_shift = h->pixel_shift;
    r->cfmt = h->ps.sps->chroma_format_idc ? h->ps.sps->chroma_format_idc : 1; /* monochrome is decoded as 4:2:0 with mid-grey chroma */
    r->field = FIELD_PICTURE(h) && !FRAME_MBAFF(h);
    for (int pl = 0; pl < 3; pl++) {
        const ptrdiff_t ls = pl ? sl->uvlinesize : sl->linesize;
        /* a field: the lines of its parity, from the frame's first (top) or second (bottom) line on */
        r->cur[pl] = h->cur_pic.f->data[pl] + (r->field && h->picture_structure == PICT_BOTTOM_FIELD ? ls : 0);
        r->ref_base[pl] = ref_base[pl];
        r->linesize[pl] = ls << r->field;
        r->pic_w[pl] = h->mb_width * (pl && r->cfmt != 3 ? 8 : 16);
        r->rows[pl] = (h->mb_height >> r->field) * (pl && r->cfmt == 1 ? 8 : 16);   /* (4:2:2: chroma 8 wide, 16 rows per macroblock) */
    }
    for (int pl = 0; pl < 3 && r->error >= 0; pl++) { /* the current picture itself must be within reach of the base (see fits32()) */
        int32_t o;
        if (!fits32(r->cur[pl] - r->ref_base[pl], &o) || !fits32(r->cur[pl] + plane_span(r, pl) - r->ref_base[pl], &o))
            r->error = FFHIP_EINVAL;
    }
    r->scratch = sl->bipred_scratchpad;
    /* tmp_y starts 16 chroma rows in and is 16 luma rows tall; both buffers are walked at mb_linesize (alloc_scratch_buffers(),
     * h264_slice.c:168-200, sizes them for field macroblocks) */
    r->scratch_size = (size_t)16 * r->linesize[1] + (size_t)16 * r->linesize[0];
    r->emu_buf = sl->edge_emu_buffer;
    r->emu_size = (size_t)21 * r->linesize[0] + ((size_t)21 << h->pixel_shift);
}

void ff_h264_hip_recorder_begin_mbaff(FFHipH264Recorder *r, FFHipH264Picture *frame_mbs, FFHipH264Picture *top_mbs, FFHipH264Picture *bottom_mbs,
                                      FFHipH264Mbaff *chains, const H264Context *h, const H264SliceContext *sl, const uint8_t *const ref_base[3])
{
    FFHipH264Picture *const pics[3] = { frame_mbs, top_mbs, bottom_mbs };
    memset(r, 0, sizeof(*r));
    r->mbaff = 1;
    r->chains = chains;
    r->pixel_shift = h->pixel_shift;
    r->cfmt = 1;
    if (!ff_h264_hip_picture_supported(h) || !FRAME_MBAFF(h))
        r->error = FFHIP_ENOSYS;
    for (int v = 0; v < 3; v++) {
        r->view[v].pic = pics[v];
        for (int pl = 0; pl < 3; pl++) {
            const ptrdiff_t ls = pl ? sl->uvlinesize : sl->linesize;
            r->view[v].cur[pl] = h->cur_pic.f->data[pl] + (v == 2 ? ls : 0);
            r->view[v].linesize[pl] = ls << (v > 0);
            r->view[v].rows[pl] = (h->mb_height >> (v > 0)) * (pl ? 8 : 16);
        }
    }
    for (int pl = 0; pl < 3; pl++) {
        r->ref_base[pl] = ref_base[pl];
        r->pic_w[pl] = h->mb_width * (pl ? 8 : 16);
    }
    r->scratch = sl->bipred_scratchpad;
    r->emu_buf = sl->edge_emu_buffer;
    r->active = -1;
    select_view(r, 0);
    for (int pl = 0; pl < 3 && r->error >= 0; pl++) { /* the frame's planes within reach of the base: so are its fields' */
        int32_t o;
        if (!fits32(r->cur[pl] - r->ref_base[pl], &o) || !fits32(r->cur[pl] + plane_span(r, pl) - r->ref_base[pl], &o))
            r->error = FFHIP_EINVAL;
    }
}

int ff_h264_hip_hl_decode_mb(FFHipH264Recorder *r, const H264Context *h, H264SliceContext *sl)
{
    const int mb_type = h->cur_pic.mb_type[sl->mb_xy];
    if (r->error < 0)
        return r->error;
    if (r->mbaff) {
        if (!FRAME_MBAFF(h))
            return r->error = FFHIP_EINVAL;
        select_view(r, MB_FIELD(sl) ? 1 + (sl->mb_y & 1) : 0);
    } else if (FRAME_MBAFF(h) || !!MB_FIELD(sl) != r->field) {
        return r->error = FFHIP_ENOSYS;   /* such a picture stays on the C path as a whole */
    }
    if (sl->qscale == 0 && h->ps.sps->transform_bypass && r->pixel_shift)
        return r->error = FFHIP_ENOSYS;   /* (ff_h264_hip_picture_supported() said so) */
    if (IS_INTRA(mb_type)) {
        FFHipH264IntraMB m = { 0 };
        /* hl_decode_mb()'s transform_bypass (h264_mb_template.c:51): the residual is added as samples; profile_idc 244: vertically /
         * horizontally predicted blocks through the pred*_add forms (libffhip's packer does what that takes) */
        if (sl->qscale == 0 && h->ps.sps->transform_bypass)
            m.flags = FFHIP_H264_INTRA_BYPASS | (h->ps.sps->profile_idc == 244 ? FFHIP_H264_INTRA_DPCM : 0);
        const int intra_qmul = 0;
        m.mb_x = (int16_t)sl->mb_x;
        /* a field picture: rows of the field (mb_y = 2 * row + bottom, h264_slice.c:2676-2680); an MBAFF frame: the row in the frame */
        m.mb_y = (int16_t)(r->mbaff ? sl->mb_y : sl->mb_y >> r->field);
        m.type = IS_INTRA_PCM(mb_type) ? FFHIP_H264_INTRA_PCM : IS_INTRA16x16(mb_type) ? FFHIP_H264_INTRA_16x16
               : IS_8x8DCT(mb_type) ? FFHIP_H264_INTRA_8x8 : FFHIP_H264_INTRA_4x4;
        m.pred16 = (uint8_t)sl->intra16x16_pred_mode;
        m.chroma_pred = (uint8_t)sl->chroma_pred_mode;
        m.cbp = (uint8_t)(sl->cbp & 0x3f);
        m.topleft_avail = (uint16_t)sl->topleft_samples_available;
        m.topright_avail = (uint16_t)sl->topright_samples_available;
        for (int i = 0; i < 16; i++)
            m.pred4[i] = (uint8_t)sl->intra4x4_pred_mode_cache[scan8[i]];
        /* luma_dc_dequant_idct's and chroma_dc_dequant_idct's qmul (h264_mb.c:707-711, h264_mb_template.c:246-253); 4:4:4: the same
         * three table entries are plane p's luma_dc_dequant_idct qmul, dequant4_coeff[p][p ? chroma_qp[p - 1] : qscale][0] (h264_mb.c:626,
         * 712), and sl->mb / sl->mb_luma_dc / the cache hold the three planes one after the other (libffhip splits them) */
        m.qmul[0] = h->ps.pps->dequant4_coeff[intra_qmul][sl->qscale][0];
        /* (4:2:2: chroma422_dc_dequant_idct's quantiser sits three steps up, h264_mb_template.c:232-236) */
        m.qmul[1] = h->ps.pps->dequant4_coeff[1][sl->chroma_qp[0] + (CHROMA422(h) ? 3 : 0)][0];
        m.qmul[2] = h->ps.pps->dequant4_coeff[2][sl->chroma_qp[1] + (CHROMA422(h) ? 3 : 0)][0];
        h->list_counts[sl->mb_xy] = sl->list_count;   /* hl_decode_mb()'s one side effect outside the pixels (h264_mb_template.c:61) */
        if (IS_INTRA_PCM(mb_type) && !h->ps.sps->chroma_format_idc) {
            /* monochrome I_PCM: 256 luma fields in the bitstream, the chroma planes are set to mid-grey (h264_mb_template.c:112-119,
             * 137-142): the record's 384 fields = the luma fields + 128 fields of 1 << (bit_depth - 1), MSB-first as the bitstream has them */
            uint8_t pcm[384 * 2];
            const int bd = h->ps.sps->bit_depth_luma, lbytes = 32 * bd;
            memcpy(pcm, sl->intra_pcm_ptr, lbytes);
            if (bd == 8) {
                memset(pcm + 256, 128, 128);
            } else {
                uint32_t acc = 0;
                int have = 0, at = lbytes;
                for (int k = 0; k < 128; k++) {
                    acc = (acc << bd) | (1u << (bd - 1));
                    have += bd;
                    while (have >= 8) {
                        have -= 8;
                        pcm[at++] = (uint8_t)(acc >> have);
                    }
                }
            }
            r->error = FFMIN(0, ffhip_h264_picture_intra_mb(r->pic, &m, sl->non_zero_count_cache, sl->mb, sl->mb_luma_dc[0], pcm));
            return r->error;
        }
        if (r->mbaff)
            r->error = FFMIN(0, ffhip_h264_mbaff_intra_mb(r->chains, &m, !!MB_FIELD(sl), sl->non_zero_count_cache, sl->mb, sl->mb_luma_dc[0],
                                                          sl->intra_pcm_ptr));
        else
            r->error = FFMIN(0, ffhip_h264_picture_intra_mb(r->pic, &m, sl->non_zero_count_cache, sl->mb, sl->mb_luma_dc[0], sl->intra_pcm_ptr));
        return r->error;
    }
    cur_rec = r;
    r->sl = sl;
    r->emu.valid = 0;
    r->npend = 0;
    ff_h264_hl_decode_mb(h, sl);
    cur_rec = NULL;
    if (r->error >= 0 && r->npend)
        r->error = FFHIP_EINVAL;   /* a prediction went to the scratchpad and no biweight claimed it */
    return r->error;
}

int ff_h264_hip_filter_mb(FFHipH264Recorder *r, const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y)
{
    if (r->error < 0)
        return r->error;
    if (r->mbaff) {
        /* loop_filter()'s addressing of the macroblock (h264_slice.c:2470-2491), on the frame's planes */
        const int field = !!MB_FIELD(sl);
        uint8_t *img[3];
        if (!FRAME_MBAFF(h))
            return r->error = FFHIP_EINVAL;
        select_view(r, 0);
        for (int pl = 0; pl < 3; pl++) {
            const ptrdiff_t ls = r->linesize[pl];
            const int bh = pl ? 8 : 16;
            img[pl] = (uint8_t *)r->cur[pl] + (((ptrdiff_t)mb_x * bh) << r->pixel_shift) + (ptrdiff_t)mb_y * bh * ls - (field && (mb_y & 1) ? (bh - 1) * ls : 0);
        }
        memset(&cur_edges, 0, sizeof(cur_edges));
        cur_edges.mb_x = mb_x;
        cur_edges.mb_y = mb_y;
        sl->mb_linesize = r->linesize[0] << field;
        sl->mb_uvlinesize = r->linesize[1] << field;
        cur_rec = r;
        ff_h264_filter_mb(h, sl, mb_x, mb_y, img[0], img[1], img[2], (unsigned)sl->mb_linesize, (unsigned)sl->mb_uvlinesize);
        cur_rec = NULL;
        return r->error;
    }
    if (FRAME_MBAFF(h) || !!MB_FIELD(sl) != r->field)
        return r->error = FFHIP_ENOSYS;
    memset(&cur_edges, 0, sizeof(cur_edges));
    cur_edges.mb_x = mb_x;
    cur_edges.mb_y = mb_y >> r->field;     /* the row inside the field: what loop_filter()'s `dest -= linesize * 15` amounts to */
    cur_rec = r;
    {
        /* the chroma planes' macroblock width and height; the macroblock's row in `pic` */
        const int cs = r->cfmt == 3 ? 16 : 8, ch = r->cfmt == 1 ? 8 : 16, fy = mb_y >> r->field;
        uint8_t *y  = (uint8_t *)r->cur[0] + ((ptrdiff_t)mb_x << h->pixel_shift) * 16 + (ptrdiff_t)fy * r->linesize[0] * 16;
        uint8_t *cb = (uint8_t *)r->cur[1] + ((ptrdiff_t)mb_x << h->pixel_shift) * cs + (ptrdiff_t)fy * r->linesize[1] * ch;
        uint8_t *cr = (uint8_t *)r->cur[2] + ((ptrdiff_t)mb_x << h->pixel_shift) * cs + (ptrdiff_t)fy * r->linesize[2] * ch;
        sl->mb_linesize = r->linesize[0];      /* as loop_filter() leaves them (h264_slice.c:2480-2491) */
        sl->mb_uvlinesize = r->linesize[1];
        ff_h264_filter_mb(h, sl, mb_x, mb_y, y, cb, cr, (unsigned)r->linesize[0], (unsigned)r->linesize[1]);
    }
    cur_rec = NULL;
    for (int pl = 0; pl < 3 && r->error >= 0; pl++)
        if (cur_edges.any[pl])
            r->error = FFMIN(0, ffhip_h264_picture_deblock_mb(r->pic, pl, mb_x, mb_y >> r->field, cur_edges.e[pl]));
    return r->error;
}
