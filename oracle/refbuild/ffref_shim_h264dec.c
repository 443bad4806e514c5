/*
 * ffref_shim_h264dec.c — ours, TEST INFRASTRUCTURE ONLY.  The reference's WHOLE H.264 decoder (h264dec.c, h264_slice.c, h264_cavlc.c,
 * h264_cabac.c, h264_ps.c, h264_refs.c, h264_picture.c, h264_direct.c, h264_mvpred.h, h2645_parse.c ... compiled where they lie by
 * oracle/refbuild/Makefile, target `h264dec`) driven through libavcodec's public API — avcodec_open2() / avcodec_send_packet() /
 * avcodec_receive_frame() — on access units a test-side bitstream writer made (tests/h264_bitstream.py), in two modes:
 *
 *   plain   the decoder as it is: the C dsp tables reconstruct and filter every macroblock in place;
 *   record  the FFmpeg-side patch of integration/avcodec_h264_picture_hip.c at its two call sites in h264_slice.c:
 *             decode_slice():  ff_h264_hl_decode_mb(h, sl)            (h264_slice.c:2632, 2648, 2703, 2713)  -> ff_h264_hip_hl_decode_mb()
 *             loop_filter():   ff_h264_filter_mb_fast() / ff_h264_filter_mb()  (h264_slice.c:2499-2505)       -> ff_h264_hip_filter_mb()
 *           h264_slice.c is compiled unchanged with the three callee names re-pointed at the hooks below (-D...: the Makefile), which
 *           is what `if (h->hip_recorder) ... else ...` at those lines amounts to.  Everything before the calls — slice headers,
 *           CAVLC, h264_mvpred.h, fill_decode_caches(), fill_filter_caches(), reference lists, the picture buffer — is the decoder's
 *           own, untouched; so is ff_h264_hl_decode_mb() / ff_h264_filter_mb() themselves, which the recorder runs over its recording
 *           dsp members.  A picture (a frame, or one field) becomes one libffhip picture object; when the decoder moves on to the
 *           next picture (or is drained) the finished one is handed to the `flush` callback of the test, which executes its lists on
 *           the picture buffer — oracle/emul_h264_picture.cpp on the host arena (CPU tier), ffhip_h264_picture_flush() on a device
 *           mirror of the arena (GPU tier) — before any later picture's motion compensation can refer to it.
 *
 * Frames come from ONE arena (get_buffer2 below): the recorder turns addresses into offsets from a single base, which is what a
 * decoded-picture buffer living in one hip allocation gives the real patch.
 */
#include <stdlib.h>
#include <string.h>

#include "libavcodec/avcodec.h"
#include "libavcodec/codec_internal.h"
#include "libavcodec/h264dec.h"
#include "libavcodec/h264_ps.h"
#include "libavcodec/h264chroma.h"
#include "libavcodec/h264dsp.h"
#include "libavcodec/h264qpel.h"
#include "libavcodec/videodsp.h"
#include "libavutil/buffer.h"
#include "libavutil/frame.h"
#include "libavutil/imgutils.h"
#include "libavutil/mem.h"
#include "libavutil/pixdesc.h"

#include "ffhip.h"
#include "avcodec_h264_picture_hip.h"

#define MAX_OUT 64

/* dst_off[pl]: the picture's planes as byte offsets into the arena (a bottom field: one line down); stride[pl]: the picture's line sizes
 * (a field: twice the frame's).  Returns 0 or < 0. */
typedef int (*ffref_h264_flush_fn)(void *opaque, void *pic, const int64_t dst_off[3], const int stride[3], int mb_w, int mb_h, int field);
/* An MBAFF frame (round 6): `flush` above is called three times first — the frame macroblocks' object (field 2), the top- and the
 * bottom-field macroblocks' objects (field 3: half the rows, twice the line sizes, the bottom one's planes a line down) — then this with the
 * FFHipH264Mbaff object, the FRAME's planes and line sizes. */
typedef int (*ffref_h264_flush_mbaff_fn)(void *opaque, void *chains, const int64_t dst_off[3], const int stride[3], int mb_w, int mb_h);

typedef struct FFRefH264Stream {
    AVCodecContext *avctx;
    AVPacket *pkt;
    int record;
    uint8_t *arena;
    size_t arena_size, arena_used;
    ffref_h264_flush_fn flush;
    void *flush_opaque;
    ffref_h264_flush_mbaff_fn flush_mbaff;
    void *flush_mbaff_opaque;
    /* an MBAFF frame being recorded: pic = the frame macroblocks' object, these the field macroblocks' objects and the chains */
    FFHipH264Picture *fpic[2];
    FFHipH264Mbaff *chains;
    long mbaff_pictures;
    long mbs_bypass;               /* recorded macroblocks decoded with the transform bypassed (qscale 0, sps->transform_bypass) */
    /* the picture being recorded */
    FFHipH264Recorder rec;
    FFHipH264Picture *pic;
    const H264Picture *cur_ptr;
    int cur_structure, cur_field, cur_mb_w, cur_mb_h;
    int cur_plain;                 /* the current picture stays on the C path as a whole (ff_h264_hip_picture_supported() said no) */
    int recording_tables;          /* h's dsp tables hold the recording members */
    long plain_pictures;
    int64_t cur_off[3];
    int cur_stride[3];
    /* output */
    AVFrame *out[MAX_OUT];
    int nout;
    /* counters */
    long pictures, mbs_hl, mbs_filter, refused, errors, decode_errors;
    long mbs_class[8];             /* recorded macroblocks by what the DECODER derived for them: see ffref_h264stream_stat() */
    int first_error;
    int64_t base_shift;            /* test knob: the base the recorder counts offsets from lies this many bytes BELOW the arena */
} FFRefH264Stream;

extern const FFCodec ff_h264_decoder;

static void arena_noop_free(void *opaque, uint8_t *data) { (void)opaque; (void)data; }

/* AVCodecContext.get_buffer2: planes cut from the arena, every frame of a stream with the same line sizes, no border (the decoder
 * emulates edges: h264_mb.c:229-260) */
static int arena_get_buffer(AVCodecContext *avctx, AVFrame *f, int flags)
{
    FFRefH264Stream *s = avctx->opaque;
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(f->format);
    const int ps = d->comp[0].depth > 8;
    (void)flags;
    for (int pl = 0; pl < 3; pl++) {
        const int w = pl ? AV_CEIL_RSHIFT(FFALIGN(f->width, 16), d->log2_chroma_w) : FFALIGN(f->width, 16);
        const int h = pl ? AV_CEIL_RSHIFT(FFALIGN(f->height, 32), d->log2_chroma_h) : FFALIGN(f->height, 32);
        const int ls = FFALIGN((w << ps) + 32, 64);
        const size_t sz = (size_t)ls * h + 256;
        if (s->arena_used + sz > s->arena_size)
            return AVERROR(ENOMEM);
        f->data[pl] = s->arena + s->arena_used;
        f->linesize[pl] = ls;
        f->buf[pl] = av_buffer_create(f->data[pl], sz, arena_noop_free, NULL, 0);
        if (!f->buf[pl])
            return AVERROR(ENOMEM);
        s->arena_used += FFALIGN(sz, 256);
    }
    f->extended_data = f->data;
    return 0;
}

static void note(FFRefH264Stream *s, int r);

static int flush_current(FFRefH264Stream *s)
{
    int r = 0;
    if (!s->pic) {
        s->cur_plain = 0;
        return 0;
    }
    if (s->rec.error < 0) {
        r = s->rec.error;
    } else if (s->chains) {
        /* an MBAFF frame: the three inter objects, then the chains (without a test callback for them the frame cannot be finished) */
        if (s->flush && s->flush_mbaff) {
            int64_t off[3];
            int st[3];
            r = s->flush(s->flush_opaque, s->pic, s->cur_off, s->cur_stride, s->cur_mb_w, s->cur_mb_h, 2);
            for (int v = 0; v < 2 && r >= 0; v++) {
                for (int pl = 0; pl < 3; pl++) {
                    off[pl] = s->cur_off[pl] + (v ? s->cur_stride[pl] : 0);
                    st[pl] = 2 * s->cur_stride[pl];
                }
                r = s->flush(s->flush_opaque, s->fpic[v], off, st, s->cur_mb_w, s->cur_mb_h / 2, 3);
            }
            if (r >= 0)
                r = s->flush_mbaff(s->flush_mbaff_opaque, s->chains, s->cur_off, s->cur_stride, s->cur_mb_w, s->cur_mb_h);
        } else if (s->flush) {
            r = FFHIP_ENOSYS;
        }
    } else if (s->flush) {
        r = s->flush(s->flush_opaque, s->pic, s->cur_off, s->cur_stride, s->cur_mb_w, s->cur_mb_h, s->cur_field);
    }
    if (r < 0) {
        s->errors++;
        if (!s->first_error)
            s->first_error = r;
    }
    ffhip_h264_picture_free(&s->pic);
    ffhip_h264_picture_free(&s->fpic[0]);
    ffhip_h264_picture_free(&s->fpic[1]);
    ffhip_h264_mbaff_free(&s->chains);
    s->pic = NULL;
    s->cur_ptr = NULL;
    return r;
}

/* the decoder has started on a picture the recorder has not seen: the finished one is executed, a picture object is made for the new one */
static int begin_picture(FFRefH264Stream *s, const H264Context *h, H264SliceContext *sl)
{
    const SPS *sps = h->ps.sps;
    const int field = FIELD_PICTURE(h) && !FRAME_MBAFF(h);
    const uint8_t *base[3] = { s->arena - s->base_shift, s->arena - s->base_shift, s->arena - s->base_shift };
    int r;
    flush_current(s);
    s->cur_plain = 0;
    if (!ff_h264_hip_picture_supported(h)) {
        /* an MBAFF frame (or a lossless stream): decided BEFORE a macroblock of the picture is recorded; the whole picture runs through
         * the reference's own functions on the C tables (made again as h264_slice.c:1022-1028 makes them), in place, on the same
         * picture buffer the recorded pictures' flushes write to */
        H264Context *hw = (H264Context *)h;
        if (s->recording_tables) {
            ff_h264dsp_init(&hw->h264dsp, sps->bit_depth_luma, sps->chroma_format_idc);
            ff_h264chroma_init(&hw->h264chroma, sps->bit_depth_chroma);
            ff_h264qpel_init(&hw->h264qpel, sps->bit_depth_luma);
            ff_videodsp_init(&hw->vdsp, sps->bit_depth_luma);
            s->recording_tables = 0;
        }
        s->cur_plain = 1;
        s->cur_ptr = h->cur_pic_ptr;
        s->cur_structure = h->picture_structure;
        s->plain_pictures++;
        return 0;
    }
    r = ffhip_h264_picture_create_fmt(&s->pic, h->mb_width, h->mb_height >> field, sps->bit_depth_luma, sps->chroma_format_idc ? sps->chroma_format_idc : 1);
    if (r >= 0 && FRAME_MBAFF(h)) {
        for (int v = 0; v < 2 && r >= 0; v++)
            r = ffhip_h264_picture_create_fmt(&s->fpic[v], h->mb_width, h->mb_height / 2, sps->bit_depth_luma, 1);
        if (r >= 0)
            r = ffhip_h264_mbaff_create_fmt(&s->chains, h->mb_width, h->mb_height, sps->bit_depth_luma);
        if (r < 0) {
            ffhip_h264_picture_free(&s->pic);
            ffhip_h264_picture_free(&s->fpic[0]);
            ffhip_h264_picture_free(&s->fpic[1]);
            ffhip_h264_mbaff_free(&s->chains);
        }
    }
    if (r < 0 || !s->pic) {
        s->errors++;
        if (!s->first_error)
            s->first_error = r < 0 ? r : FFHIP_ENOMEM;
        s->pic = NULL;
        return -1;
    }
    ffhip_h264_picture_begin(s->pic);
    /* the dsp tables may have been made anew for this picture's format (h264_slice.c init_dimensions / h264_init_ps) */
    ff_h264_hip_recorder_install((H264Context *)h);
    s->recording_tables = 1;
    if (s->chains) {
        ffhip_h264_picture_begin(s->fpic[0]);
        ffhip_h264_picture_begin(s->fpic[1]);
        ffhip_h264_mbaff_begin(s->chains);
        ff_h264_hip_recorder_begin_mbaff(&s->rec, s->pic, s->fpic[0], s->fpic[1], s->chains, h, sl, base);
        s->mbaff_pictures++;
    } else {
        ff_h264_hip_recorder_begin(&s->rec, s->pic, h, sl, base);
    }
    note(s, s->rec.error);
    s->cur_ptr = h->cur_pic_ptr;
    s->cur_structure = h->picture_structure;
    s->cur_field = field;
    s->cur_mb_w = h->mb_width;
    s->cur_mb_h = h->mb_height >> field;
    for (int pl = 0; pl < 3; pl++) {
        s->cur_off[pl] = s->rec.cur[pl] - s->arena;
        s->cur_stride[pl] = (int)s->rec.linesize[pl];
    }
    s->pictures++;
    return 0;
}

static FFRefH264Stream *session_of(const H264Context *h)
{
    FFRefH264Stream *s = h->avctx->opaque;
    return s && s->record ? s : NULL;
}

/* 1: record the macroblock; 0: nothing to do (an error was noted); -1: this picture runs on the C path */
static int ready(FFRefH264Stream *s, const H264Context *h, H264SliceContext *sl)
{
    if (s->cur_ptr != h->cur_pic_ptr || s->cur_structure != h->picture_structure || (!s->pic && !s->cur_plain))
        if (begin_picture(s, h, sl) < 0)
            return 0;
    if (s->cur_plain)
        return -1;
    /* a later slice of the picture: its own reference lists / scratch buffers are the slice context's, which begin() bound once; the
     * recorder reads them through r->sl per macroblock */
    return 1;
}

static void note(FFRefH264Stream *s, int r)
{
    if (r < 0) {
        if (r == FFHIP_ENOSYS)
            s->refused++;
        s->errors++;
        if (!s->first_error)
            s->first_error = r;
    }
}

/* ---- the three names h264_slice.c calls (see the Makefile's -D for that file) ---- */
void ffref_hook_hl_decode_mb(const H264Context *h, H264SliceContext *sl)
{
    FFRefH264Stream *s = session_of(h);
    if (!s) {
        ff_h264_hl_decode_mb(h, sl);
        return;
    }
    switch (ready(s, h, sl)) {
    case 0: return;
    case -1: ff_h264_hl_decode_mb(h, sl); return;
    }
    s->mbs_hl++;
    {
        /* what kind of macroblock the decoder's own parsing and derivation (h264_cavlc.c, h264_direct.c, h264_slice.c implicit_weight_table /
         * h264_parse.c pred_weight_table) made of it — the tests assert that the streams reach these paths */
        const int mt = h->cur_pic.mb_type[sl->mb_xy];
        if (!IS_INTRA(mt)) {
            s->mbs_class[0] += (mt & (MB_TYPE_P0L0 | MB_TYPE_P1L0)) && (mt & (MB_TYPE_P0L1 | MB_TYPE_P1L1));   /* uses both lists */
            s->mbs_class[1] += !!IS_DIRECT(mt) || (IS_8X8(mt) && (IS_DIRECT(sl->sub_mb_type[0]) || IS_DIRECT(sl->sub_mb_type[1]) ||
                                                                  IS_DIRECT(sl->sub_mb_type[2]) || IS_DIRECT(sl->sub_mb_type[3])));
            s->mbs_class[3] += sl->pwt.use_weight == 1;
            s->mbs_class[4] += sl->pwt.use_weight == 2;
            s->mbs_class[5] += sl->slice_type_nos == AV_PICTURE_TYPE_B;
        } else {
            s->mbs_class[6] += IS_8x8DCT(mt) && !IS_INTRA16x16(mt) && !IS_INTRA_PCM(mt);                         /* Intra8x8 */
        }
        s->mbs_class[2] += !!IS_8x8DCT(mt) && (sl->cbp & 15);
        s->mbs_class[7] += !!MB_FIELD(sl);
        s->mbs_bypass += sl->qscale == 0 && h->ps.sps->transform_bypass;
    }
    note(s, ff_h264_hip_hl_decode_mb(&s->rec, h, sl));
}

void ffref_hook_filter_mb_fast(const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y, uint8_t *img_y, uint8_t *img_cb, uint8_t *img_cr,
                               unsigned int linesize, unsigned int uvlinesize)
{
    FFRefH264Stream *s = session_of(h);
    if (!s) {
        ff_h264_filter_mb_fast(h, sl, mb_x, mb_y, img_y, img_cb, img_cr, linesize, uvlinesize);
        return;
    }
    switch (ready(s, h, sl)) {
    case 0: return;
    case -1: ff_h264_filter_mb_fast(h, sl, mb_x, mb_y, img_y, img_cb, img_cr, linesize, uvlinesize); return;
    }
    s->mbs_filter++;
    note(s, ff_h264_hip_filter_mb(&s->rec, h, sl, mb_x, mb_y));
}

void ffref_hook_filter_mb(const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y, uint8_t *img_y, uint8_t *img_cb, uint8_t *img_cr,
                          unsigned int linesize, unsigned int uvlinesize)
{
    FFRefH264Stream *s = session_of(h);
    if (!s) {
        ff_h264_filter_mb(h, sl, mb_x, mb_y, img_y, img_cb, img_cr, linesize, uvlinesize);
        return;
    }
    switch (ready(s, h, sl)) {
    case 0: return;
    case -1: ff_h264_filter_mb(h, sl, mb_x, mb_y, img_y, img_cb, img_cr, linesize, uvlinesize); return;
    }
    s->mbs_filter++;
    note(s, ff_h264_hip_filter_mb(&s->rec, h, sl, mb_x, mb_y));
}

/* ---- the driver ---- */
FFRefH264Stream *ffref_h264stream_open(int record, size_t arena_bytes)
{
    FFRefH264Stream *s = av_mallocz(sizeof(*s));
    if (!s)
        return NULL;
    s->record = record;
    s->arena_size = arena_bytes;
    s->arena = av_malloc(arena_bytes);
    s->pkt = av_packet_alloc();
    s->avctx = avcodec_alloc_context3(&ff_h264_decoder.p);
    if (!s->arena || !s->pkt || !s->avctx)
        return NULL;
    memset(s->arena, 0x55, arena_bytes);
    s->avctx->opaque = s;
    s->avctx->get_buffer2 = arena_get_buffer;
    s->avctx->thread_count = 1;
    s->avctx->flags |= AV_CODEC_FLAG_OUTPUT_CORRUPT;
    s->avctx->err_recognition = AV_EF_EXPLODE | AV_EF_CRCCHECK | AV_EF_BITSTREAM;
    if (avcodec_open2(s->avctx, &ff_h264_decoder.p, NULL) < 0)
        return NULL;
    return s;
}

/* the decoded-picture buffer "allocated" far from the base the records count from: offsets beyond 32 bits must be refused by the
 * recorder (FFHIP_EINVAL), never wrapped */
void ffref_h264stream_set_base_shift(FFRefH264Stream *s, int64_t shift) { s->base_shift = shift; }

void ffref_h264stream_set_flush(FFRefH264Stream *s, ffref_h264_flush_fn fn, void *opaque)
{
    s->flush = fn;
    s->flush_opaque = opaque;
}

void ffref_h264stream_set_flush_mbaff(FFRefH264Stream *s, ffref_h264_flush_mbaff_fn fn, void *opaque)
{
    s->flush_mbaff = fn;
    s->flush_mbaff_opaque = opaque;
}

static int drain_frames(FFRefH264Stream *s)
{
    for (;;) {
        AVFrame *f = av_frame_alloc();
        int r = avcodec_receive_frame(s->avctx, f);
        if (r < 0) {
            av_frame_free(&f);
            return r == AVERROR(EAGAIN) || r == AVERROR_EOF ? 0 : r;
        }
        if (f->decode_error_flags || (f->flags & AV_FRAME_FLAG_CORRUPT))
            s->decode_errors++;
        if (s->nout < MAX_OUT)
            s->out[s->nout++] = f;
        else
            av_frame_free(&f);
    }
}

/* one access unit (Annex B bytes); size 0: end of stream — the decoder is drained and the last picture executed.  0 or an AVERROR. */
int ffref_h264stream_decode(FFRefH264Stream *s, const uint8_t *au, int size)
{
    int r;
    if (size > 0) {
        if (av_new_packet(s->pkt, size) < 0)
            return AVERROR(ENOMEM);
        memcpy(s->pkt->data, au, size);
        r = avcodec_send_packet(s->avctx, s->pkt);
        av_packet_unref(s->pkt);
    } else {
        r = avcodec_send_packet(s->avctx, NULL);
    }
    if (r < 0) {
        s->decode_errors++;
        return r;
    }
    r = drain_frames(s);
    if (size <= 0 && s->record)
        flush_current(s);
    return r;
}

int ffref_h264stream_nframes(const FFRefH264Stream *s) { return s->nout; }

/* output frame i (output order): plane offsets into the arena, line sizes, size in samples, bit depth */
int ffref_h264stream_frame(const FFRefH264Stream *s, int i, int64_t off[3], int linesize[3], int *w, int *h, int *bit_depth)
{
    const AVFrame *f;
    if (i < 0 || i >= s->nout)
        return -1;
    f = s->out[i];
    for (int pl = 0; pl < 3; pl++) {
        off[pl] = f->data[pl] - s->arena;
        linesize[pl] = f->linesize[pl];
    }
    *w = f->width;
    *h = f->height;
    *bit_depth = av_pix_fmt_desc_get(f->format)->comp[0].depth;
    return 0;
}

uint8_t *ffref_h264stream_arena(const FFRefH264Stream *s, size_t *used)
{
    if (used)
        *used = s->arena_used;
    return s->arena;
}

/* counters: 0 pictures recorded, 1 hl_decode_mb calls recorded, 2 filter calls recorded, 3 refused (FFHIP_ENOSYS), 4 recorder / flush
 * errors, 5 the first such error, 6 frames the decoder flagged as damaged, 7 pictures left on the C path as a whole */
long ffref_h264stream_stat(const FFRefH264Stream *s, int what)
{
    switch (what) {
    case 0: return s->pictures;
    case 1: return s->mbs_hl;
    case 2: return s->mbs_filter;
    case 3: return s->refused;
    case 4: return s->errors;
    case 5: return s->first_error;
    case 6: return s->decode_errors;
    case 7: return s->plain_pictures;
    /* 8.. recorded macroblocks: 8 bi-predicted, 9 direct (whole or a sub-macroblock), 10 8x8 transform with coded luma, 11 explicit weights,
     * 12 implicit weights, 13 of B slices, 14 Intra8x8, 15 field macroblocks */
    case 8: case 9: case 10: case 11: case 12: case 13: case 14: case 15: return s->mbs_class[what - 8];
    case 16: return s->mbaff_pictures;   /* MBAFF frames recorded (each counted once in 0 as well) */
    case 17: return s->mbs_bypass;
    }
    return -1;
}

void ffref_h264stream_close(FFRefH264Stream *s)
{
    if (!s)
        return;
    if (s->pic)
        ffhip_h264_picture_free(&s->pic);
    ffhip_h264_picture_free(&s->fpic[0]);
    ffhip_h264_picture_free(&s->fpic[1]);
    ffhip_h264_mbaff_free(&s->chains);
    for (int i = 0; i < s->nout; i++)
        av_frame_free(&s->out[i]);
    avcodec_free_context(&s->avctx);
    av_packet_free(&s->pkt);
    av_free(s->arena);
    av_free(s);
}
