/*
 * integration/avcodec_h264_picture_hip.h — libavcodec/hip/h264_picture.h of the FFmpeg-side patch: the macroblock loop of the H.264
 * decoder recorded into a libffhip picture object (include/ffhip.h, FFHipH264Picture) instead of executed block by block.
 */
#ifndef FFHIP_INTEGRATION_H264_PICTURE_HIP_H
#define FFHIP_INTEGRATION_H264_PICTURE_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "libavcodec/h264dec.h"

#include "ffhip.h"

/* One recorder per slice thread (the decoder's H264SliceContext owns the buffers it watches). */
typedef struct FFHipH264Recorder {
    FFHipH264Picture *pic;
    int pixel_shift;
    int cfmt;                       /* sps->chroma_format_idc: 1 (also for monochrome: mid-grey chroma), 2 or 3 (Cb / Cr through the luma members) */
    int field;                      /* a field picture (PAFF): `pic` is the FIELD — every second line of the frame buffer from cur[] on */
    int error;                      /* first libffhip error (< 0), sticky until begin() */
    /* the picture being decoded: h->cur_pic.f->data[] as the DEVICE addresses of the hip frame, and the base every reference
     * picture's data[] is counted from (the decoded-picture-buffer allocation: what ffhip_h264_picture_flush() gets as ref[]).
     * ONE allocation: the records hold 32-bit offsets from ref_base[] — the current picture and all its references must lie within
     * 2 GiB of it; a picture or a reference outside that reach is refused (FFHIP_EINVAL), never wrapped */
    const uint8_t *cur[3];
    const uint8_t *ref_base[3];
    ptrdiff_t linesize[3];
    int rows[3];
    int pic_w[3];                   /* samples per row of the plane */
    const H264SliceContext *sl;     /* the slice context of the running hl_decode_mb() call (its reference lists) */
    const uint8_t *last_ref[3];     /* data[pl] of the reference picture the last luma-table block of the plane read */
    const uint8_t *scratch;         /* sl->bipred_scratchpad and its size: never dereferenced, only recognised */
    size_t scratch_size;
    const uint8_t *emu_buf;         /* sl->edge_emu_buffer, likewise */
    size_t emu_size;
    /* the last h->vdsp.emulated_edge_mc() call: the next qpel / chroma call that reads sl->edge_emu_buffer stands for it */
    struct {
        const uint8_t *src;
        ptrdiff_t linesize;
        int src_x, src_y, valid;
    } emu;
    /* predictions put into sl->bipred_scratchpad wait here for the biweight call that names their destination */
    struct FFHipH264Pending {
        int plane, luma_tab;        /* luma_tab: a qpel record (q), else a chroma MC record (c) */
        const uint8_t *tmp;
        FFHipQpelBlock q;
        FFHipChromaBlock c;
    } pend[8];
    int npend;
    /* an MBAFF frame (round 6; 4:2:0): FOUR objects over the same planes.  view[0] the frame macroblocks, view[1] / view[2] the top- /
     * bottom-field macroblocks (half the rows at twice the line size; the bottom one a frame line further down) — the object a macroblock is
     * recorded into is chosen per macroblock and copied into pic / cur / linesize / rows above, so everything that serves a frame or a
     * field picture serves a macroblock of either kind; `chains` takes the intra macroblocks and the loop filter's calls */
    int mbaff;
    int active;                     /* the view pic / cur / linesize / rows currently stand for */
    struct FFHipH264View {
        FFHipH264Picture *pic;
        const uint8_t *cur[3];
        ptrdiff_t linesize[3];
        int rows[3];
    } view[3];
    FFHipH264Mbaff *chains;
} FFHipH264Recorder;

/* Replaces, in h, the dsp members hl_decode_mb() calls for inter macroblocks (h264qpel, h264chroma, weight / biweight, idct_add16 /
 * idct8_add4 / idct_add8, add_pixels4_clear / add_pixels8_clear, vdsp.emulated_edge_mc, vdsp.prefetch) and the loop-filter members ff_h264_filter_mb() calls with recording
 * ones; chroma_dc_dequant_idct and everything else stay what ff_h264dsp_init() left.  Call once after the decoder's own init. */
void ff_h264_hip_recorder_install(H264Context *h);

/* 1 when the picture the decoder is about to decode can be recorded as a whole: a lossless (transform-bypass) stream only at 8 bits; an MBAFF frame (FRAME_MBAFF(h): begin it with ff_h264_hip_recorder_begin_mbaff()) only at 4:2:0.  Ask
 * before ff_h264_hip_recorder_begin(): a refusal in the middle of a picture cannot be undone (the per-macroblock calls still return
 * FFHIP_ENOSYS for such macroblocks, as a guard). */
int ff_h264_hip_picture_supported(const H264Context *h);

/* A new picture: `pic` was made for h->mb_width x h->mb_height at the stream's bit depth and has had begin() called.
 * A FIELD picture (h->picture_structure != PICT_FRAME, no MBAFF): `pic` was made for h->mb_width x h->mb_height / 2 — the field is a picture
 * of its own whose lines are every second line of the frame buffer: flush() gets data[pl] (+ one line for the bottom field) as dst[pl]
 * and TWICE the frame's line sizes as strides; the references' fields are addressed the same way through ref_base[]. */
void ff_h264_hip_recorder_begin(FFHipH264Recorder *r, FFHipH264Picture *pic, const H264Context *h, const H264SliceContext *sl,
                                const uint8_t *const ref_base[3]);

/* A new MBAFF frame (FRAME_MBAFF(h)): frame_mbs / top_mbs / bottom_mbs are picture objects made for h->mb_width x h->mb_height and (the two
 * field ones) h->mb_width x h->mb_height / 2, `chains` a FFHipH264Mbaff made for h->mb_width x h->mb_height, all at the stream's bit depth; all
 * four have had begin() called.
 * When the frame is complete: ffhip_h264_picture_flush() of the three — frame_mbs with data[] and the frame's line sizes, top_mbs with data[]
 * and TWICE the line sizes, bottom_mbs with data[] + one line and twice the line sizes — then ffhip_h264_mbaff_flush(chains, data[], the
 * frame's line sizes), on one stream. */
void ff_h264_hip_recorder_begin_mbaff(FFHipH264Recorder *r, FFHipH264Picture *frame_mbs, FFHipH264Picture *top_mbs, FFHipH264Picture *bottom_mbs,
                                      FFHipH264Mbaff *chains, const H264Context *h, const H264SliceContext *sl, const uint8_t *const ref_base[3]);

/* ff_h264_hl_decode_mb(h, sl) with the dsp calls recorded into r->pic (intra macroblocks: one FFHipH264IntraMB record).  Returns 0 or
 * the first libffhip error. */
int ff_h264_hip_hl_decode_mb(FFHipH264Recorder *r, const H264Context *h, H264SliceContext *sl);

/* ff_h264_filter_mb(h, sl, mb_x, mb_y, …) with the loop-filter calls recorded as the macroblock's edge records. */
int ff_h264_hip_filter_mb(FFHipH264Recorder *r, const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y);

#endif
