/*
 * ffref_shim_vp9lpf.c — ours, TEST INFRASTRUCTURE ONLY.  Drives the reference's own ff_vp9_loopfilter_sb() (libavcodec/vp9lpf.c:180,
 * compiled where it lies) on one superblock: a zeroed VP9Context with just the fields the function reads (the current frame's
 * planes and line sizes, ss_h / ss_v, bytesperpixel, filter_lut, dsp) filled in from the arguments.  It pins
 * oracle/ffo_vp9.c's ffo_vp9_loopfilter_sb() — which dsp call covers which segments, in which order — to the reference.
 */
#include <stdlib.h>
#include <string.h>

#include "libavcodec/avcodec.h"
#include "libavcodec/vp9dec.h"
#include "libavutil/frame.h"
#include "libavutil/mem.h"

int ffref_vp9_loopfilter_sb(int bpp, int ss_h, int ss_v, const uint8_t *lflvl_level, const uint8_t *lflvl_mask, int row, int col, uint8_t *y,
                            uint8_t *u, uint8_t *v, int ls_y, int ls_uv, const uint8_t *lim_lut, const uint8_t *mblim_lut)
{
    AVCodecContext *avctx = av_mallocz(sizeof(*avctx));
    VP9Context *s = av_mallocz(sizeof(*s));
    AVFrame *f = av_frame_alloc();
    VP9Filter lf;
    if (!avctx || !s || !f)
        abort();
    avctx->priv_data = s;
    s->s.frames[CUR_FRAME].tf.f = f;
    f->data[0] = y;
    f->data[1] = u;
    f->data[2] = v;
    f->linesize[0] = ls_y;
    f->linesize[1] = f->linesize[2] = ls_uv;
    s->ss_h = ss_h;
    s->ss_v = ss_v;
    s->bytesperpixel = bpp > 8 ? 2 : 1;
    memcpy(s->filter_lut.lim_lut, lim_lut, 64);
    memcpy(s->filter_lut.mblim_lut, mblim_lut, 64);
    ff_vp9dsp_init(&s->dsp, bpp, 1);
    memcpy(lf.level, lflvl_level, sizeof(lf.level));
    memcpy(lf.mask, lflvl_mask, sizeof(lf.mask));
    ff_vp9_loopfilter_sb(avctx, &lf, row, col, 0, 0); /* the planes arrive pointing at the superblock: yoff = uvoff = 0 */
    av_frame_free(&f);
    av_free(s);
    av_free(avctx);
    return 0;
}
