/* sws_kernels.h — argument blocks and launchers of the swscale kernels (internal to libffhip). */
#ifndef FFHIP_SWS_KERNELS_H
#define FFHIP_SWS_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

/* closed-form yuv2rgb constants, all int32 (checked at context creation) */
struct FFHipYuv2RgbK {
    int cy;
    int kb;           /* yb0 + 0x8000 */
    int crv, cbu, cgu, cgv;
    int off_r, off_b; /* yoffs - (crv>>9), yoffs - (cbu>>9)            */
    int off_g;        /* yoffs - (cgu>>9) - (cgv>>9)                   */
};

struct FFHipYuv2RgbArgs {
    const uint8_t *y, *u, *v;
    uint8_t *dst;
    ptrdiff_t y_stride, u_stride, v_stride, dst_stride;
    size_t y_fp, u_fp, v_fp, dst_fp; /* frame pitches (bytes) */
    int wvalid;                      /* width & ~1 */
    int h;                           /* rows of this slice (even) */
    int dst_y0;                      /* srcSliceY: first destination row */
    int nframes;
    int flat;                        /* set by the launcher: chunks numbered through the frame's row pairs (k_yuv420p_rgb24_t) */
    int xcd;                         /* workgroup numbering of k_yuv420p_rgb24_t: 0 plain; 1 an eighth of the launch per XCD (chosen per context by the
                                      * launch tuner of sws_api.hip: which of the two is faster depends on the box, profiles/r06_arena_offset_sweep.txt);
                                      * 1 + k (measure build): XCD-contiguous chunks of 2^k workgroups dealt round-robin */
    FFHipYuv2RgbK k;
    /* the converter's other forms (yuv2rgb.c:238-320, 540-553: YUV422FUNC, yuva2rgba_c / yuva2argb_c, yuv420p_gbrp_c) */
    int c422 = 0;                    /* 4:2:2 source: luma row 2k + 1 takes chroma row 2k + 1 (u_stride / v_stride are the planes' own) */
    const uint8_t *alpha = nullptr;  /* the source's alpha plane drives the alpha byte of a 32-bit target (else 255) */
    ptrdiff_t alpha_stride = 0;
    size_t alpha_fp = 0;
    uint8_t *dst1 = nullptr, *dst2 = nullptr; /* layout 6 (gbrp): dst = G, dst1 = B, dst2 = R planes */
    ptrdiff_t dst1_stride = 0, dst2_stride = 0;
    size_t dst1_fp = 0, dst2_fp = 0;
};
/* packed layout: 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra (32-bit: alpha = 255 unless a.alpha), 6 planar gbrp */
int ffhip_launch_yuv420p_rgb24(const FFHipYuv2RgbArgs &a, int layout, hipStream_t stream);

/* one separable bank resident in HBM */
struct FFHipDevFilter {
    const int16_t *filter;
    const int32_t *pos;
    int size, n;
};

/*
 * Fused H+V scaling of one "channel group": C == 1 (a luma plane, or one planar chroma plane) or
 * C == 2 (U and V together, each possibly interleaved with the other in memory).
 */
struct FFHipScalePlaneArgs {
    const uint8_t *src[2];      /* channel c sample i at src[c][i * src_step] */
    uint8_t *dst[2];
    ptrdiff_t src_stride[2], dst_stride[2];
    size_t src_fp[2], dst_fp[2];
    int src_step, dst_step;     /* 1 planar, 2 interleaved */
    int srcW, srcH, dstW, dstH; /* of this channel group */
    int nframes;
    FFHipDevFilter h, v;
    int tw, th;                 /* output tile */
    int max_cols, max_rows;     /* source footprint of the widest / tallest tile */
    int tiles_x, tiles_y;
    /* range conversion of the horizontal intermediates (lum / chrRangeTo / FromJpeg_c, swscale.c:160-207): v = (v * rc_coeff +
     * rc_offset) >> 14, clipped to 2^15 - 1 when rc_clip (ToJpeg); rc_coeff == 0: none */
    int rc_coeff, rc_offset, rc_clip;
};
int ffhip_launch_scale_yuv(const FFHipScalePlaneArgs &lum, const FFHipScalePlaneArgs &chr, hipStream_t stream);
/* picks tw/th/max_* for a bank pair from HOST copies of the position tables */
int ffhip_plan_scale_plane(FFHipScalePlaneArgs *a, int channels, const int32_t *hpos_host, const int32_t *vpos_host);


/*
 * Column-walking fast path (sws_colwalk.hip) for 4-tap x 4-tap banks.  A launch runs up to three
 * jobs: kind 0/1 = one plane (1 or 2 column groups per lane), kind 2/3/4 = a U/V pair that is
 * byte-interleaved (NV12/NV21) on both sides / on the source only / on the destination only.
 */
struct FFHipCwJob {
    const uint8_t *src[2];  /* kind 2: U,V planes, or src[0] = the interleaved plane when src_il   */
    uint8_t *dst[2];
    ptrdiff_t sstride[2], dstride[2];
    size_t sfp[2], dfp[2];
    int kind, src_swap, dst_swap;               /* *_swap: V is the first byte (NV21)             */
    int srcW, srcH, dstW, dstH;                   /* in samples of this channel                      */
    const int16_t *hf; const int32_t *hp;         /* device banks                                    */
    const int16_t *vf; const int32_t *vp;
    int ncb, nstrips, strip_rows, unit_begin;
};
struct FFHipCwArgs {
    FFHipCwJob job[3];
    int njobs, units_per_frame, nframes, flags;   /* bit0: plain shift/clamp instead of v_ashr_pk_u8_i32; bit1: OPT variant; bit2: DUP (needs OPT) */
};
/* 1 when no horizontal sum of the bank can fall below -32768 after >> 7 (then int16 saturation == the
 * reference's min(.,32767) + truncation, which the OPT variant relies on) */
int  ffhip_cw_bank_nowrap(const int16_t *filter, int size, int n);
int  ffhip_cw_bank_dup12(const int32_t *hpos, int hn);
int  ffhip_cw_bank_ok(const int32_t *hpos, int hsize, int hn, int srcW, const int32_t *vpos, int vsize, int vn, int srcH);
void ffhip_cw_plan_job(FFHipCwJob *j, int groups_per_lane, int strip_target);
int  ffhip_launch_colwalk(FFHipCwArgs &A, int luma_groups, int depth, hipStream_t stream);

/*
 * Exact-2x fast path (sws_up2.hip): 4-tap banks re-expressed on the regular windows of the edge-replicated rows.
 * A job is one plane (pair 0: 8 output columns per lane) or one byte-interleaved U/V pair (pair 1: 4 + 4 per lane).
 */
struct FFHipUp2Job {
    const uint8_t *src; uint8_t *dst;   /* pair: the interleaved plane (the lower of the two channel pointers) */
    ptrdiff_t sstride, dstride;
    size_t sfp, dfp;
    int pair, swap;                     /* swap: the channel at the EVEN destination bytes sits at the ODD source bytes */
    int srcW, srcH;                     /* samples per channel; the destination is 2 srcW x 2 srcH */
    int ngroups;                        /* 8-byte destination groups per row: plane srcW / 4, pair srcW / 2 */
    const uint32_t *hfv;                /* device: virtual horizontal bank, 2 srcW x 2 dwords */
    const uint32_t *vfv;                /* device: virtual vertical bank, row y at dwords 2 (y + 1): (2 srcH + 18) x 2 dwords, 16-byte aligned */
    int nfull, upj;                     /* full 64-lane column blocks per row; units per (pack of frames, strip) */
    int nstrips, steps_per_strip, unit_begin;
    /* samples above 8 bits (k_sws_up2<., ., 1>; 0: bytes): depths 9..14 of the little-endian uint16 samples on either side, and
     * whether they sit in the high bits (P010 / P012).  Groups are then 16 destination bytes. */
    int hb_sdepth, hb_ddepth, hb_smsb, hb_dmsb;
    /* range conversion on the 15-bit horizontal samples (lum / chrRangeToJpeg_c, ...FromJpeg_c, libswscale/swscale.c:160-207):
     * h = (h * rc_coeff + rc_offset) >> 14, clipped to 32767 (k_sws_up2<., ., 0, 1>; rc_coeff 0: none) */
    int rc_coeff, rc_offset;
    /* round 6: the horizontal bank as scalars (k_sws_up2<..., SC = 1>): every column but the three next to either end of a row has the
     * coefficients of its parity — hco[0..1] an even column's two dwords, [2..3] an odd one's, [4..9] columns 0, 1, 2, [10..15] the last
     * three; hco_ok: the bank has that shape (ffhip_up2_hco) */
    uint32_t hco[16];
    int hco_ok;
};
struct FFHipUp2Args {
    FFHipUp2Job job[3];
    int njobs, units_per_pack, npacks, nframes, fshift; /* a pack = 1 << fshift frames; they share the ragged-end blocks (64 >> fshift lanes each) */
    int xcd;                                            /* XCD-contiguous unit order */
};
#ifdef __cplusplus
#include <vector>
int  ffhip_up2_virtual_bank(const int16_t *filter, const int32_t *pos, int n_dst, int n_src, std::vector<uint32_t> *out);
int  ffhip_up2_hco(const std::vector<uint32_t> &h, uint32_t out[16]);
#endif
void ffhip_up2_plan_job(FFHipUp2Job *j, int lanes_per_frame, int want_steps);
int  ffhip_launch_up2(FFHipUp2Args &A, int depth, int var, hipStream_t stream);

/*
 * Exact-2x up-scaling of 4:2:0 (planar yuv420p or NV12 / NV21) into packed RGB (sws_up2rgb.hip): 4-tap banks on all four axes re-expressed on the regular
 * windows of the edge-replicated rows — luma 2x both ways, chroma 2x horizontally and (chrDstH == dstH) 4x vertically.
 */
struct FFHipUp2RgbArgs {
    const uint8_t *src[3];      /* Y, U, V planes; sil: src[1] = the byte-interleaved chroma plane (NV12; swap: NV21) */
    uint8_t *dst;
    ptrdiff_t sstride[3], dstride;
    size_t sfp[3], dfp;
    int sil, swap;
    int srcW, srcH;             /* luma; the picture is 2 srcW x 2 srcH, chroma srcW / 2 x srcH / 2 */
    int ngroups;                /* 8-pixel groups per output row: srcW / 4 (even: the chroma rows end on a dword) */
    int nframes;
    const uint32_t *hco;        /* device: the horizontal virtual banks in 32 dwords (ffhip_up2rgb_hco) */
    const uint32_t *vt;         /* device: row y at dwords 4 (y + 1): (luma c01, luma c23, chroma c01, chroma c23); rows -1 and >= dstH zero */
    int nstrips, steps_per_strip;
    int fpp, wpp, npacks;       /* frames per pack (1, 2, 4: their groups share the pack's waves lane by lane), waves per pack and strip, packs */
    int vround;                 /* seed of the vertical sums: 1 << 18 (yuv2rgb_X) */
    int lay;                    /* 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra */
    FFHipYuv2RgbK k;
};
#ifdef __cplusplus
int  ffhip_upn_virtual_bank(const int16_t *filter, const int32_t *pos, int n_dst, int n_src, int ratio, std::vector<uint32_t> *out);
int  ffhip_up2rgb_hco(const std::vector<uint32_t> &hl, const std::vector<uint32_t> &hc, uint32_t out[32]);
#endif
void ffhip_up2rgb_plan(FFHipUp2RgbArgs *a, int want_steps, int fpp_forced);
int  ffhip_launch_up2rgb(FFHipUp2RgbArgs &A, int var, hipStream_t stream);

/*
 * Exact-2:1 fast path (sws_down2.hip): banks of up to 8 taps re-expressed on the regular windows 2x - 3 .. 2x + 4 of the
 * edge-replicated rows.  A job is one plane (4 output columns per lane) or one byte-interleaved U/V pair (2 + 2 per lane).
 */
struct FFHipDn2Job {
    const uint8_t *src; uint8_t *dst;   /* pair: the interleaved plane (the lower of the two channel pointers) */
    ptrdiff_t sstride, dstride;
    size_t sfp, dfp;
    int pair, swap;                     /* swap: the channel at the EVEN destination bytes sits at the ODD source bytes */
    int srcH, dstH;                     /* source rows; output rows = srcH / 2 */
    int ngroups;                        /* 4-byte destination groups per row: plane dstW / 4, pair dstW / 2 (>= 3) */
    const uint32_t *hfv;                /* device: virtual horizontal bank, dstW x 4 dwords */
    const uint32_t *vfv;                /* device: virtual vertical bank, (dstH + 8) x 4 dwords, 64-byte aligned */
    int ncb, nstrips, steps_per_strip, unit_begin;
    int hb_sdepth, hb_ddepth, hb_smsb, hb_dmsb; /* samples above 8 bits (k_sws_down2<1>; 0: bytes), as in FFHipUp2Job; groups are 8 destination bytes
                                         * (hb_ddepth == 8: an 8-bit target with the ordered dither, 4 destination bytes) */
    int y16;                            /* 8-bit plane job: the vertical sums >> 19 stored UNCLIPPED as int16 (8 bytes per group): see FFHipLwJob.y16 */
    int dither_off;                     /* hb_ddepth == 8, plane job: the dither column offset (3 for the V plane) */
    int v1;                             /* 8-bit job without a vertical filter: dstH = srcH rows, each clip_u8((horizontal sum + 64) >> 7) (sws_down2.hip) */
};
struct FFHipDn2Args {
    FFHipDn2Job job[3];
    int njobs, units_per_frame, nframes;
    int xcd;                            /* XCD-contiguous unit order */
};
#ifdef __cplusplus
int  ffhip_down2_virtual_bank(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, std::vector<uint32_t> *out);
#endif
/*
 * Exact 3:2 down-scaling (sws_down32.hip): a job is one plane or one byte-interleaved U/V pair (NV12 / NV21 in and out).
 */
struct FFHipD32Job {
    const uint8_t *src; uint8_t *dst;   /* pair: the interleaved plane (the lower of the two channel pointers) */
    ptrdiff_t sstride, dstride;
    size_t sfp, dfp;
    int pair, swap;                     /* swap: the channel at the EVEN destination bytes sits at the ODD source bytes */
    int srcH, dstH;                     /* source rows; output rows = srcH * 2 / 3, even */
    int ngroups;                        /* 8-byte destination groups per row: plane dstW / 8, pair dstW / 4 (>= 3) */
    const uint32_t *hfv;                /* device: virtual horizontal bank, dstW x 3 dwords */
    const uint32_t *vfv;                /* device: virtual vertical bank, dstH x 4 dwords */
    int ncb, nstrips, strip_rows, unit_begin;
    int dither_off;                     /* the 16-bit twin into an 8-bit target, plane jobs: 3 for the V plane (its dither row is read three entries on) */
};
struct FFHipD32Args {
    FFHipD32Job job[3];
    int njobs, units_per_frame, nframes;
    /* round 6, the 16-bit twin (k_sws_down32h): samples of 9..14 bits on both sides, as FFHipUp2Job.hb_*; groups are then 8 destination bytes
     * too — plane dstW / 4, pair dstW / 2 */
    int hb, sdepth, ddepth, smsb, dmsb;
    int ratio43;                        /* the twin's second period: 4 in, 3 out (1440p -> 1080p); groups of 12 destination bytes */
};
#ifdef __cplusplus
int  ffhip_d32_virtual_bank(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, int pitch, std::vector<uint32_t> *out, int pin = 3, int pout = 2);
#endif
int  ffhip_launch_down32(FFHipD32Args &A, hipStream_t stream);

/*
 * Exact 3:2 up-scaling of 9..14-bit samples (sws_up32.hip): a job is one plane of words or one interleaved (u, v) plane of words (P01x in
 * and out).
 */
struct FFHipU32Job {
    const uint8_t *src; uint8_t *dst;
    ptrdiff_t sstride, dstride;
    size_t sfp, dfp;
    int pair;
    int srcH, dstH;                     /* source rows (even); output rows = srcH * 3 / 2 */
    int ngroups;                        /* 12-byte destination groups per row: plane dstW / 6, pair dstW / 3 (>= 3) */
    const uint32_t *hfv;                /* device: virtual horizontal bank, dstW x 2 dwords */
    const uint32_t *vfv;                /* device: virtual vertical bank, dstH x 2 dwords */
    int ncb, nstrips, strip_rows, unit_begin;
};
struct FFHipU32Args {
    FFHipU32Job job[3];
    int njobs, units_per_frame, nframes;
    int sdepth, ddepth, smsb, dmsb;     /* as FFHipUp2Job.hb_* */
    int bytes;                          /* round 6, the 8-bit twin (k_sws_up32b): planes of bytes / NV12 pairs, groups of 4 POUT destination bytes */
    int ratio43;                        /* 0: 3:2 (period 2 in, 3 out; groups of 6 / 3 outputs); 1: 4:3 (3 in, 4 out; groups of 8 / 4: 16 destination bytes) */
};
#ifdef __cplusplus
int  ffhip_u32_virtual_bank(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, int pin, int pout, std::vector<uint32_t> *out);
#endif
int  ffhip_launch_up32(FFHipU32Args &A, hipStream_t stream);

/* exact 2:1 from NV12 / NV21 into packed RGB, fused (k_sws_down2_rgb in sws_down2.hip) */
struct FFHipDn2RgbArgs {
    const uint8_t *ysrc, *csrc;         /* the luma plane; the interleaved chroma plane, or the U plane */
    const uint8_t *csrc2;               /* planar chroma: the V plane (same stride and frame pitch as U's), else null */
    uint8_t *dst;
    ptrdiff_t ysstride, csstride, dstride;
    size_t ysfp, csfp, dfp;
    int swap;                           /* NV21: v at the even bytes (interleaved chroma only) */
    int srcH, chrH, dstH;               /* luma rows, chroma rows (= dstH), output rows (= srcH / 2) */
    int ngroups;                        /* groups of four pixels per output row: dstW / 4 (>= 3) */
    const uint32_t *hfv_l, *hfv_c;      /* device: virtual horizontal banks, dstW x 4 and (dstW / 2) x 4 dwords */
    const uint32_t *vfv;                /* device: the luma's virtual vertical bank, (dstH + 8) x 4 dwords */
    int ncb, nstrips, steps_per_strip, nframes, xcd, lay;
    FFHipYuv2RgbK k;
};
int  ffhip_launch_down2_rgb(FFHipDn2RgbArgs &A, int want_rows, hipStream_t stream);
void ffhip_down2_plan_job(FFHipDn2Job *j, int want_rows);
int  ffhip_launch_down2(FFHipDn2Args &A, hipStream_t stream);

/*
 * MFMA-horizontal variant of the fast path (k_sws_mfma in sws_colwalk.hip): a job is one plane or one
 * byte-interleaved U/V pair (NV12/NV21 in and out).
 */
struct FFHipMfJob {
    const uint8_t *src; uint8_t *dst;
    ptrdiff_t sstride, dstride;
    size_t sfp, dfp;
    int pair, dst_swap;
    int srcH, dstW, dstH;        /* dstW in samples per channel */
    const uint8_t *tiles;        /* device: ntiles records of 2320 bytes (ffhip_mf_build_tiles) */
    const int16_t *vf; const int32_t *vp;
    const int32_t *ys;           /* device: ys[p] = first output row whose window starts at source row >= p; ys[srcH] = dstH */
    int ntiles, ncb, nstrips, strip_rows, unit_begin;
};
struct FFHipMfArgs {
    FFHipMfJob job[3];
    int njobs, units_per_frame, nframes;
};
#ifdef __cplusplus
#include <vector>
int ffhip_mf_build_tiles(std::vector<uint8_t> *out, const int16_t *hf, const int32_t *hp, int n, int srcW, int pair, int src_swap);
#endif
int ffhip_launch_mfma(FFHipMfArgs &A, hipStream_t stream);

/* Fused H+V scaling + yuv2rgb for packed rgb24/bgr24 output (yuv2rgb{1,2,X} dispatch in-kernel). */
struct FFHipScaleRgbArgs {
    const uint8_t *src[3];      /* Y, U, V (U/V may alias an interleaved plane with chr_step 2) */
    uint8_t *dst;
    ptrdiff_t src_stride[3], dst_stride;
    size_t src_fp[3], dst_fp;
    int chr_step;
    int srcW, srcH, chrSrcW, chrSrcH, dstW, dstH;
    int nframes;
    FFHipDevFilter hl, hc, vl, vc;
    int tw, th;
    int max_cols_l, max_rows_l, max_cols_c, max_rows_c;
    int tiles_x, tiles_y;
    int bgr;
    FFHipYuv2RgbK k;
    int full;           /* SWS_FULL_CHR_H_INT: hc has dstW entries, the yuv2rgb_full_* writers with fk[] (FFHipSwsTables.yuv2rgb_full) */
    int fk[6];
    /* a source alpha plane into the alpha byte of a 32-bit target (yuv2rgba32_{1,2,X}_c / the _full twins, libswscale/output.c:
     * 1789-1939, 2160-2310): the plane is scaled by the LUMA banks beside Y (lum_h_scale / the writers' alpSrc lines) */
    int has_alpha = 0;          /* set before ffhip_plan_scale_rgb(): the tiles make room for the alpha plane */
    const uint8_t *alpha = nullptr;
    ptrdiff_t alpha_stride = 0;
    size_t alpha_fp = 0;
};
int ffhip_launch_scale_rgb(const FFHipScaleRgbArgs &a, hipStream_t stream);
int ffhip_plan_scale_rgb(FFHipScaleRgbArgs *a, const int32_t *hl, const int32_t *hc, const int32_t *vl,
                         const int32_t *vc);

/*
 * Column walker with packed rgb24/bgr24 output (k_sws_colwalk_rgb): 4-tap banks on all four axes, dstW % 8 == 0,
 * chroma vertical bank of dstH rows (chrDstH == dstH), chroma horizontal bank of dstW / 2 columns.
 */
struct FFHipCwRgbArgs {
    const uint8_t *src[3];     /* Y, U, V; sil: src[1] = the byte-interleaved chroma plane */
    uint8_t *dst;
    ptrdiff_t sstride[3], dstride;
    size_t sfp[3], dfp;
    int sil, src_swap, bgr;
    int srcW, srcH, chrSrcW, chrSrcH, dstW, dstH;
    const int16_t *hlf; const int32_t *hlp; const int16_t *hcf; const int32_t *hcp;
    const int16_t *vlf; const int32_t *vlp; const int16_t *vcf; const int32_t *vcp;
    int ncb, nstrips, strip_rows, nframes;
    int vround;                /* seed of the vertical sums: 1 << 18 (yuv2rgb_X, and _1 which equals it), 0 (yuv2rgb_2) */
    FFHipYuv2RgbK k;
    int nts;                   /* non-temporal stores of the picture (round 5; measure build: FFHIP_CWRGB_NTS=0 turns them off) */
};
int ffhip_launch_colwalk_rgb(FFHipCwRgbArgs &A, hipStream_t stream);

/*
 * Wide-bank walker (sws_lwalk.hip): banks padded to 4*ht horizontal and 2*vt vertical taps (ht 2|4, vt 4|8).
 * A job is one plane or (pair) a U/V pair, byte-interleaved or planar on either side.
 */
struct FFHipLwJob {
    const uint8_t *src[2];   /* sil: src[0] = the interleaved plane */
    uint8_t *dst[2];         /* dil: dst[0] = the interleaved plane */
    ptrdiff_t sstride[2], dstride[2];
    size_t sfp[2], dfp[2];
    int pair, sil, dil, src_swap, dst_swap;
    int srcW, srcH, dstW, dstH;                 /* in samples of this channel */
    const int16_t *hf; const int32_t *hp;       /* device: padded banks */
    const int16_t *vf; const int32_t *vp;
    int ncb, nstrips, strip_rows, unit_begin;
    int y16;                                    /* plane jobs: the vertical sums >> 19 stored UNCLIPPED as int16 (dstride in bytes): the luma
                                                 * of a packed-RGB target's first stage (sws_y16rgb.hip is the second) */
};
struct FFHipLwArgs {
    FFHipLwJob job[3];
    int njobs, units_per_frame, nframes, ht, vt;
};
int  ffhip_lw_bank_ok(const int32_t *hpos, int ht, int hn, int srcW, const int32_t *vpos, int vt, int vn, int srcH, int pair);
void ffhip_lw_plan_job(FFHipLwJob *j);
int  ffhip_launch_lwalk(FFHipLwArgs &A, hipStream_t stream);

/*
 * 4:2:0 (planar or NV12 / NV21) into packed RGB at the source's size through the scaler's arithmetic (sws_eqrgb.hip): one-tap luma
 * and horizontal banks, the 4-tap vertical chroma bank of an exact 2x (a chroma line per output line).
 */
struct FFHipEqRgbArgs {
    const uint8_t *src[3];      /* Y, U, V planes; sil: src[1] = the byte-interleaved chroma plane (NV12; swap: NV21) */
    uint8_t *dst;
    ptrdiff_t sstride[3], dstride;
    size_t sfp[3], dfp;
    int sil, swap;
    int chrH;                   /* chroma rows; the picture has 2 chrH rows */
    int ngroups;                /* 8-pixel groups per row: width / 8 */
    int nframes;
    const uint32_t *vt;         /* device: the virtual vertical chroma bank, row y at dwords 2 (y + 1): (c01, c23); rows -1 and >= dstH zero */
    int nstrips, steps_per_strip;
    int fpp, wpp, npacks;       /* frames per pack (1, 2, 4), waves per pack and strip, packs: as FFHipUp2RgbArgs */
    int vround;                 /* seed of the vertical sums: 1 << 18 (yuv2rgb_X) */
    int lay;                    /* 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra */
    FFHipYuv2RgbK k;
};
void ffhip_eqrgb_plan(FFHipEqRgbArgs *a, int want_steps, int fpp_forced);
int  ffhip_launch_eqrgb(FFHipEqRgbArgs &A, hipStream_t stream);

/*
 * Second stage of a scaled packed-RGB target that has no fused kernel (sws_y16rgb.hip): the scaler's output at the target's own
 * geometry — luma w x h as UNCLIPPED int16, chroma w / 2 x h (a chroma line per output line, half the columns: what yuv2packedX
 * is handed) — through the yuv2rgb tables' closed form into one of the six packed layouts.
 */
struct FFHipY16RgbArgs {
    const uint8_t *y, *u, *v;   /* y: int16 samples */
    uint8_t *dst;
    ptrdiff_t ystride, cstride, dstride;
    size_t yfp, cfp, dfp;
    int w, h, nframes, lay;     /* w even (a chroma sample per pixel pair); lay: 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra */
    int uvi;                    /* u: ONE plane of (u, v) byte pairs (v unused) */
    FFHipYuv2RgbK k;
};
int ffhip_launch_y16_rgb(const FFHipY16RgbArgs &a, hipStream_t stream);

/* planar 4:4:4 into packed RGB at the source's size: yuv2rgb_full_1_c + yuv2rgb_write_full on one-tap banks (sws_full444.hip) */
struct FFHipFull444Args {
    const uint8_t *src[3];      /* Y, U, V planes */
    uint8_t *dst;
    ptrdiff_t sstride[3], dstride;
    size_t sfp[3], dfp;
    int w, h, nframes, lay;     /* w >= 8; lay: 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra */
    int fk[6];                  /* FFHipSwsTables.yuv2rgb_full: y_coeff, y_offset, v2r, v2g, u2g, u2b */
};
int ffhip_launch_full444(const FFHipFull444Args &a, hipStream_t stream);

/* 4:2:0 between its planar and semi-planar layouts at the same size (sws_copy420.hip): a job fills one destination plane */
struct FFHipCopy420Job {
    const uint8_t *src[2];      /* WEAVE: the planes of the even / odd destination bytes; else src[0] */
    uint8_t *dst;
    ptrdiff_t sstride[2], dstride;
    size_t sfp[2], dfp;
    int kind, k;                /* 0 copy, 1 pick channel k of (a, b) pairs, 2 weave two planes into pairs, 3 swap the bytes of every pair */
    int wbytes, rows;           /* the destination row in bytes (>= 16; weave / swap: even) */
    int ncb, unit_begin;
};
struct FFHipCopy420Args {
    FFHipCopy420Job job[3];
    int njobs, units_per_frame, nframes;
};
int ffhip_launch_copy420(FFHipCopy420Args &A, hipStream_t stream);

/*
 * The column walker above 8 bits (sws_walk16.hip): banks padded to ht, vt in {4, 8} taps.  A job is one plane (nch 1) or the two
 * chroma channels together (nch 2: an interleaved (u, v) plane on the source and / or the target side; a planar side has the two
 * planes in src[] / dst[]).  Samples are little-endian uint16 of 9..14 bits, in the high bits for P01x.
 */
struct FFHipW16Job {
    const uint8_t *src[2]; uint8_t *dst[2];
    ptrdiff_t sstride[2], dstride[2];
    size_t sfp[2], dfp[2];
    int nch, sstep, dstep;                    /* sample step inside a row: 1 planar, 2 interleaved */
    int srcH, dstW, dstH;                     /* in samples of one channel */
    const int16_t *hf; const int32_t *hp;     /* device: dstW x ht taps, positions in samples of the channel */
    const int16_t *vf; const int32_t *vp;     /* device: dstH x vt taps */
    int ncb, nstrips, strip_rows, unit_begin;
    int dither_off;                           /* ddepth 8, plane jobs: 3 for the V plane (its dither row is read three entries on), else 0 */
    int y16;                                  /* ddepth 8, a plane job: the vertical sums >> 19 stored UNCLIPPED as int16 (8 bytes per lane): the luma of a
                                               * packed-RGB target's first stage, as FFHipLwJob.y16 */
    int srcW;                                 /* samples of one channel per source row */
    int stage;                                /* round 6: the span of a wave's windows in a source row fits 1 KiB (and the positions ascend): the row
                                               * segment is loaded ONCE per wave into LDS and the lanes' windows are read from there */
};
struct FFHipW16Args {
    FFHipW16Job job[3];
    int njobs, units_per_frame, nframes, ht, vt;
    int sdepth, ddepth, smsb, dmsb;
    int flat_dither;                          /* ddepth 8: 1: every dither entry is 64; 2: every entry is 0 (yuv2rgb_2's sums carry no rounding term); 1 is (c->lumDither8 = ff_sws_pb_64: the source is not one swscale dithers — a
                                               * packed-RGB source's converter output, swscale.c:291) */
};
#ifdef __cplusplus
bool ffhip_w16_pad_bank(const int16_t *filter, const int32_t *pos, int size, int n, int nsrc, int T, std::vector<int16_t> *of, std::vector<int32_t> *op);
#endif
void ffhip_w16_plan_job(FFHipW16Job *j, int strip_target);
#ifdef __cplusplus
int  ffhip_w16_span(const int32_t *pos, int n, int cols, int bytes_per_column, int taps);
#endif
int  ffhip_launch_walk16(FFHipW16Args &A, hipStream_t stream);

/* the scaler above 8 bits (sws_scale16.hip): one record per output plane (an interleaved UV plane is two records, one per channel) */
struct FFHipScale16Plane {
    const uint8_t *src;      /* source plane of this channel */
    uint8_t *dst;
    ptrdiff_t src_stride, dst_stride;
    size_t src_fp, dst_fp;   /* frame pitches */
    int sdepth, sstep, schan, smsb;   /* source samples: depth, sample step (2: interleaved pair), channel inside the pair, samples in the high bits */
    int ddepth, dstep, dchan, dmsb;
    int dstW, dstH;
    int dither, dither_off;  /* 8-bit target fed from a deeper source: ff_dither_8x8_128[y & 7][(x + dither_off) & 7] */
    FFHipDevFilter h, v;
    /* range conversion of the horizontal intermediates: 15-bit ones as in FFHipScalePlaneArgs, 19-bit ones (targets above 14 bits)
     * with 64-bit products and >> 18 (lumRangeToJpeg16_c & co, swscale.c:209-255); rc_coeff == 0: none */
    uint32_t rc_coeff;
    int rc_clip;
    int64_t rc_offset;
};
struct FFHipScale16Args {
    FFHipScale16Plane pl[3];
    int nplanes, max_rows;
    int sw_pitch; /* samples per staged source row in LDS (>= the widest 64-column reach, even) */
    int staged;   /* every 64-column run of every plane's horizontal bank reaches <= 320 source columns: footprints through LDS */
};
int ffhip_launch_scale16(const FFHipScale16Args &a, int nframes, hipStream_t stream);

/* per-line parity faces */
int ffhip_launch_hscale8to15(int16_t *dst, int dstW, ptrdiff_t dstPitch, const uint8_t *src, ptrdiff_t srcPitch,
                             int nlines, const int16_t *filter, const int32_t *pos, int fs, hipStream_t stream);
int ffhip_launch_yuv2nv12cX(int swap, const uint8_t *dither8, const int16_t *filter, int fs, const int16_t *usrc, const int16_t *vsrc,
                            ptrdiff_t srcPitch, uint8_t *dest, int chrDstW, hipStream_t stream);
/* mode 0 yuv2rgb_X, 1 _2, 2 _1; lines at base + j * pitch bytes */
int ffhip_launch_yuv2packed_line(int mode, const int16_t *lf, const int16_t *lum, int lfs, const int16_t *cf, const int16_t *cu,
                                 const int16_t *cv, int cfs, ptrdiff_t pitch, int yalpha, int uvalpha, uint8_t *dest, int dstW, int layout,
                                 const FFHipYuv2RgbK &k, hipStream_t stream);
int ffhip_launch_yuv2planeX8(const int16_t *filter, int fs, const int16_t *src, ptrdiff_t srcPitch, uint8_t *dest,
                             int dstW, const uint8_t *dither8, int offset, hipStream_t stream);

/* sws_rgbin.hip: a packed 8-bit RGB source's converter pass into 14-bit planar lines (4:2:2 at half chroma width, else 4:4:4) */
struct FFHipRgbInArgs {
    const uint8_t *src; ptrdiff_t src_stride; size_t src_fp;
    uint8_t *dst[3]; ptrdiff_t dst_stride[3]; size_t dst_fp[3];
    int w, h;                           /* pixels; rows of this call */
    int ro, go, bo;                     /* byte of the component inside a pixel */
    int ry, gy, by, ru, gu, bu, rv, gv, bv; /* input_rgb2yuv_table (swscale_internal.h:468-477) */
    uint8_t *y8; ptrdiff_t y8_stride; size_t y8_fp; /* non-null: the target's 8-bit luma plane is written instead of dst[0] (identity luma banks) */
    int c8;                             /* with y8: the chroma banks are the identity too (a 4:2:2 / 4:4:4 planar target at the source's size): dst[1], dst[2]
                                         * are the target's 8-bit chroma planes and nothing else runs */
};
/* k_sws_rgb420 (sws_rgbin.hip): a packed RGB source into yuv420p / NV12 at the source's size, fused; `in` as for k_sws_rgb_in with y8 = the luma plane */
struct FFHipRgb420Args {
    FFHipRgbInArgs in;
    uint8_t *cdst[2];            /* NV12: the interleaved plane; planar: U, V */
    ptrdiff_t cstride;
    size_t cfp;
    int chrH;                    /* chroma rows = h / 2 */
    const uint32_t *vfv;         /* device: the vertical chroma bank on the windows 2y - 3 .. 2y + 4, (chrH + 8) x 4 dwords */
    int nframes, ncb, nstrips, steps_per_strip;
};
int ffhip_launch_sws_rgb420(FFHipRgb420Args &A, int bpp, int nv, hipStream_t stream);
int ffhip_launch_sws_rgb_in(const FFHipRgbInArgs &a, int bpp, int half, int nframes, hipStream_t stream);
/* an 8-bit plane (wbytes bytes per row; an interleaved pair plane: both channels) as 16-bit samples: dst rows of 2 * wbytes bytes, 16-byte aligned */
int ffhip_launch_sws_widen8(const uint8_t *src, ptrdiff_t sstride, size_t sfp, uint8_t *dst, ptrdiff_t dstride, size_t dfp, int wbytes, int rows,
                            int nframes, hipStream_t stream);

#endif
