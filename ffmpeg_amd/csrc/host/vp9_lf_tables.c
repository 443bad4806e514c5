/*
 * vp9_lf_tables.c — host side of the VP9 loop filter in superblock order (plain C, no device): turns a superblock's VP9Filter
 * (libavcodec/vp9dec.h:79-83: level[64] + mask[2][2][8][4], built by the decoder in vp9block.c) into the tables
 * k_vp9_lf_frame walks.
 *
 * ff_vp9_loopfilter_sb() (libavcodec/vp9lpf.c:180-203) filters, per plane, all column edges of the superblock
 * (filter_plane_cols, :27-99) and then all row edges (filter_plane_rows, :101-178), choosing one dsp function per 16-line piece
 * of an edge from the mask bits.  Underneath, every such call is one or two 8-LINE SEGMENTS filtered with a width (4 / 8 / 16)
 * and the limits of a level — and inside one direction, segments on different lines never touch the same sample, while along
 * the filter axis they do (a 16-wide filter reaches 8 samples either way), so the only order that matters is the order of the
 * POSITIONS.  Hence the table: entry [dir][position p = 4 p samples along the filter axis][segment of 8 lines] = width + limits, or
 * nothing; a wave applies position after position with lane = line.  Which entry a call pattern produces is read off the
 * reference's branches (the "second half" rules are not symmetric: see the comments below).
 */
#include <string.h>

#include "ffhip_internal.h"

static uint32_t lf_entry(int wd, int L, const uint8_t *lim_lut, const uint8_t *mblim_lut)
{
    const uint32_t wd_idx = wd == 16 ? 2 : wd == 8 ? 1 : 0;
    return 0x80000000u | wd_idx << 24 | (uint32_t)(L >> 4) << 16 | (uint32_t)lim_lut[L] << 8 | mblim_lut[L];
}

/* filter_plane_cols: tab[p][seg], p < np positions, seg < 8 (luma) / 4 (4:2:0 chroma) */
static void lf_cols(uint32_t *tab, int nseg, int col, int ss_h, int ss_v, const uint8_t *lvl, const uint8_t (*mask)[4],
                    const uint8_t *lim_lut, const uint8_t *mblim_lut)
{
    for (int yi = 0; yi < (ss_v ? 2 : 4); yi++) {
        const int y = yi * (2 << ss_v);
        const uint8_t *h1 = mask[y], *h2 = mask[y + 1 + ss_v], *lrow = lvl + yi * (16 << ss_v);
        const unsigned hm1 = h1[0] | h1[1] | h1[2], hm13 = h1[3], hm2 = h2[1] | h2[2], hm23 = h2[3];
        const unsigned hm = hm1 | hm2 | hm13 | hm23;
        for (int xi = 0; xi < 8; xi++) {
            const unsigned x = 1u << xi;
            if (!(hm & ~(x - 1))) /* the reference's loop has ended */
                break;
            const uint8_t *l = lrow + (ss_h ? 2 * (xi >> 1) : xi);
            const int p = ss_h ? xi : 2 * xi;
            uint32_t *up = tab + p * nseg + 2 * yi, *lo = up + 1;
            if (col || xi) {
                if (hm1 & x) {
                    *up = lf_entry((h1[0] & x) ? 16 : (h1[1] & x) ? 8 : 4, l[0], lim_lut, mblim_lut);
                    if (h1[0] & x) {
                        /* loop_filter_16 when the lower half is 16 wide too — with the UPPER half's level; otherwise the lower
                         * half is not filtered at this position at all, whatever its own 8 / 4 bits say (vp9lpf.c:48-55) */
                        if (h2[0] & x)
                            *lo = lf_entry(16, l[0], lim_lut, mblim_lut);
                    } else if (hm2 & x) {
                        *lo = lf_entry((h2[1] & x) ? 8 : 4, l[8 << ss_v], lim_lut, mblim_lut);
                    }
                } else if (hm2 & x) {
                    *lo = lf_entry((h2[1] & x) ? 8 : 4, l[8 << ss_v], lim_lut, mblim_lut);
                }
            }
            if (!ss_h) { /* the inner edge of 4x4 transforms, 4 samples further */
                if (hm13 & x)
                    up[nseg] = lf_entry(4, l[0], lim_lut, mblim_lut);
                if (hm23 & x)
                    lo[nseg] = lf_entry(4, l[8 << ss_v], lim_lut, mblim_lut);
            }
        }
    }
}

/* filter_plane_rows: tab[p][seg], p = row position in units of 4 rows, seg = 8 columns */
static void lf_rows(uint32_t *tab, int nseg, int row, int ss_h, int ss_v, const uint8_t *lvl, const uint8_t (*mask)[4],
                    const uint8_t *lim_lut, const uint8_t *mblim_lut)
{
    for (int y = 0; y < 8; y++) {
        const uint8_t *vmask = mask[y], *lrow = ss_v ? lvl + 16 * (y >> 1) : lvl + 8 * y;
        const unsigned vm = vmask[0] | vmask[1] | vmask[2], vm3 = vmask[3];
        const int p = ss_v ? y : 2 * y;
        for (int k = 0; k < (ss_h ? 2 : 4); k++) {
            const unsigned x = 1u << (k * (2 << ss_h)), x2 = x << (1 + ss_h);
            if (!(vm & ~(x - 1))) /* the loop runs on vm alone: an inner edge beyond its last bit is never reached (vp9lpf.c:116) */
                break;
            const uint8_t *l = lrow + k * (2 << ss_h);
            uint32_t *first = tab + p * nseg + 2 * k, *second = first + 1;
            if (row || y) {
                if (vm & x) {
                    *first = lf_entry((vmask[0] & x) ? 16 : (vmask[1] & x) ? 8 : 4, l[0], lim_lut, mblim_lut);
                    if (vmask[0] & x) {
                        if (vmask[0] & x2)
                            *second = lf_entry(16, l[0], lim_lut, mblim_lut);
                    } else if (vm & x2) {
                        *second = lf_entry((vmask[1] & x2) ? 8 : 4, l[1 + ss_h], lim_lut, mblim_lut);
                    }
                } else if (vm & x2) {
                    *second = lf_entry((vmask[1] & x2) ? 8 : 4, l[1 + ss_h], lim_lut, mblim_lut);
                }
            }
            if (!ss_v) {
                if (vm3 & x)
                    first[nseg] = lf_entry(4, l[0], lim_lut, mblim_lut);
                if (vm3 & x2)
                    second[nseg] = lf_entry(4, l[1 + ss_h], lim_lut, mblim_lut);
            }
        }
    }
}

int ffhip_vp9_lf_sb_tables(FFHipVp9LfSb *out, const FFHipVp9Filter *lflvl, int row, int col, int ss_h, int ss_v, const uint8_t *lim_lut,
                           const uint8_t *mblim_lut)
{
    if (!out || !lflvl || !lim_lut || !mblim_lut || ((ss_h | ss_v) & ~1)) {
        ffhip_set_error("ffhip_vp9_lf_sb_tables: null argument, or a sub-sampling shift other than 0 / 1");
        return FFHIP_EINVAL;
    }
    memset(out, 0, sizeof(*out));
    lf_cols(&out->y[0][0][0], 8, col, 0, 0, lflvl->level, lflvl->mask[0][0], lim_lut, mblim_lut);
    lf_rows(&out->y[1][0][0], 8, row, 0, 0, lflvl->level, lflvl->mask[0][1], lim_lut, mblim_lut);
    if (!ss_h || !ss_v)
        return 0; /* 4:4:4: the chroma planes take the luma tables (ffhip_vp9_loopfilter_frame_ss_dev); 4:2:2 / 4:4:0: tables of their own
                   * (ffhip_vp9_lf_sb_ctables); uv stays empty */
    lf_cols(&out->uv[0][0][0], 4, col, ss_h, ss_v, lflvl->level, lflvl->mask[1][0], lim_lut, mblim_lut);
    lf_rows(&out->uv[1][0][0], 4, row, ss_h, ss_v, lflvl->level, lflvl->mask[1][1], lim_lut, mblim_lut);
    /* a 16-wide chroma filter on the superblock's last 4-sample position would reach 4 samples into the next superblock: mask_edges
     * never asks for it (16-wide chroma edges sit on multiples of 16 samples, vp9block.c:1215-1228), and the kernel's tile ends
     * with the superblock */
    for (int d = 0; d < 2; d++)
        for (int sg = 0; sg < 4; sg++)
            if ((out->uv[d][7][sg] >> 31) && ((out->uv[d][7][sg] >> 24) & 3) == 2) {
                ffhip_set_error("ffhip_vp9_lf_sb_tables: a 16-wide chroma filter 4 samples before the superblock's end (no stream produces it)");
                return FFHIP_EINVAL;
            }
    return 0;
}

/* The chroma planes of a 4:2:2 (ss_h 1, ss_v 0) or 4:4:0 (0, 1) superblock: filter_plane_cols / _rows with the two shifts apart
 * (vp9lpf.c:185-201, uv_masks = lflvl->mask[ss_h | ss_v] = mask[1]) — lf_cols / lf_rows above already follow the reference's branches for
 * either shift; only the tile is rectangular: column edges [position][segment] = 8 x 8 (4:2:2) / 16 x 4 (4:4:0), row edges 16 x 4 / 8 x 8. */
int ffhip_vp9_lf_sb_ctables(FFHipVp9LfSbC *out, const FFHipVp9Filter *lflvl, int row, int col, int ss_h, int ss_v, const uint8_t *lim_lut,
                            const uint8_t *mblim_lut)
{
    if (!out || !lflvl || !lim_lut || !mblim_lut || ((ss_h | ss_v) & ~1) || ss_h == ss_v) {
        ffhip_set_error("ffhip_vp9_lf_sb_ctables: null argument, or a chroma format other than 4:2:2 (1, 0) / 4:4:0 (0, 1)");
        return FFHIP_EINVAL;
    }
    const int npc = ss_h ? 8 : 16, nsc = ss_v ? 4 : 8, npr = ss_v ? 8 : 16, nsr = ss_h ? 4 : 8;
    memset(out, 0, sizeof(*out));
    lf_cols(out->t, nsc, col, ss_h, ss_v, lflvl->level, lflvl->mask[1][0], lim_lut, mblim_lut);
    lf_rows(out->t + npc * nsc, nsr, row, ss_h, ss_v, lflvl->level, lflvl->mask[1][1], lim_lut, mblim_lut);
    /* a 16-wide filter on a tile's last 4-sample position would reach into the next superblock (never asked for: vp9block.c:1215-1228) */
    for (int sg = 0; sg < nsc; sg++)
        if ((out->t[(npc - 1) * nsc + sg] >> 31) && ((out->t[(npc - 1) * nsc + sg] >> 24) & 3) == 2)
            goto wide;
    for (int sg = 0; sg < nsr; sg++)
        if ((out->t[npc * nsc + (npr - 1) * nsr + sg] >> 31) && ((out->t[npc * nsc + (npr - 1) * nsr + sg] >> 24) & 3) == 2)
            goto wide;
    return 0;
wide:
    ffhip_set_error("ffhip_vp9_lf_sb_ctables: a 16-wide chroma filter 4 samples before the superblock's end (no stream produces it)");
    return FFHIP_EINVAL;
}
