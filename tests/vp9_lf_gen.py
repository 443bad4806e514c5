"""Synthetic VP9Filter state (libavcodec/vp9dec.h:79-83) for the superblock-order loop-filter tests, with the frame's filter_lut
(libavcodec/vp9.c:683-697).  Two generators: `structured` builds level[] and mask[] from a random block / transform partition by
the decoder's rules (the tail of ff_vp9_decode_block, libavcodec/vp9block.c:1433-1447, and mask_edges, :1141-1262, restated) —
what a stream produces, picture edges included; `random_bits` sets arbitrary bits — every branch of ff_vp9_loopfilter_sb,
including combinations no stream produces.  Test infrastructure: any mask is a valid input of the function under test."""
import numpy as np

FILTER_DT = np.dtype([("level", "u1", (64,)), ("mask", "u1", (2, 2, 8, 4))])
assert FILTER_DT.itemsize == 64 + 128
TABLE_WORDS = 320


def filter_lut(sharp):
    lim, mblim = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
    for i in range(1, 64):
        limit = i
        if sharp > 0:
            limit >>= (sharp + 3) >> 2
            limit = min(limit, 9 - sharp)
        limit = max(limit, 1)
        lim[i], mblim[i] = limit, 2 * (i + 2) + limit
    return lim, mblim


def random_bits(rng, density):
    f = np.zeros((), FILTER_DT)
    f["level"] = rng.integers(0, 64, 64)
    m = rng.integers(0, 256, (2, 2, 8, 4))
    for _ in range(int(density)):
        m &= rng.integers(0, 256, (2, 2, 8, 4))
    m[1, 0, :, 0] &= 0x7F            # no 16-wide chroma filter on the superblock's last position (it would leave the superblock;
    m[1, 1, 7, 0] = 0                # mask_edges never sets it, ffhip_vp9_lf_sb_tables rejects it)
    f["mask"] = m
    return f


def mask_edges(mask, ss_h, ss_v, row_and_7, col_and_7, w, h, col_end, row_end, tx, skip_inter):
    wide_col, wide_row = (0x11, 0x01), (0x03, 0x07)
    if tx == 0 and (ss_v | ss_h):
        if h == ss_v:
            if row_and_7 & 1:
                return
            if not row_end:
                h += 1
        if w == ss_h:
            if col_and_7 & 1:
                return
            if not col_end:
                w += 1
    t = 1 << col_and_7
    m_col = (t << w) - t
    ys = range(row_and_7, h + row_and_7)
    if tx == 0 and not skip_inter:
        m_row_8 = m_col & wide_col[ss_h]
        m_row_4 = m_col - m_row_8
        for y in ys:
            cid = 2 - (0 if (y & wide_row[ss_v]) else 1)
            mask[0][y][1] |= m_row_8
            mask[0][y][2] |= m_row_4
            if (ss_h & ss_v) and (col_end & 1) and (y & 1):
                mask[1][y][cid] |= (t << (w - 1)) - t
            else:
                mask[1][y][cid] |= m_col
            if not ss_h:
                mask[0][y][3] |= m_col
            if not ss_v:
                mask[1][y][3] |= ((t << (w - 1)) - t) if (ss_h and (col_end & 1)) else m_col
    elif not skip_inter:
        masks = (0xff, 0x55, 0x11, 0x01)
        mask_id = int(tx == 1)
        l2 = tx + ss_h - 1
        m_row = m_col & masks[l2]
        if ss_h and tx > 1 and (w ^ (w - 1)) == 1:
            m_row_16 = ((t << (w - 1)) - t) & masks[l2]
            for y in ys:
                mask[0][y][0] |= m_row_16
                mask[0][y][1] |= m_row - m_row_16
        else:
            for y in ys:
                mask[0][y][mask_id] |= m_row
        l2 = tx + ss_v - 1
        step = 1 << l2
        if ss_v and tx > 1 and (h ^ (h - 1)) == 1:
            y = row_and_7
            while y < h + row_and_7 - 1:
                mask[1][y][0] |= m_col
                y += step
            if y - row_and_7 == h - 1:
                mask[1][y][1] |= m_col
        else:
            for y in range(row_and_7, h + row_and_7, step):
                mask[1][y][mask_id] |= m_col
    elif tx != 0:
        mask[1][row_and_7][int(tx == 1 or h == ss_v)] |= m_col
        mid = int(tx == 1 or w == ss_h)
        for y in ys:
            mask[0][y][mid] |= t
    else:
        t8 = t & wide_col[ss_h]
        for y in ys:
            mask[0][y][2] |= t - t8
            mask[0][y][1] |= t8
        mask[1][row_and_7][2 - (0 if (row_and_7 & wide_row[ss_v]) else 1)] |= m_col


def structured(rng, sb_row, sb_col, cols, rows, ss_h=1, ss_v=1, p_zero=.1):
    """VP9Filter of the superblock at (sb_row, sb_col) of a picture of cols x rows 8x8 blocks"""
    lvl_tab = np.zeros((8, 8), np.int64)
    mask = np.zeros((2, 2, 8, 4), np.int64)

    def block(r7, c7, w4, h4, sub8):
        row, col = sb_row * 8 + r7, sb_col * 8 + c7
        if row >= rows or col >= cols:
            return
        lvl = 0 if rng.random() < p_zero else int(rng.integers(1, 64))
        if lvl == 0:
            return
        max_tx = 0 if sub8 else min(3, int(np.log2(min(w4, h4))) + 1)
        tx = int(rng.integers(0, max_tx + 1))
        uvtx = tx - int((ss_h and w4 * 2 == (1 << tx)) or (ss_v and h4 * 2 == (1 << tx)))
        skip_inter = int(rng.random() < .3)
        x_end, y_end = min(cols - col, w4), min(rows - row, h4)
        lvl_tab[r7:r7 + h4, c7:c7 + w4] = lvl                    # setctx_2d writes the full block, clipped only by the table
        mask_edges(mask[0], 0, 0, r7, c7, x_end, y_end, 0, 0, tx, skip_inter)
        mask_edges(mask[1], ss_h, ss_v, r7, c7, x_end, y_end, (cols & 7) if (cols & 1 and col + w4 >= cols) else 0,
                   (rows & 7) if (rows & 1 and row + h4 >= rows) else 0, uvtx, skip_inter)

    def part(r7, c7, n):
        k = rng.random()
        if n > 1 and k < .55:
            h = n // 2
            for dr, dc in ((0, 0), (0, h), (h, 0), (h, h)):
                part(r7 + dr, c7 + dc, h)
        elif n > 1 and k < .7:
            block(r7, c7, n, n // 2, False)
            block(r7 + n // 2, c7, n, n // 2, False)
        elif n > 1 and k < .85:
            block(r7, c7, n // 2, n, False)
            block(r7, c7 + n // 2, n // 2, n, False)
        else:
            block(r7, c7, n, n, n == 1 and rng.random() < .5)
    part(0, 0, 8)
    f = np.zeros((), FILTER_DT)
    f["level"] = lvl_tab.reshape(64)
    f["mask"] = mask & 0xFF
    return f


def run_tables(O, tab, bd, planes, strides, ss_h=1, ss_v=1):
    """executes one superblock's FFHipVp9LfSb the way the kernel does — all column edges, then all row edges, position by position,
    segment by segment — with the oracle's per-edge filter; planes = (y, u, v) byte addresses of the superblock's first sample"""
    import ctypes as C
    u8 = C.POINTER(C.c_uint8)
    ps = 2 if bd > 8 else 1
    t = np.asarray(tab, np.uint32).reshape(TABLE_WORDS)
    for pl in range(3):
        npos, nseg, base = (16, 8, 0) if pl == 0 else (8, 4, 256)
        for d in (0, 1):
            for p in range(npos):
                for sg in range(nseg):
                    e = int(t[base + (d * npos + p) * nseg + sg])
                    if not e >> 31:
                        continue
                    wd = (4, 8, 16)[(e >> 24) & 3]
                    st = strides[0 if pl == 0 else 1]
                    at = planes[pl] + ((8 * sg * st + 4 * p * ps) if d == 0 else (4 * p * st + 8 * sg * ps))
                    O.ffo_vp9_loop_filter_bd(bd, wd, d, C.cast(at, u8), st, e & 0xFF, (e >> 8) & 0xFF, (e >> 16) & 0xFF)


def run_ctables(O, ctab, bd, planes_uv, stride, ss_h, ss_v):
    """executes one superblock's FFHipVp9LfSbC (4:2:2 / 4:4:0) the way k_vp9_lf_frame_ssc does, on both chroma planes: all column edges
    position by position, then all row edges"""
    import ctypes as C
    u8 = C.POINTER(C.c_uint8)
    ps = 2 if bd > 8 else 1
    t = np.asarray(ctab, np.uint32).reshape(128)
    npc, nsc, npr, nsr = (8 if ss_h else 16), (4 if ss_v else 8), (8 if ss_v else 16), (4 if ss_h else 8)
    for base_at in planes_uv:
        for d, (npos, nseg, off) in enumerate(((npc, nsc, 0), (npr, nsr, npc * nsc))):
            for p in range(npos):
                for sg in range(nseg):
                    e = int(t[off + p * nseg + sg])
                    if not e >> 31:
                        continue
                    wd = (4, 8, 16)[(e >> 24) & 3]
                    at = base_at + ((8 * sg * stride + 4 * p * ps) if d == 0 else (4 * p * stride + 8 * sg * ps))
                    O.ffo_vp9_loop_filter_bd(bd, wd, d, C.cast(at, u8), stride, e & 0xFF, (e >> 8) & 0xFF, (e >> 16) & 0xFF)
