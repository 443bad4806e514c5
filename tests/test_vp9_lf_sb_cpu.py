"""The VP9 loop filter in the decoder's order, CPU side:
 * the oracle's ffo_vp9_loopfilter_sb == the reference's own ff_vp9_loopfilter_sb (libavcodec/vp9lpf.c:180-203, compiled in place
   under oracle/_ref), at 8, 10 and 12 bits, 4:2:0 and 4:4:4, first and later rows / columns;
 * the product's host converter ffhip_vp9_lf_sb_tables (ffmpeg_amd/csrc/host/vp9_lf_tables.c) — executed the way the kernel
   executes it (vp9_lf_gen.run_tables) — leaves the same samples as the oracle's walk of the masks."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p
import vp9_lf_gen as G


def _planes(rng, bd, smooth):
    """a 3x3-superblock neighbourhood per plane (the filters reach 8 samples out of the middle one); smooth content so that the flat
    and the 4-tap branches all trigger"""
    dt = np.uint8 if bd == 8 else np.uint16
    out = []
    for n in (192, 96, 96):
        if smooth:
            base = np.cumsum(rng.integers(-2, 3, (n, n + 7)), axis=1) + rng.integers(0, 1 << bd)
            base = (base << (bd - 8)) // 4 + rng.integers(0, 1 << (bd - 8) + 1, (n, n + 7)) + (1 << (bd - 1))
        else:
            base = rng.integers(0, 1 << bd, (n, n + 7))
        out.append(np.clip(base, 0, (1 << bd) - 1).astype(dt))
    return out


def _at(a, r, c):
    return a.ctypes.data + r * a.strides[0] + c * a.itemsize


CASES = [(bd, ss, kind, seed) for bd in (8, 10, 12) for ss in (1, 0) for kind in ("structured", "bits0", "bits1", "bits2") for seed in range(3)]


@pytest.mark.parametrize("bd,ss,kind,seed", CASES)
def test_oracle_vs_reference(bd, ss, kind, seed):
    R = ffi.ref()
    if R is None or not hasattr(R, "ffref_vp9_loopfilter_sb"):
        pytest.skip("oracle/_ref not built")
    O = ffi.oracle()
    rng = np.random.default_rng(hash((bd, ss, kind, seed)) & 0xFFFFFF)
    lim, mblim = G.filter_lut(int(rng.integers(0, 8)))
    for row, col in ((0, 0), (0, 8), (8, 0), (8, 8)):
        f = G.structured(rng, row // 8, col // 8, 24 - int(rng.integers(0, 3)), 24 - int(rng.integers(0, 3)), ss, ss) if kind == "structured" \
            else G.random_bits(rng, int(kind[-1]))
        a = _planes(rng, bd, seed != 2)
        if not ss:
            a[1], a[2] = _planes(rng, bd, True)[0], _planes(rng, bd, True)[0]
        b = [p.copy() for p in a]
        m = 64 if not ss else 32
        level, mask = np.ascontiguousarray(f["level"]), np.ascontiguousarray(f["mask"])
        args = (bd, ss, ss, ptr(level, u8p), ptr(mask, u8p), row, col)
        O.ffo_vp9_loopfilter_sb(*args, *(C.cast(_at(p, k, k), u8p) for p, k in zip(a, (64, m, m))), a[0].strides[0], a[1].strides[0],
                                ptr(lim, u8p), ptr(mblim, u8p))
        R.ffref_vp9_loopfilter_sb(*args, *(C.cast(_at(p, k, k), u8p) for p, k in zip(b, (64, m, m))), b[0].strides[0], b[1].strides[0],
                                  ptr(lim, u8p), ptr(mblim, u8p))
        for p, q in zip(a, b):
            assert np.array_equal(p, q)


@pytest.mark.parametrize("bd,kind,seed", [(bd, kind, seed) for bd in (8, 10) for kind in ("structured", "bits0", "bits1", "bits2") for seed in range(4)])
def test_tables_vs_oracle(bd, kind, seed):
    from ffmpeg_amd import _lib
    L = _lib.lib()
    O = ffi.oracle()
    rng = np.random.default_rng(hash((bd, kind, seed, 7)) & 0xFFFFFF)
    lim, mblim = G.filter_lut(int(rng.integers(0, 8)))
    changed = 0
    for row, col in ((0, 0), (0, 8), (8, 0), (8, 8), (16, 16)):
        f = G.structured(rng, row // 8, col // 8, 24 - int(rng.integers(0, 3)), 24 - int(rng.integers(0, 3))) if kind == "structured" \
            else G.random_bits(rng, int(kind[-1]))
        a = _planes(rng, bd, seed != 3)
        b = [p.copy() for p in a]
        c = [p.copy() for p in a]
        level, mask = np.ascontiguousarray(f["level"]), np.ascontiguousarray(f["mask"])
        O.ffo_vp9_loopfilter_sb(bd, 1, 1, ptr(level, u8p), ptr(mask, u8p), row, col, *(C.cast(_at(p, k, k), u8p) for p, k in zip(a, (64, 32, 32))),
                                a[0].strides[0], a[1].strides[0], ptr(lim, u8p), ptr(mblim, u8p))
        tab = np.zeros(G.TABLE_WORDS, np.uint32)
        fb = np.frombuffer(f.tobytes(), np.uint8).copy()
        assert L.ffhip_vp9_lf_sb_tables(tab.ctypes.data, fb.ctypes.data, row, col, 1, 1, lim.ctypes.data, mblim.ctypes.data) == 0
        G.run_tables(O, tab, bd, [_at(p, k, k) for p, k in zip(b, (64, 32, 32))], [b[0].strides[0], b[1].strides[0]])
        for p, q in zip(a, b):
            assert np.array_equal(p, q)
        changed += sum(int((p != q).sum()) for p, q in zip(a, c))
    assert changed > 200 or seed == 3                         # noise rarely passes the filter mask


def test_tables_reject():
    from ffmpeg_amd import _lib
    L = _lib.lib()
    tab = np.zeros(G.TABLE_WORDS, np.uint32)
    f = np.zeros(192, np.uint8)
    lim, mblim = G.filter_lut(0)
    assert L.ffhip_vp9_lf_sb_tables(tab.ctypes.data, f.ctypes.data, 0, 0, 0, 0, lim.ctypes.data, mblim.ctypes.data) == 0  # 4:4:4: the luma tables
    assert L.ffhip_vp9_lf_sb_tables(tab.ctypes.data, f.ctypes.data, 0, 0, 1, 0, lim.ctypes.data, mblim.ctypes.data) == 0  # 4:2:2 / 4:4:0: the luma part
    assert L.ffhip_vp9_lf_sb_tables(tab.ctypes.data, f.ctypes.data, 0, 0, 2, 0, lim.ctypes.data, mblim.ctypes.data) < 0
    ctab = np.zeros(128, np.uint32)
    assert L.ffhip_vp9_lf_sb_ctables(ctab.ctypes.data, f.ctypes.data, 0, 0, 1, 0, lim.ctypes.data, mblim.ctypes.data) == 0
    assert L.ffhip_vp9_lf_sb_ctables(ctab.ctypes.data, f.ctypes.data, 0, 0, 1, 1, lim.ctypes.data, mblim.ctypes.data) < 0  # 4:2:0 / 4:4:4: the other tables
    assert L.ffhip_vp9_lf_sb_tables(None, f.ctypes.data, 0, 0, 1, 1, lim.ctypes.data, mblim.ctypes.data) < 0
    g = np.zeros((), G.FILTER_DT)
    g["mask"][1, 0, 0, 0] = 0x80                               # 16-wide chroma column edge at the superblock's last position
    assert L.ffhip_vp9_lf_sb_tables(tab.ctypes.data, np.frombuffer(g.tobytes(), np.uint8).copy().ctypes.data, 0, 0, 1, 1, lim.ctypes.data,
                                    mblim.ctypes.data) < 0


@pytest.mark.parametrize("bd,ss,kind,seed", [(bd, ss, kind, seed) for bd in (8, 10) for ss in ((1, 0), (0, 1)) for kind in ("structured", "bits0", "bits1", "bits2")
                                             for seed in range(3)])
def test_422_440_oracle_reference_and_ctables(bd, ss, kind, seed):
    """the two sub-sampling shifts apart (VP9 4:2:2: ss_h 1, ss_v 0; 4:4:0: 0, 1): the oracle's ffo_vp9_loopfilter_sb == the reference's
    ff_vp9_loopfilter_sb (filter_plane_cols / _rows, vp9lpf.c:27-178, with uv_masks = mask[1]), and the product's chroma table
    (ffhip_vp9_lf_sb_ctables: 32 x 64 / 64 x 32 chroma superblocks) executed in the kernel's order leaves the same chroma samples; luma by
    the y part of ffhip_vp9_lf_sb_tables"""
    from ffmpeg_amd import _lib
    L = _lib.lib()
    R = ffi.ref()
    if R is None or not hasattr(R, "ffref_vp9_loopfilter_sb"):
        pytest.skip("oracle/_ref not built")
    O = ffi.oracle()
    ss_h, ss_v = ss
    rng = np.random.default_rng(hash((bd, ss, kind, seed, 11)) & 0xFFFFFF)
    lim, mblim = G.filter_lut(int(rng.integers(0, 8)))
    dt = np.uint8 if bd == 8 else np.uint16
    cw, chh = 64 >> ss_h, 64 >> ss_v
    changed = 0
    for row, col in ((0, 0), (0, 8), (8, 0), (8, 8)):
        f = G.structured(rng, row // 8, col // 8, 24 - int(rng.integers(0, 3)), 24 - int(rng.integers(0, 3)), ss_h, ss_v) if kind == "structured" \
            else G.random_bits(rng, int(kind[-1]))
        y = _planes(rng, bd, seed != 2)[0]
        uv = [np.clip(np.cumsum(rng.integers(-2, 3, (3 * chh, 3 * cw + 7)), axis=1) * (1 << (bd - 8)) + (1 << (bd - 1)), 0, (1 << bd) - 1).astype(dt)
              for _ in range(2)]
        a, b, c = [y] + uv, [y.copy()] + [p.copy() for p in uv], [y.copy()] + [p.copy() for p in uv]
        before = [p.copy() for p in a]
        level, mask = np.ascontiguousarray(f["level"]), np.ascontiguousarray(f["mask"])
        args = (bd, ss_h, ss_v, ptr(level, u8p), ptr(mask, u8p), row, col)
        pos = ((64, 64), (chh, cw), (chh, cw))
        O.ffo_vp9_loopfilter_sb(*args, *(C.cast(_at(p, r, k), u8p) for p, (r, k) in zip(a, pos)), a[0].strides[0], a[1].strides[0], ptr(lim, u8p),
                                ptr(mblim, u8p))
        R.ffref_vp9_loopfilter_sb(*args, *(C.cast(_at(p, r, k), u8p) for p, (r, k) in zip(b, pos)), b[0].strides[0], b[1].strides[0], ptr(lim, u8p),
                                  ptr(mblim, u8p))
        for p, q in zip(a, b):
            assert np.array_equal(p, q)
        tab, ctab = np.zeros(320, np.uint32), np.zeros(128, np.uint32)
        rt = L.ffhip_vp9_lf_sb_tables(tab.ctypes.data, f.ctypes.data if hasattr(f, "ctypes") else np.ascontiguousarray(f).ctypes.data, row, col, ss_h,
                                      ss_v, lim.ctypes.data, mblim.ctypes.data)
        rc = L.ffhip_vp9_lf_sb_ctables(ctab.ctypes.data, np.ascontiguousarray(f).ctypes.data, row, col, ss_h, ss_v, lim.ctypes.data, mblim.ctypes.data)
        if rc != 0 and kind != "structured":
            continue                              # arbitrary bits may ask for a 16-wide filter at a tile's last position: refused by name
        assert rt == 0 and rc == 0 and not tab[256:].any()
        G.run_tables(O, np.concatenate([tab[:256], np.zeros(64, np.uint32)]), bd, [_at(c[0], 64, 64), _at(c[1], chh, cw), _at(c[2], chh, cw)],
                     [c[0].strides[0], c[1].strides[0]])
        G.run_ctables(O, ctab, bd, [_at(c[1], chh, cw), _at(c[2], chh, cw)], c[1].strides[0], ss_h, ss_v)
        for p, q in zip(a, c):
            assert np.array_equal(p, q)
        changed += sum(int((p != q).sum()) for p, q in zip(a[1:], before[1:]))
    assert changed > 0 or kind != "structured"
