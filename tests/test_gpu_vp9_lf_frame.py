"""GPU parity of the VP9 loop filter in the decoder's order: ffhip_vp9_loopfilter_frame_dev — one launch for a
picture, superblocks as a wavefront — vs the oracle's ffo_vp9_loopfilter_sb called superblock by superblock in raster order
(pinned to the reference's ff_vp9_loopfilter_sb in tests/test_vp9_lf_sb_cpu.py).  Bit-exact at 8, 10 and 12 bits."""
import ctypes as C

import numpy as np
import pytest

import ffi
from ffi import ptr, u8p
import vp9_lf_gen as G

pytestmark = pytest.mark.gpu


def _plane(rng, h, w, pad, bd):
    base = np.cumsum(rng.integers(-2, 3, (h, w + pad)), axis=1) + np.cumsum(rng.integers(-2, 3, (h, 1)), axis=0) + (1 << 7)
    base = (base << (bd - 8)) + rng.integers(0, (1 << (bd - 8)) + 1, (h, w + pad))
    base[rng.integers(0, h, 40), :] += 9 << (bd - 8)                                 # a few real edges
    return np.clip(base, 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("sbc,sbr,kind", [(9, 5, "structured"), (9, 5, "bits0"), (7, 6, "bits1"), (1, 1, "structured"), (1, 7, "bits1"), (12, 1, "structured"),
                                          (2, 40, "bits2"), (30, 17, "structured")])
def test_vp9_loopfilter_frame(sbc, sbr, kind, bd):
    import torch
    from ffmpeg_amd import vp9
    assert torch.cuda.is_available()
    rng = np.random.default_rng(1000 * sbc + 10 * sbr + bd + len(kind))
    lim, mblim = G.filter_lut(int(rng.integers(0, 8)))
    cols, rows = 8 * sbc, 8 * sbr
    if kind == "structured":                    # arbitrary mask bits know no picture edge; the decoder's masks end with the picture
        cols, rows = cols - int(rng.integers(0, 8)), rows - int(rng.integers(0, 8))
    planes = [_plane(rng, 64 * sbr, 64 * sbc, 12, bd), _plane(rng, 32 * sbr, 32 * sbc, 4, bd), _plane(rng, 32 * sbr, 32 * sbc, 4, bd)]
    before = [p.copy() for p in planes]
    filt = np.zeros(sbr * sbc, G.FILTER_DT)
    O = ffi.oracle()
    for r in range(sbr):
        for c in range(sbc):
            f = G.structured(rng, r, c, cols, rows) if kind == "structured" else G.random_bits(rng, int(kind[-1]))
            filt[r * sbc + c] = f
            level, mask = np.ascontiguousarray(f["level"]), np.ascontiguousarray(f["mask"])
            at = [p.ctypes.data + r * (64 >> (k > 0)) * p.strides[0] + c * (64 >> (k > 0)) * p.itemsize for k, p in enumerate(planes)]
            O.ffo_vp9_loopfilter_sb(bd, 1, 1, ptr(level, u8p), ptr(mask, u8p), 8 * r, 8 * c, *(C.cast(a, u8p) for a in at),
                                    planes[0].strides[0], planes[1].strides[0], ptr(lim, u8p), ptr(mblim, u8p))
    tabs = vp9.lf_sb_tables(filt.view(np.uint8).reshape(sbr * sbc, 192), sbc, sbr, lim, mblim)
    dev = [torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).cuda() for b in before]
    d_tabs = torch.from_numpy(tabs.view(np.int32)).cuda()
    vp9.loopfilter_frame(dev[0], dev[1], dev[2], before[0].strides[0], before[1].strides[0], cols, rows, d_tabs, bit_depth=bd)
    torch.cuda.synchronize()
    from ffmpeg_amd import _lib
    assert _lib.lib().ffhip_stream_synchronize(None) == 0
    changed = 0
    for k, (d, want, b) in enumerate(zip(dev, planes, before)):
        got = d.cpu().numpy().view(want.dtype).reshape(want.shape)
        h, w = (8 * rows, 8 * cols) if k == 0 else (4 * rows, 4 * cols)       # the picture proper: cols x rows 8x8 blocks
        bad = np.argwhere(got[:h, :w] != want[:h, :w])
        assert bad.size == 0, (k, bad[:5], len(bad))
        outside = got != b                                                      # beyond it nothing is written (a decoder's frame
        outside[:h, :w] = False                                                 # buffer ends there, give or take its alignment)
        assert not outside.any()
        changed += int((want != b).sum())
    assert changed > (20 * sbc * sbr if sbc * sbr > 8 else -1)


def test_vp9_loopfilter_frame_rejects():
    import torch
    from ffmpeg_amd import vp9
    y = torch.zeros(64 * 64 + 8, dtype=torch.uint8, device="cuda")
    t = torch.zeros(320, dtype=torch.int32, device="cuda")
    with pytest.raises(Exception):
        vp9.loopfilter_frame(y, y, y, 64, 32, 8, 8, t, bit_depth=9)
    with pytest.raises(Exception):
        vp9.loopfilter_frame(y[1:], y, y, 64, 32, 8, 8, t)          # misaligned plane


@pytest.mark.parametrize("knobs", [{"FFHIP_VP9_LF_WPB": "1"}, {"FFHIP_VP9_LF_WPB": "2"}, {"FFHIP_VP9_LF_WPB": "3"}, {"FFHIP_VP9_LF_OLD": "1"}],
                         ids=["1-row-workgroups", "2-row-workgroups", "3-row-workgroups", "row-kernel"])
def test_vp9_loopfilter_frame_workgroup_shapes(knobs, monkeypatch, measure_build):
    """1 .. 3 superblock rows per workgroup (hand-offs through LDS inside a workgroup, through memory between workgroups) and the
    one-wave-per-row kernel of round 2 == the serial order, 8 and 10 bits"""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    test_vp9_loopfilter_frame(9, 5, "structured", 8)
    test_vp9_loopfilter_frame(2, 40, "bits2", 8)
    test_vp9_loopfilter_frame(7, 6, "bits1", 10)


def test_vp9_loopfilter_frame_lost_handoff_is_reported(monkeypatch, measure_build):
    """a wave that never receives a hand-off (FFHIP_VP9_LF_FAULT=1: nothing is published, neither the LDS counters of a workgroup nor
    the counters in memory) times out and the launch is REPORTED at the next synchronisation point"""
    import torch
    from ffmpeg_amd import vp9, _lib
    L = _lib.lib()
    assert L.ffhip_stream_synchronize(None) == 0
    sbc, sbr = 2, 9
    y = torch.zeros(64 * sbr * 64 * sbc, dtype=torch.uint8, device="cuda")
    u = torch.zeros(32 * sbr * 32 * sbc, dtype=torch.uint8, device="cuda")
    t = torch.zeros(sbr * sbc * 320, dtype=torch.int32, device="cuda")
    monkeypatch.setenv("FFHIP_VP9_LF_FAULT", "1")
    vp9.loopfilter_frame(y, u, u.clone(), 64 * sbc, 32 * sbc, 8 * sbc, 8 * sbr, t)
    monkeypatch.delenv("FFHIP_VP9_LF_FAULT")
    assert L.ffhip_stream_synchronize(None) == -5                       # FFHIP_EIO
    assert L.ffhip_stream_synchronize(None) == 0                       # reported once
    test_vp9_loopfilter_frame(9, 5, "structured", 8)                      # and the pool keeps working


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("sbc,sbr,kind", [(9, 5, "structured"), (7, 6, "bits1"), (1, 1, "structured"), (2, 13, "bits2")])
def test_vp9_loopfilter_frame_444(sbc, sbr, kind, bd):
    """4:4:4 (ss_h = ss_v = 0): ff_vp9_loopfilter_sb filters the chroma planes with luma's masks and levels (vp9lpf.c:185-201) — the
    frame call with ss = (0, 0) == the oracle's ffo_vp9_loopfilter_sb(ss_h 0, ss_v 0) superblock by superblock, all three planes"""
    import torch
    from ffmpeg_amd import vp9, _lib
    rng = np.random.default_rng(2000 * sbc + 10 * sbr + bd + len(kind))
    lim, mblim = G.filter_lut(int(rng.integers(0, 8)))
    cols, rows = 8 * sbc, 8 * sbr
    if kind == "structured":
        cols, rows = cols - int(rng.integers(0, 8)), rows - int(rng.integers(0, 8))
    planes = [_plane(rng, 64 * sbr, 64 * sbc, 12 if k == 0 else 4, bd) for k in range(3)]
    before = [p.copy() for p in planes]
    filt = np.zeros(sbr * sbc, G.FILTER_DT)
    O = ffi.oracle()
    for r in range(sbr):
        for c in range(sbc):
            f = G.structured(rng, r, c, cols, rows) if kind == "structured" else G.random_bits(rng, int(kind[-1]))
            filt[r * sbc + c] = f
            level, mask = np.ascontiguousarray(f["level"]), np.ascontiguousarray(f["mask"])
            at = [p.ctypes.data + r * 64 * p.strides[0] + c * 64 * p.itemsize for p in planes]
            O.ffo_vp9_loopfilter_sb(bd, 0, 0, ptr(level, u8p), ptr(mask, u8p), 8 * r, 8 * c, *(C.cast(a, u8p) for a in at),
                                    planes[0].strides[0], planes[1].strides[0], ptr(lim, u8p), ptr(mblim, u8p))
    tabs = vp9.lf_sb_tables(filt.view(np.uint8).reshape(sbr * sbc, 192), sbc, sbr, lim, mblim)
    dev = [torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).cuda() for b in before]
    d_tabs = torch.from_numpy(tabs.view(np.int32)).cuda()
    vp9.loopfilter_frame(dev[0], dev[1], dev[2], before[0].strides[0], before[1].strides[0], cols, rows, d_tabs, bit_depth=bd, ss=(0, 0))
    torch.cuda.synchronize()
    assert _lib.lib().ffhip_stream_synchronize(None) == 0
    changed = 0
    for k, (d, want, b) in enumerate(zip(dev, planes, before)):
        got = d.cpu().numpy().view(want.dtype).reshape(want.shape)
        h, w = 8 * rows, 8 * cols
        bad = np.argwhere(got[:h, :w] != want[:h, :w])
        assert bad.size == 0, (k, bad[:5], len(bad))
        outside = got != b
        outside[:h, :w] = False
        assert not outside.any()
        changed += int((want != b).sum())
    assert changed > (30 * sbc * sbr if sbc * sbr > 8 else -1)
    with pytest.raises(Exception):
        vp9.loopfilter_frame(dev[0], dev[1], dev[2], before[0].strides[0], before[1].strides[0], cols, rows, d_tabs, bit_depth=bd, ss=(1, 0))


@pytest.mark.parametrize("bd,ss,sbc,sbr,npics", [(8, (1, 1), 9, 5, 6), (10, (1, 1), 7, 6, 3), (8, (0, 0), 5, 4, 4), (8, (1, 1), 3, 2, 70), (8, (1, 1), 30, 17, 2)])
def test_vp9_loopfilter_frames_batch(bd, ss, sbc, sbr, npics):
    """ffhip_vp9_loopfilter_frames_dev (round 4): the pictures of a batch — each its own planes and tables — come out exactly as from
    launches of their own (which the tests above pin to the oracle superblock by superblock); 70 pictures: three launches"""
    import torch
    from ffmpeg_amd import vp9, _lib
    rng = np.random.default_rng(7000 + 100 * sbc + sbr + npics + bd)
    lim, mblim = G.filter_lut(int(rng.integers(0, 8)))
    cols, rows = 8 * sbc - int(rng.integers(0, 8)), 8 * sbr - int(rng.integers(0, 8))
    sub = 1 if ss == (1, 1) else 0
    batch, single, keep = [], [], []
    sy = suv = None
    for i in range(npics):
        planes = [_plane(rng, 64 * sbr, 64 * sbc, 12, bd), _plane(rng, (64 >> sub) * sbr, (64 >> sub) * sbc, 12 if not sub else 4, bd),
                  _plane(rng, (64 >> sub) * sbr, (64 >> sub) * sbc, 12 if not sub else 4, bd)]
        sy, suv = planes[0].strides[0], planes[1].strides[0]
        filt = np.zeros(sbr * sbc, G.FILTER_DT)
        for r in range(sbr):
            for c in range(sbc):
                filt[r * sbc + c] = G.structured(rng, r, c, cols, rows)
        tabs = vp9.lf_sb_tables(filt.view(np.uint8).reshape(sbr * sbc, 192), sbc, sbr, lim, mblim)   # (4:4:4 reads the luma tables only)
        d_tabs = torch.from_numpy(tabs.view(np.int32)).cuda()
        a = [torch.from_numpy(p.view(np.uint8).reshape(-1).copy()).cuda() for p in planes]
        b = [t.clone() for t in a]
        batch.append((a[0], a[1], a[2], d_tabs))
        single.append(b)
        keep.append((planes, d_tabs))
    vp9.loopfilter_frames(batch, sy, suv, cols, rows, bit_depth=bd, ss=ss)
    for i in range(npics):
        b = single[i]
        vp9.loopfilter_frame(b[0], b[1], b[2], sy, suv, cols, rows, keep[i][1], bit_depth=bd, ss=ss)
    torch.cuda.synchronize()
    assert _lib.lib().ffhip_stream_synchronize(None) == 0
    changed = 0
    for i in range(npics):
        for k in range(3):
            x, y = batch[i][k].cpu().numpy(), single[i][k].cpu().numpy()
            assert np.array_equal(x, y), (i, k)
            changed += int((x != keep[i][0][k].view(np.uint8).reshape(-1)).sum())
    assert changed > 1000


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("ss", [(1, 0), (0, 1)], ids=["422", "440"])
@pytest.mark.parametrize("sbc,sbr,kind", [(9, 5, "structured"), (7, 6, "bits1"), (1, 1, "structured"), (2, 13, "bits0"), (30, 17, "structured")])
def test_vp9_loopfilter_frame_422_440(sbc, sbr, kind, ss, bd):
    """VP9 4:2:2 (ss_h 1, ss_v 0) and 4:4:0 (0, 1): rectangular chroma superblocks (ffhip_vp9_loopfilter_frame_ssc_dev, tables from
    ffhip_vp9_lf_sb_tables + ffhip_vp9_lf_sb_ctables) == the oracle's ffo_vp9_loopfilter_sb superblock by superblock (pinned to the
    reference's ff_vp9_loopfilter_sb at these shifts in tests/test_vp9_lf_sb_cpu.py)"""
    import torch
    from ffmpeg_amd import vp9, _lib
    ss_h, ss_v = ss
    rng = np.random.default_rng(1000 * sbc + 10 * sbr + bd + len(kind) + 7 * ss_h)
    lim, mblim = G.filter_lut(int(rng.integers(0, 8)))
    cols, rows = 8 * sbc, 8 * sbr
    if kind == "structured":
        cols, rows = cols - int(rng.integers(0, 8)), rows - int(rng.integers(0, 8))
    cw, chh = 64 >> ss_h, 64 >> ss_v
    planes = [_plane(rng, 64 * sbr, 64 * sbc, 12, bd), _plane(rng, chh * sbr, cw * sbc, 4, bd), _plane(rng, chh * sbr, cw * sbc, 4, bd)]
    before = [p.copy() for p in planes]
    filt = np.zeros(sbr * sbc, G.FILTER_DT)
    O = ffi.oracle()
    L = _lib.lib()
    for r in range(sbr):
        for c in range(sbc):
            for _ in range(50):                   # arbitrary bits may ask for a 16-wide chroma filter at a tile's last position: drawn again
                f = G.structured(rng, r, c, cols, rows, ss_h, ss_v) if kind == "structured" else G.random_bits(rng, int(kind[-1]))
                probe = np.zeros(128, np.uint32)
                if L.ffhip_vp9_lf_sb_ctables(probe.ctypes.data, np.ascontiguousarray(f).ctypes.data, 8 * r, 8 * c, ss_h, ss_v, lim.ctypes.data,
                                             mblim.ctypes.data) == 0:
                    break
            else:
                pytest.fail("no acceptable filter drawn")
            filt[r * sbc + c] = f
            level, mask = np.ascontiguousarray(f["level"]), np.ascontiguousarray(f["mask"])
            at = [planes[0].ctypes.data + r * 64 * planes[0].strides[0] + c * 64 * planes[0].itemsize] + \
                 [p.ctypes.data + r * chh * p.strides[0] + c * cw * p.itemsize for p in planes[1:]]
            O.ffo_vp9_loopfilter_sb(bd, ss_h, ss_v, ptr(level, u8p), ptr(mask, u8p), 8 * r, 8 * c, *(C.cast(a, u8p) for a in at),
                                    planes[0].strides[0], planes[1].strides[0], ptr(lim, u8p), ptr(mblim, u8p))
    tabs, ctabs = vp9.lf_sb_tables_ss(filt.view(np.uint8).reshape(sbr * sbc, 192), sbc, sbr, lim, mblim, ss)
    dev = [torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).cuda() for b in before]
    d_tabs, d_ctabs = torch.from_numpy(tabs.view(np.int32)).cuda(), torch.from_numpy(ctabs.view(np.int32)).cuda()
    vp9.loopfilter_frame_ssc(dev[0], dev[1], dev[2], before[0].strides[0], before[1].strides[0], cols, rows, d_tabs, d_ctabs, ss, bit_depth=bd)
    torch.cuda.synchronize()
    assert L.ffhip_stream_synchronize(None) == 0
    changed = 0
    for k, (d, want, b) in enumerate(zip(dev, planes, before)):
        got = d.cpu().numpy().view(want.dtype).reshape(want.shape)
        h, w = (8 * rows, 8 * cols) if k == 0 else ((8 >> ss_v) * rows, (8 >> ss_h) * cols)
        bad = np.argwhere(got[:h, :w] != want[:h, :w])
        assert bad.size == 0, (k, bad[:5], len(bad))
        outside = got != b
        outside[:h, :w] = False
        assert not outside.any()
        changed += int((want[1:] != b[1:]).sum()) if k else 0
    assert changed > (20 * sbc * sbr if sbc * sbr > 8 else -1)
    with pytest.raises(Exception):               # the 4:2:0 / 4:4:4 entry points refuse these formats by name
        vp9.loopfilter_frame(dev[0], dev[1], dev[2], before[0].strides[0], before[1].strides[0], cols, rows, d_tabs, bit_depth=bd, ss=ss)


@pytest.mark.parametrize("bd,ss,sbc,sbr,npics", [(8, (1, 0), 9, 5, 5), (10, (0, 1), 7, 6, 3), (8, (0, 1), 3, 2, 40), (12, (1, 0), 12, 9, 2)])
def test_vp9_loopfilter_frames_ssc_batch(bd, ss, sbc, sbr, npics):
    """ffhip_vp9_loopfilter_frames_ssc_dev (round 5): N 4:2:2 / 4:4:0 pictures — each its own planes, luma tables and chroma tables — in one
    launch come out exactly as from launches of their own (ffhip_vp9_loopfilter_frame_ssc_dev, pinned to the oracle superblock by superblock
    above); 40 pictures: two launches.  ffhip_vp9_loopfilter_frames_dev refuses these formats by name."""
    import torch
    from ffmpeg_amd import vp9, _lib
    ss_h, ss_v = ss
    rng = np.random.default_rng(9000 + 100 * sbc + sbr + npics + bd)
    lim, mblim = G.filter_lut(int(rng.integers(0, 8)))
    cols, rows = 8 * sbc - int(rng.integers(0, 8)), 8 * sbr - int(rng.integers(0, 8))
    cw, chh = 64 >> ss_h, 64 >> ss_v
    L = _lib.lib()
    batch, single, keep = [], [], []
    sy = suv = None
    for i in range(npics):
        planes = [_plane(rng, 64 * sbr, 64 * sbc, 12, bd), _plane(rng, chh * sbr, cw * sbc, 4, bd), _plane(rng, chh * sbr, cw * sbc, 4, bd)]
        sy, suv = planes[0].strides[0], planes[1].strides[0]
        filt = np.zeros(sbr * sbc, G.FILTER_DT)
        for r in range(sbr):
            for c in range(sbc):
                filt[r * sbc + c] = G.structured(rng, r, c, cols, rows, ss_h, ss_v)
        tabs, ctabs = vp9.lf_sb_tables_ss(filt.view(np.uint8).reshape(sbr * sbc, 192), sbc, sbr, lim, mblim, ss)
        d_tabs, d_ctabs = torch.from_numpy(tabs.view(np.int32)).cuda(), torch.from_numpy(ctabs.view(np.int32)).cuda()
        a = [torch.from_numpy(p.view(np.uint8).reshape(-1).copy()).cuda() for p in planes]
        b = [t.clone() for t in a]
        batch.append((a[0], a[1], a[2], d_tabs, d_ctabs))
        single.append(b)
        keep.append((planes, d_tabs, d_ctabs))
    vp9.loopfilter_frames_ssc(batch, sy, suv, cols, rows, ss, bit_depth=bd)
    for i in range(npics):
        b = single[i]
        vp9.loopfilter_frame_ssc(b[0], b[1], b[2], sy, suv, cols, rows, keep[i][1], keep[i][2], ss, bit_depth=bd)
    torch.cuda.synchronize()
    assert L.ffhip_stream_synchronize(None) == 0
    changed = 0
    for i in range(npics):
        for k in range(3):
            x, y = batch[i][k].cpu().numpy(), single[i][k].cpu().numpy()
            assert np.array_equal(x, y), (i, k)
            changed += int((x != keep[i][0][k].view(np.uint8).reshape(-1)).sum())
    assert changed > 1000
    with pytest.raises(Exception, match="chroma tables"):
        vp9.loopfilter_frames([t[:4] for t in batch[:1]], sy, suv, cols, rows, bit_depth=bd, ss=ss)
