"""GPU tier of tests/test_h264_stream_cpu.py: the reference's WHOLE H.264 decoder (oracle/_ref/libffref_h264dec.so) decodes the streams of
tests/h264_bitstream.py with the `hip` recorder at its two call sites, and every finished picture is executed by
ffhip_h264_picture_flush() on a DEVICE mirror of the decoder's picture arena — the decoded-picture buffer lives in HBM, later pictures'
motion compensation reads what earlier flushes wrote there, nothing comes back to the host before the end of the stream.  The
downloaded pictures must equal the plain decode of the same stream, sample for sample: I / P pictures with one to three references,
several slices per picture, disable_deblocking_filter_idc 0 / 1 / 2, field pictures (PAFF); 8 and 10 bits."""
import ctypes as C

import numpy as np
import pytest

import h264_stream_driver as D

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not D.have(), reason="oracle/_ref/libffref_h264dec.so not built")]


def _gpu_flush_factory():
    import torch
    from ffmpeg_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    L = _lib.lib()
    state = {}

    def make(arena_base, arena_bytes):
        dev = torch.full((arena_bytes,), 0x55, dtype=torch.uint8, device="cuda:0")
        state["dev"] = dev
        counts = {"pictures": 0}
        stream = torch.cuda.current_stream().cuda_stream

        def flush(opaque, pic, dst_off, stride, mb_w, mb_h, field):
            d = dev.data_ptr()
            dp = (C.c_void_p * 3)(*[d + dst_off[i] for i in range(3)])
            rp = (C.c_void_p * 3)(d, d, d)
            st = (C.c_int * 3)(stride[0], stride[1], stride[2])
            counts["pictures"] += not (field & 2)     # (an MBAFF frame's three inter objects: counted once, with its chains)
            r = L.ffhip_h264_picture_flush(pic, dp, st, rp, stream)
            torch.cuda.synchronize()           # the decoder frees the picture object when this returns
            return r

        def flush_mbaff(opaque, chains, dst_off, stride, mb_w, mb_h):
            d = dev.data_ptr()
            dp = (C.c_void_p * 3)(*[d + dst_off[i] for i in range(3)])
            st = (C.c_int * 3)(stride[0], stride[1], stride[2])
            counts["pictures"] += 1
            counts["mbaff_frames"] = counts.get("mbaff_frames", 0) + 1
            r = L.ffhip_h264_mbaff_flush(chains, dp, st, stream)
            torch.cuda.synchronize()
            return r
        flush.mbaff = flush_mbaff
        return flush, counts

    def read_back(arena_base, used):
        host = state["dev"][:used].cpu().numpy()
        C.memmove(arena_base, host.ctypes.data, used)
    return make, read_back


def _check(aus, npictures):
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) > 0
    make, read_back = _gpu_flush_factory()
    got, st, counts = D.decode(aus, make_flush=make, read_back=read_back)
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0, st
    assert st["pictures"] == npictures == counts["pictures"] and st["mbs_hl"] > 0 and st["mbs_filter"] > 0, (st, counts)
    assert len(plain) == len(got)
    for i, (a, b) in enumerate(zip(plain, got)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())


@pytest.mark.parametrize("bit_depth", [8, 10])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_i_and_p_pictures(bit_depth, seed):
    aus, ws = D.stream_ip(bit_depth, seed)
    _check(aus, 5)


@pytest.mark.parametrize("bit_depth", [8, 10])
@pytest.mark.parametrize("seed", [4, 5])
def test_several_slices_and_deblocking_modes(bit_depth, seed):
    aus, ws = D.stream_slices(bit_depth, seed)
    _check(aus, 5)


@pytest.mark.parametrize("bit_depth", [8, 10])
def test_field_pictures(bit_depth):
    aus, ws = D.stream_fields(bit_depth, 6)
    _check(aus, 6)


def test_a_larger_picture():
    """CIF-sized pictures: 22 x 18 macroblocks, six pictures, three slices"""
    import h264_bitstream as B
    p = B.Params(mb_w=22, mb_h=18, seed=11)
    w = B.StreamWriter(p)
    pics = [{"type": "I", "slices": [0, 150], "deblock": [(0, 0, 0), (0, 1, 1)]}]
    for k in range(1, 6):
        pics.append({"type": "P", "slices": [0, 100 + 7 * k, 300], "deblock": [(0, 0, 0), (2, -1, 1), (0, 2, -2)], "num_ref": min(k, 3)})
    _check(w.stream(pics), 6)


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_mbaff_frames_between_field_pictures(seed):
    """MBAFF frames (round 6; 4:2:0) between plain field pictures: the frame's three inter objects through ffhip_h264_picture_flush(),
    its intra macroblocks and loop-filter calls through ffhip_h264_mbaff_flush() (ffmpeg_amd/csrc/h264_mbaff.hip), all on the device mirror"""
    aus, ws = D.stream_mbaff_and_fields(seed)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == 4
    make, read_back = _gpu_flush_factory()
    got, st, counts = D.decode(aus, make_flush=make, read_back=read_back)
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0 and st["plain_pictures"] == 0, st
    assert st["pictures"] == 6 == counts["pictures"] and st["mbaff_pictures"] == 2 == counts["mbaff_frames"], (st, counts)
    for i, (a, b) in enumerate(zip(plain, got)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())


@pytest.mark.parametrize("name", sorted(D.MBAFF_CASES))
def test_mbaff_streams(name):
    """Streams of MBAFF frames (see tests/test_h264_stream_cpu.py): I / P / B, every intra mode the MBAFF neighbourhood allows, field
    macroblocks on reference fields, direct prediction, weights, the 8x8 transform, all deblocking modes — executed on the device"""
    gen, kw, npic, dstats, cstats = D.MBAFF_CASES[name]
    aus, ws = gen(**kw)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == npic
    make, read_back = _gpu_flush_factory()
    got, st, counts = D.decode(aus, make_flush=make, read_back=read_back)
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0 and st["plain_pictures"] == 0, st
    assert st["pictures"] == npic == counts["pictures"] == st["mbaff_pictures"] == counts["mbaff_frames"], (st, counts)
    for k in dstats:
        assert st[k] > 0, (k, st)
    for i, (a, b) in enumerate(zip(plain, got)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())


@pytest.mark.parametrize("bit_depth", [8, 10])
def test_mbaff_1080i(bit_depth):
    """1920 x 1088 MBAFF frames (120 x 68 macroblocks: 34 macroblock-pair rows, i.e. 34 waves walking 120 pairs each behind one another in the
    intra and in the filter kernel — the dependency protocol under real concurrency): I + P, ~25 000 macroblocks, half of them field
    macroblocks, ~200 000 recorded filter calls"""
    aus, ws = D.stream_mbaff_p(seed=47, mb_w=120, mb_h=68, n=2, bit_depth=bit_depth)
    plain, st0, _ = D.decode(aus, arena_bytes=320 << 20)
    assert st0["damaged"] == 0 and len(plain) == 2
    make, read_back = _gpu_flush_factory()
    got, st, counts = D.decode(aus, make_flush=make, read_back=read_back, arena_bytes=320 << 20)
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0 and st["plain_pictures"] == 0, st
    assert st["pictures"] == 2 == counts["pictures"] == st["mbaff_pictures"] and st["mbs_field"] > 4000, (st, counts)
    for i, (a, b) in enumerate(zip(plain, got)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())


@pytest.mark.parametrize("name", sorted(D.LOSSLESS_CASES))
def test_lossless_streams(name):
    """The lossless transform bypass (see tests/test_h264_stream_cpu.py): intra macroblocks through the bypass form of the reconstruction
    phases (DPCM under profile_idc 244), inter macroblocks through add_pixels4 / 8_clear records — executed on the device"""
    gen, kw, npic, dstats = D.LOSSLESS_CASES[name]
    aus, ws = gen(**kw)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == (npic // 2 if kw.get("fields") else npic)
    make, read_back = _gpu_flush_factory()
    got, st, counts = D.decode(aus, make_flush=make, read_back=read_back)
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0 and st["plain_pictures"] == 0, st
    assert st["pictures"] == npic == counts["pictures"] and st["mbs_bypass"] > 0, (st, counts)
    for k in dstats:
        assert st[k] > 0, (k, st)
    for i, (a, b) in enumerate(zip(plain, got)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())


@pytest.mark.parametrize("name", sorted(D.ROUND6_CASES))
def test_b_weighted_8x8_transform_and_422_streams(name):
    """Round 6 (see tests/test_h264_stream_cpu.py): B slices with spatial / temporal direct prediction, explicit and implicit weights, the 8x8
    transform (Intra8x8 and inter), non-reference B pictures, High 4:2:2; the decoder derives the state, the device executes the lists."""
    gen, kw, npic, dstats, wstats = D.ROUND6_CASES[name]
    aus, ws = gen(**kw)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == (npic // 2 if kw.get("fields") else npic)
    make, read_back = _gpu_flush_factory()
    got, st, counts = D.decode(aus, make_flush=make, read_back=read_back)
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0 and st["plain_pictures"] == 0, st
    assert st["pictures"] == npic == counts["pictures"], (st, counts)
    for k in dstats:
        assert st[k] > 0, (k, st)
    for i, (a, b) in enumerate(zip(plain, got)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())
