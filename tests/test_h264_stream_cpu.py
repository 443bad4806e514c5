"""The reference's WHOLE H.264 decoder with the `hip` recorder at its two call sites, on the CPU tier.

tests/h264_bitstream.py writes multi-picture CAVLC streams (syntax elements only: I_PCM, Intra16x16, Intra4x4, P_L0_16x16 / 16x8 / 8x16,
P_8x8 with every sub-type, P_Skip, motion vector differences that carry blocks far outside the picture, residual blocks with escapes, several
slices per picture, disable_deblocking_filter_idc 0 / 1 / 2, field pictures).  oracle/_ref/libffref_h264dec.so — h264dec.c,
h264_slice.c, h264_cavlc.c, h264_mvpred.h, h264_refs.c, h264_picture.c ... compiled where they lie — decodes each stream twice through
avcodec_send_packet() / avcodec_receive_frame():
  plain:   the decoder as it is;
  record:  decode_slice()'s ff_h264_hl_decode_mb() and loop_filter()'s ff_h264_filter_mb_fast() / ff_h264_filter_mb() replaced by
           ff_h264_hip_hl_decode_mb() / ff_h264_hip_filter_mb() (integration/avcodec_h264_picture_hip.c); a finished picture's lists are
           executed on the decoder's own picture buffer by oracle/emul_h264_picture.cpp before the next picture refers to it.
The per-macroblock state (mv_cache, ref_cache, non_zero_count_cache, neighbour types, qp tables, reference lists) is derived by the
decoder from the bitstream — not written by a test.  Every output frame must be identical, sample for sample."""
import numpy as np
import pytest

import h264_stream_driver as D

pytestmark = pytest.mark.skipif(not D.have(), reason="oracle/_ref/libffref_h264dec.so or oracle/libffemul.so not built")


def _both(aus):
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and st0["pictures"] == 0 and len(plain) > 0
    rec, st1, counts = D.decode(aus, make_flush=lambda base, size: D.cpu_flush(base))
    return plain, rec, st1, counts


def _check(aus, wstats, min_pictures):
    plain, rec, st, counts = _both(aus)
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0, st
    assert st["pictures"] == min_pictures and counts["pictures"] == min_pictures, (st, counts)
    assert st["mbs_hl"] > 0 and st["mbs_filter"] > 0 and counts["inter_blocks"] > 0 and counts["intra_mbs"] > 0
    assert len(plain) == len(rec)
    for i, (a, b) in enumerate(zip(plain, rec)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())
    # the pictures are not trivial: the streams move
    assert any(not np.array_equal(plain[0][0], f[0]) for f in plain[1:])
    return st, counts


@pytest.mark.parametrize("bit_depth", [8, 10])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_i_and_p_pictures(bit_depth, seed):
    aus, ws = D.stream_ip(bit_depth, seed)
    st, counts = _check(aus, ws, 5)
    assert ws["pcm"] and ws["i16"] and ws["i4"] and ws["p16"] and ws["p168"] and ws["p88"] and ws["skip"] and ws["escapes"]


@pytest.mark.parametrize("bit_depth", [8, 10])
@pytest.mark.parametrize("seed", [4, 5])
def test_several_slices_and_deblocking_modes(bit_depth, seed):
    aus, ws = D.stream_slices(bit_depth, seed)
    _check(aus, ws, 5)


@pytest.mark.parametrize("bit_depth", [8, 10])
def test_field_pictures(bit_depth):
    aus, ws = D.stream_fields(bit_depth, 6)
    _check(aus, ws, 6)


def test_a_larger_picture():
    """CIF-sized pictures: 22 x 18 macroblocks, six pictures, three slices"""
    import h264_bitstream as B
    p = B.Params(mb_w=22, mb_h=18, seed=11)
    w = B.StreamWriter(p)
    pics = [{"type": "I", "slices": [0, 150], "deblock": [(0, 0, 0), (0, 1, 1)]}]
    for k in range(1, 6):
        pics.append({"type": "P", "slices": [0, 100 + 7 * k, 300], "deblock": [(0, 0, 0), (2, -1, 1), (0, 2, -2)], "num_ref": min(k, 3)})
    _check(w.stream(pics), w.stats, 6)


def test_a_picture_buffer_out_of_reach_of_the_base_is_refused():
    """The records hold 32-bit offsets from one base (flush()'s ref[]): a decoded-picture buffer 3 GiB away from it must make the recorder
    fail the picture (FFHIP_EINVAL, sticky) — not wrap the offsets (integration/avcodec_h264_picture_hip.c fits32())."""
    aus, ws = D.stream_ip(8, 1, n=2)
    calls = []

    def make(base, size):
        def flush(*a):
            calls.append(a)
            return 0
        return flush, {}
    frames, st, _ = D.decode(aus, make_flush=make, base_shift=3 << 30)
    assert st["errors"] > 0 and st["first_error"] == -22 and st["pictures"] == 2
    assert not calls, "a picture whose recorder failed must not be flushed"


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_mbaff_frames_between_field_pictures(seed):
    """mb_adaptive_frame_field_flag = 1 (round 6): the stream's frames are MBAFF frames — frame and field macroblock pairs mixed, the writer
    emits mb_field_decoding_flag per pair (skipped top macroblocks included) — between plain field pictures that predict from them and are
    predicted from by them.  An MBAFF frame is recorded into FOUR objects (integration/avcodec_h264_picture_hip.c: the frame macroblocks,
    the top- and the bottom-field macroblocks, the chains) and nothing stays on the C path."""
    aus, ws = D.stream_mbaff_and_fields(seed)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == 4
    rec, st, counts = D.decode(aus, make_flush=lambda base, size: D.cpu_flush(base))
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0, st
    assert st["plain_pictures"] == 0 and st["pictures"] == 6 == counts["pictures"] and st["mbaff_pictures"] == 2 == counts["mbaff_frames"], (st, counts)
    assert counts["mbaff_field_intra_mbs"] > 0 and counts["mbaff_calls_mbaff_member"] > 0 and counts["mbaff_calls_field_stride"] > 0, counts
    for i, (a, b) in enumerate(zip(plain, rec)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())


@pytest.mark.parametrize("name", sorted(D.MBAFF_CASES))
def test_mbaff_streams(name):
    """Streams of MBAFF frames only (4:2:0; 8 and 10 bits): I / P / B slices starting on macroblock pairs, every Intra16x16 and chroma prediction mode
    the MBAFF neighbour derivation (6.4.12.2) allows, isolated Intra4x4 / Intra8x8 macroblocks with every mode, field macroblocks predicting
    from reference FIELDS (ref_list[l][16 + 2 i + parity]) in and beyond the picture, direct prediction (spatial, temporal), explicit and
    implicit weights, the 8x8 transform, disable_deblocking_filter_idc 0 / 1 / 2.  The decoder derives every macroblock's state; the four
    objects' lists executed on the CPU (oracle/emul_h264_picture.cpp for the three inter objects, oracle/emul_h264_mbaff.cpp for the
    chains) must give the plain decode, sample for sample."""
    gen, kw, npic, dstats, cstats = D.MBAFF_CASES[name]
    aus, ws = gen(**kw)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == npic
    assert ws["i16"] > 0 and ws["i4"] + ws["i8"] > 0, ws
    rec, st, counts = D.decode(aus, make_flush=lambda base, size: D.cpu_flush(base))
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0 and st["plain_pictures"] == 0, st
    assert st["pictures"] == npic == counts["pictures"] == st["mbaff_pictures"] == counts["mbaff_frames"], (st, counts)
    for k in dstats:
        assert st[k] > 0, (k, st)
    for k in cstats:
        assert counts[k] > 0, (k, counts)
    assert len(rec) == len(plain)
    for i, (a, b) in enumerate(zip(plain, rec)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())
    assert any(not np.array_equal(plain[0][0], f[0]) for f in plain[1:])


def test_an_mbaff_frame_at_422_stays_on_the_c_path():
    """ff_h264_hip_picture_supported(): MBAFF is taken at 4:2:0 (8 - 14 bits) only — a 4:2:2 MBAFF frame is refused BEFORE its first
    macroblock and runs through the reference's functions on the C tables as a whole, between recorded pictures."""
    import h264_bitstream as B
    p = B.Params(mb_w=6, mb_h=6, bit_depth=8, chroma_format=2, frame_mbs_only=0, mbaff=1, seed=17)
    w = B.StreamWriter(p)
    pics = [{"type": "I", "slices": [0], "deblock": [(0, 0, 0)]},
            {"type": "P", "slices": [0], "deblock": [(0, 1, 0)], "field": "top", "num_ref": 2},
            {"type": "P", "slices": [0], "deblock": [(0, 0, 0)], "field": "bottom", "second_field": True, "num_ref": 3},
            {"type": "P", "slices": [0, 12], "deblock": [(0, 1, 1), (0, 0, 0)], "num_ref": 2}]
    aus = w.stream(pics)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == 3
    rec, st, counts = D.decode(aus, make_flush=lambda base, size: D.cpu_flush(base))
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0, st
    assert st["plain_pictures"] == 2 and st["pictures"] == 2 == counts["pictures"] and st["mbaff_pictures"] == 0, (st, counts)
    for i, (a, b) in enumerate(zip(plain, rec)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())


@pytest.mark.parametrize("name", sorted(D.LOSSLESS_CASES))
def test_lossless_streams(name):
    """qpprime_y_zero_transform_bypass_flag = 1 (round 6; 8 bits, 4:2:0): macroblocks whose QP'Y is 0 are decoded with the transform bypassed
    (hl_decode_mb()'s transform_bypass, h264_mb_template.c:51,190-221; h264_mb.c:614-772) — the residual added as samples (add_pixels4 /
    8_clear, modulo 256), no DC transforms — among ordinary macroblocks of the same slice (the writer walks QP through 0 .. 6).  profile_idc
    100: that is all; profile_idc 244 (High 4:4:4 Predictive): vertically / horizontally predicted Intra4x4, Intra8x8, Intra16x16 blocks and
    chroma planes are DPCM-coded (pred4x4_add, pred8x8l_filter_add, pred16x16_add, pred8x8_add: h264pred_template.c:1067-1330), which
    libffhip's packer turns into ordinary residuals of the V / H prediction by running sums.  I / P / B, fields, MBAFF, the 8x8 transform;
    intra macroblocks through the reconstruction phases' bypass form, inter macroblocks through recorded add_pixels calls."""
    gen, kw, npic, dstats = D.LOSSLESS_CASES[name]
    aus, ws = gen(**kw)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == (npic // 2 if kw.get("fields") else npic)
    rec, st, counts = D.decode(aus, make_flush=lambda base, size: D.cpu_flush(base))
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0 and st["plain_pictures"] == 0, st
    assert st["pictures"] == npic == counts["pictures"], (st, counts)
    for k in dstats:
        assert st[k] > 0, (k, st)
    assert st["mbs_bypass"] > st["mbs_hl"] // 3 and st["mbs_bypass"] < st["mbs_hl"], st      # both kinds of macroblock in the stream
    assert counts["bypass_inter_blocks"] > 0, counts
    if "mbaff" not in name:                                   # (an MBAFF frame's intra macroblocks live in its chains object)
        assert counts["bypass_intra_mbs"] > 0 and (counts["dpcm_regions"] > 0) == (kw["lossless"] == 2), counts
    assert len(rec) == len(plain)
    for i, (a, b) in enumerate(zip(plain, rec)):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())


def test_a_lossless_stream_above_8_bits_stays_on_the_c_path():
    """ff_h264_hip_picture_supported(): the transform bypass is taken at 8 bits (4:2:0 / 4:4:4) — a 10-bit lossless stream runs through the
    reference's functions as a whole"""
    aus, ws = D.stream_p_features(bit_depth=10, seed=59, lossless=2, n=3)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == 3
    rec, st, counts = D.decode(aus, make_flush=lambda base, size: D.cpu_flush(base))
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0, st
    assert st["plain_pictures"] == 3 and st["pictures"] == 0, st
    for a, b in zip(plain, rec):
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl])


@pytest.mark.parametrize("name", sorted(D.ROUND6_CASES))
def test_b_weighted_8x8_transform_and_422_streams(name):
    """Round 6: what the picture layer accepted with test-written macroblock state, now with the state the DECODER derives — B slices
    (every B macroblock and sub-macroblock type, B_Skip / B_Direct with spatial and temporal direct prediction: h264_direct.c:208-729),
    explicit weights from a pred_weight_table (h264_parse.c:30) and implicit ones (h264_slice.c:689), transform_size_8x8_flag on inter
    macroblocks and Intra8x8 (h264_cavlc.c:634), non-reference B pictures decoded out of output order, High 4:2:2 — 8 and 10 bits.  The
    decoder's own counters say that such macroblocks were recorded."""
    gen, kw, npic, dstats, wstats = D.ROUND6_CASES[name]
    aus, ws = gen(**kw)
    for k in wstats:
        assert ws[k] > 0, (k, ws)
    plain, st0, _ = D.decode(aus)
    assert st0["damaged"] == 0 and len(plain) == (npic // 2 if kw.get("fields") else npic)
    rec, st, counts = D.decode(aus, make_flush=lambda base, size: D.cpu_flush(base))
    assert st["errors"] == 0 and st["refused"] == 0 and st["damaged"] == 0 and st["plain_pictures"] == 0, st
    assert st["pictures"] == npic == counts["pictures"], (st, counts)
    for k in dstats:
        assert st[k] > 0, (k, st)
    assert len(rec) == len(plain)
    for i, (a, b) in enumerate(zip(plain, rec)):
        for pl in range(3):
            assert a[pl].shape == b[pl].shape
            assert np.array_equal(a[pl], b[pl]), "frame %d plane %d: %d samples differ" % (i, pl, (a[pl] != b[pl]).sum())
    assert any(not np.array_equal(plain[0][0], f[0]) for f in plain[1:])
