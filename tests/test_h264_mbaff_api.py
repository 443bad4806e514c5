"""The FFHipH264Mbaff object's host side (include/ffhip.h, ffmpeg_amd/csrc/h264_mbaff.hip) without a device: what it accepts, the order it
insists on, the lists it exports, and that flush() refuses — before it touches the device — a call that reaches outside its macroblock
pair's tile."""
import ctypes as C

import numpy as np
import pytest

from ffmpeg_amd import _lib

EINVAL = -22


class Edge(C.Structure):            # == FFHipH264Edge
    _fields_ = [("offset", C.c_int32), ("kind", C.c_uint8), ("alpha", C.c_uint8), ("beta", C.c_uint8), ("pad", C.c_uint8), ("tc0", C.c_int8 * 4)]


class MbaffLists(C.Structure):      # == FFHipH264MbaffLists
    _fields_ = [("mb_w", C.c_int), ("mb_h", C.c_int), ("recs", C.c_void_p), ("geo", C.c_void_p), ("coefs", C.c_void_p), ("intra_row", C.c_void_p),
                ("nrecs", C.c_int32), ("ncoefs", C.c_int32), ("calls", C.c_void_p * 3), ("pair_end", C.c_void_p * 3), ("ncalls", C.c_int32 * 3),
                ("bit_depth", C.c_int)]


def make(mb_w=4, mb_h=4):
    L = _lib.lib()
    m = C.c_void_p()
    assert L.ffhip_h264_mbaff_create(C.byref(m), mb_w, mb_h) == 0 and m
    L.ffhip_h264_mbaff_begin(m)
    return L, m


def test_geometry():
    L = _lib.lib()
    m = C.c_void_p()
    assert L.ffhip_h264_mbaff_create(C.byref(m), 4, 5) == EINVAL          # macroblock PAIRS: an even number of rows
    assert L.ffhip_h264_mbaff_create(C.byref(m), 0, 4) == EINVAL
    assert L.ffhip_h264_mbaff_create(None, 4, 4) == EINVAL
    assert L.ffhip_h264_mbaff_create_fmt(C.byref(m), 4, 4, 11) == EINVAL     # 8, 9, 10, 12, 14
    assert L.ffhip_h264_mbaff_create_fmt(C.byref(m), 4, 4, 10) == 0 and m
    ls = MbaffLists()
    assert L.ffhip_h264_mbaff_lists(m, C.byref(ls)) == 0 and ls.bit_depth == 10
    L.ffhip_h264_mbaff_free(C.byref(m))
    assert not m
    L.ffhip_h264_mbaff_free(C.byref(m))                                   # (a null object: nothing to do)


def test_calls_are_kept_in_order_per_pair():
    L, m = make()
    H_LUMA, V_LUMA = 1, 0          # FFHIP_H264_LF_H_LUMA / V_LUMA
    stride = 64 + 32
    e = Edge(offset=4, kind=H_LUMA, alpha=20, beta=5, pad=2, tc0=(1, 1, 1, 1))
    assert L.ffhip_h264_mbaff_filter_call(m, 0, 1, 0, C.byref(e)) == 0                      # pair (1, 0), from its top macroblock
    e2 = Edge(offset=16 * stride + 16, kind=V_LUMA, alpha=20, beta=5, pad=1, tc0=(0, 0, 0, 0))
    assert L.ffhip_h264_mbaff_filter_call(m, 0, 1, 1, C.byref(e2)) == 0                     # the same pair, its bottom macroblock
    assert L.ffhip_h264_mbaff_filter_call(m, 0, 3, 2, C.byref(e2)) == 0                     # pair (3, 1): pairs in between have no calls
    assert L.ffhip_h264_mbaff_filter_call(m, 0, 0, 0, C.byref(e)) == EINVAL                 # back to an earlier pair
    bad = Edge(offset=6, kind=H_LUMA, alpha=1, beta=1, pad=0)
    assert L.ffhip_h264_mbaff_filter_call(m, 0, 3, 3, C.byref(bad)) == EINVAL               # not a multiple of four
    bad = Edge(offset=8, kind=9, alpha=1, beta=1, pad=0)
    assert L.ffhip_h264_mbaff_filter_call(m, 0, 3, 3, C.byref(bad)) == EINVAL               # no such member
    assert L.ffhip_h264_mbaff_filter_call(m, 3, 3, 3, C.byref(e)) == EINVAL                 # no such plane
    assert L.ffhip_h264_mbaff_filter_call(m, 0, 4, 0, C.byref(e)) == EINVAL                 # no such macroblock
    ls = MbaffLists()
    assert L.ffhip_h264_mbaff_lists(m, C.byref(ls)) == 0
    assert (ls.mb_w, ls.mb_h, ls.nrecs, ls.ncalls[0], ls.ncalls[1], ls.ncalls[2]) == (4, 4, 0, 3, 0, 0)
    ends = np.ctypeslib.as_array((C.c_int32 * 8).from_address(ls.pair_end[0]))
    assert list(ends) == [0, 2, 2, 2, 2, 2, 2, 3]                                           # one past each pair's last call, row-major
    assert list(np.ctypeslib.as_array((C.c_int32 * 3).from_address(ls.intra_row))) == [0, 0, 0]
    L.ffhip_h264_mbaff_begin(m)
    assert L.ffhip_h264_mbaff_lists(m, C.byref(ls)) == 0 and ls.ncalls[0] == 0
    L.ffhip_h264_mbaff_free(C.byref(m))
    assert not m


def test_flush_refuses_a_call_outside_its_pair_before_touching_the_device():
    L, m = make()
    stride = 96
    # pair (1, 0) claims a call whose samples lie in pair (2, 0)
    e = Edge(offset=32 + 4, kind=1, alpha=20, beta=5, pad=0, tc0=(1, 1, 1, 1))
    assert L.ffhip_h264_mbaff_filter_call(m, 0, 1, 0, C.byref(e)) == 0
    planes = (C.c_void_p * 3)(4096, 4096, 4096)       # never dereferenced: the refusal comes first
    st = (C.c_int * 3)(stride, stride // 2, stride // 2)
    assert L.ffhip_h264_mbaff_flush(m, planes, st, None) == EINVAL
    assert b"outside macroblock pair" in L.ffhip_last_error()
    st_bad = (C.c_int * 3)(stride + 2, stride // 2, stride // 2)
    assert L.ffhip_h264_mbaff_flush(m, planes, st_bad, None) == EINVAL
    L.ffhip_h264_mbaff_free(C.byref(m))


def test_intra_macroblocks_arrive_in_decoding_order():
    L, m = make()

    class IntraMB(C.Structure):     # the head of FFHipH264IntraMB (108 bytes)
        _fields_ = [("mb_x", C.c_int16), ("mb_y", C.c_int16), ("type", C.c_uint8), ("rest", C.c_uint8 * 103)]
    assert C.sizeof(IntraMB) == 108
    nnz = (C.c_uint8 * (15 * 8))()
    mb = (C.c_int16 * (16 * 48))()
    dc = (C.c_int16 * (3 * 16 * 2))()

    def rec(x, y, field):
        d = IntraMB(mb_x=x, mb_y=y, type=1)   # FFHIP_H264_INTRA_16x16
        return L.ffhip_h264_mbaff_intra_mb(m, C.byref(d), field, nnz, mb, dc, None)
    assert rec(0, 0, 1) == 0 and rec(0, 1, 1) == 0 and rec(2, 1, 0) == 0 and rec(1, 2, 0) == 0
    assert rec(1, 2, 0) == EINVAL          # twice
    assert rec(0, 2, 0) == EINVAL          # a pair further left in the same row
    assert rec(1, 1, 0) == EINVAL          # an earlier row
    assert rec(4, 3, 0) == EINVAL          # outside
    ls = MbaffLists()
    assert L.ffhip_h264_mbaff_lists(m, C.byref(ls)) == 0 and ls.nrecs == 4
    geo = np.ctypeslib.as_array((C.c_uint32 * 4).from_address(ls.geo))
    assert [(int(g) & 0xFFF, (int(g) >> 12) & 0xFFF, int(g) >> 24) for g in geo] == [(0, 0, 1), (0, 1, 1), (2, 1, 0), (1, 2, 0)]
    assert list(np.ctypeslib.as_array((C.c_int32 * 3).from_address(ls.intra_row))) == [0, 3, 4]
    L.ffhip_h264_mbaff_free(C.byref(m))
