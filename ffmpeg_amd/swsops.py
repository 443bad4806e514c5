"""ffmpeg_amd.swsops — host-side mirror of libswscale's micro-op interface: the structs of include/ffhip.h's
SwsOpBackend section (layout-identical to SwsUOp / SwsFilterWeights / SwsOpExec, libswscale/uops.h:262-281, filters.h:83-121,
ops_dispatch.h:36-85) and a thin wrapper over the compiled list."""
import ctypes as C

import numpy as np

from . import _lib

U8, U16, U32, F32 = 1, 2, 3, 4
NP = {U8: np.uint8, U16: np.uint16, U32: np.uint32, F32: np.float32}
SIZE = {U8: 1, U16: 2, U32: 4, F32: 4}
INT_OF = {1: U8, 2: U16, 4: U32}
(READ_PLANAR, READ_PLANAR_FH, READ_PLANAR_FV, READ_PLANAR_FV_FMA, READ_PACKED, READ_NIBBLE, READ_BIT, READ_PALETTE, WRITE_PLANAR,
 WRITE_PACKED, WRITE_NIBBLE, WRITE_BIT, RW_SHUFFLE, PERMUTE, COPY, SWAP_BYTES, EXPAND_BIT, EXPAND_PAIR, EXPAND_QUAD, TO_U8, TO_U16,
 TO_U32, TO_F32, SCALE, ADD, MIN, MAX, UNPACK, PACK, LSHIFT, RSHIFT, CLEAR, LINEAR, LINEAR_FMA, DITHER, LUT_3D) = range(1, 37)
READS = (READ_PLANAR, READ_PLANAR_FH, READ_PLANAR_FV, READ_PACKED, READ_NIBBLE, READ_BIT, READ_PALETTE)
WRITES = (WRITE_PLANAR, WRITE_PACKED, WRITE_NIBBLE, WRITE_BIT)
PIXELS, LINES = 64, 16
STRIDE = PIXELS * 16           # sizeof(uint32_t[4]) per pixel, as checkasm's planes


class Pixel(C.Union):
    _fields_ = [("data", C.c_char * 4), ("u8", C.c_uint8), ("u16", C.c_uint16), ("u32", C.c_uint32), ("f32", C.c_float)]


class FilterWeights(C.Structure):
    _fields_ = [("filter_size", C.c_int), ("weights", C.POINTER(C.c_int)), ("num_weights", C.c_size_t), ("offsets", C.POINTER(C.c_int)),
                ("src_size", C.c_int), ("dst_size", C.c_int), ("virtual_size", C.c_double), ("offset", C.c_double),
                ("name", C.c_char * 16), ("sum_positive", C.c_int), ("sum_negative", C.c_int)]


class _Shuffle(C.Structure):
    _fields_ = [("clear_value", C.c_uint8), ("read_size", C.c_uint8), ("write_size", C.c_uint8)]


class _Filter(C.Structure):
    _fields_ = [("type", C.c_int32)]


class _Shift(C.Structure):
    _fields_ = [("amount", C.c_uint8)]


class _Move(C.Structure):
    _fields_ = [("num_moves", C.c_int32), ("dst", C.c_int8 * 6), ("src", C.c_int8 * 6)]


class _Pack(C.Structure):
    _fields_ = [("pattern", C.c_uint8 * 4)]


class _Clear(C.Structure):
    _fields_ = [("one", C.c_uint8), ("zero", C.c_uint8)]


class _Lin(C.Structure):
    _fields_ = [("one", C.c_uint32), ("zero", C.c_uint32), ("exact", C.c_uint32)]


class _Dither(C.Structure):
    _fields_ = [("y_offset", C.c_uint8 * 4), ("size_log2", C.c_uint8)]


class _Lut3d(C.Structure):
    _fields_ = [("dynamic", C.c_int32)]


class Par(C.Union):
    _fields_ = [("shuffle", _Shuffle), ("filter", _Filter), ("shift", _Shift), ("move", _Move), ("pack", _Pack), ("clear", _Clear),
                ("lin", _Lin), ("dither", _Dither), ("lut3d", _Lut3d)]


class _ShuffleMask(C.Structure):
    _fields_ = [("mask", C.c_int8 * 16), ("pixels", C.c_uint8)]


class Data(C.Union):
    _fields_ = [("kernel", C.POINTER(FilterWeights)), ("ptr", C.POINTER(Pixel)), ("scalar", Pixel), ("vec4", Pixel * 4),
                ("mat4", (Pixel * 5) * 4), ("shuffle", _ShuffleMask), ("lut3d", C.c_void_p), ("opaque", C.c_void_p)]


class UOp(C.Structure):
    _fields_ = [("type", C.c_int32), ("uop", C.c_int32), ("mask", C.c_uint8), ("par", Par), ("data", Data)]


class Exec(C.Structure):
    _fields_ = [("in_", C.c_void_p * 4), ("out", C.c_void_p * 4), ("in_stride", C.c_ssize_t * 4), ("out_stride", C.c_ssize_t * 4),
                ("in_bump", C.c_ssize_t * 4), ("out_bump", C.c_ssize_t * 4), ("width", C.c_int32), ("height", C.c_int32),
                ("slice_y", C.c_int32), ("slice_h", C.c_int32), ("block_size_in", C.c_int32 * 4), ("block_size_out", C.c_int32 * 4),
                ("in_sub_y", C.c_uint8 * 4), ("out_sub_y", C.c_uint8 * 4), ("in_sub_x", C.c_uint8 * 4), ("out_sub_x", C.c_uint8 * 4),
                ("in_bump_y", C.POINTER(C.c_int32)), ("in_offset_x", C.POINTER(C.c_int32))]


assert C.sizeof(UOp) == 112 and C.sizeof(Exec) == 272 and C.sizeof(FilterWeights) == 80

OPFUNC = C.CFUNCTYPE(None, C.POINTER(Exec), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)


class Kernel:
    """a filter kernel and the storage it points into"""

    def __init__(self, weights, offsets, filter_size, src_size):
        self.w = np.ascontiguousarray(weights, np.int32)
        self.o = np.ascontiguousarray(offsets, np.int32)
        self.c = FilterWeights(filter_size=filter_size, weights=self.w.ctypes.data_as(C.POINTER(C.c_int)), num_weights=self.w.size,
                               offsets=self.o.ctypes.data_as(C.POINTER(C.c_int)), src_size=src_size, dst_size=self.o.size)


class UOpList:
    """a list rebuilt from its fixture: the UOp array plus the arrays its pointers point into"""

    def __init__(self, z, k):
        raw = np.ascontiguousarray(z["l%d" % k])
        self.n = raw.shape[0]
        self.uops = (UOp * self.n).from_buffer_copy(raw.tobytes())
        self.keep = []
        for i in range(self.n):
            u = self.uops[i]
            if u.uop in (READ_PLANAR_FH, READ_PLANAR_FV):
                meta = z["l%d_k%d_m" % (k, i)]
                kern = Kernel(z["l%d_k%d_w" % (k, i)], z["l%d_k%d_o" % (k, i)], int(meta[0]), int(meta[1]))
                self.keep.append(kern)
                u.data.kernel = C.pointer(kern.c)
            elif u.uop == DITHER:
                m = np.ascontiguousarray(z["l%d_d%d" % (k, i)], np.uint32)
                self.keep.append(m)
                u.data.ptr = m.ctypes.data_as(C.POINTER(Pixel))

    @property
    def read(self):
        return self.uops[0]

    @property
    def write(self):
        return self.uops[self.n - 1]


class _Prefixed:
    def __init__(self, z, prefix):
        self.z, self.p = z, prefix

    def __getitem__(self, k):
        return self.z[self.p + k]


def load_lists(z, prefix=""):
    z = _Prefixed(z, prefix)
    return [UOpList(z, k) for k in range(int(z["count"][0]))]


def rw_geometry(u):
    """(planes mask, bits per pixel per plane) of a read or write micro-op"""
    ts = 8 * SIZE[u.type]
    el = 4 if u.mask & 8 else 3 if u.mask & 4 else 2 if u.mask & 2 else 1
    if u.uop in (READ_PLANAR, READ_PLANAR_FH, READ_PLANAR_FV, WRITE_PLANAR):
        return u.mask, ts
    if u.uop in (READ_PACKED, WRITE_PACKED):
        return 1, ts * el
    if u.uop in (READ_NIBBLE, WRITE_NIBBLE):
        return 1, 4
    if u.uop in (READ_BIT, WRITE_BIT):
        return 1, 1
    return 3, 8       # palette


def plain_exec(lst, src_planes, src_strides, dst_planes, dst_strides, w, h, block=1):
    """the SwsOpExec of one whole picture for an unfiltered or filtered list, as op_pass_setup builds it (ops_dispatch.c:207-290,
    620-690): planes are integer addresses (host or device)"""
    e = Exec(width=w, height=h, slice_h=h)
    rd, wr = lst.read, lst.write
    _, bi = rw_geometry(rd)
    _, bo = rw_geometry(wr)
    keep = []
    nb = (w + block - 1) // block
    for i in range(4):
        e.in_[i] = src_planes[i] if i < len(src_planes) else None
        e.out[i] = dst_planes[i] if i < len(dst_planes) else None
        e.in_stride[i] = src_strides[i] if i < len(src_strides) else 0
        e.out_stride[i] = dst_strides[i] if i < len(dst_strides) else 0
        e.block_size_in[i], e.block_size_out[i] = block * bi >> 3, block * bo >> 3
        e.in_bump[i] = e.in_stride[i] - nb * e.block_size_in[i]
        e.out_bump[i] = e.out_stride[i] - nb * e.block_size_out[i]
    if rd.uop == READ_PLANAR_FV:
        k = rd.data.kernel.contents
        o = np.ctypeslib.as_array(k.offsets, (k.dst_size,))
        b = np.zeros(k.dst_size, np.int32)
        b[:-1] = o[1:] - o[:-1] - 1
        keep.append(b)
        e.in_bump_y = b.ctypes.data_as(C.POINTER(C.c_int32))
        for i in range(4):
            if e.in_[i]:
                e.in_[i] += int(o[0]) * e.in_stride[i]
    elif rd.uop == READ_PLANAR_FH:
        k = rd.data.kernel.contents
        o = np.ctypeslib.as_array(k.offsets, (k.dst_size,))
        n = nb * block
        b = np.full(n, int(o[-1]) * bi >> 3, np.int32)
        b[:k.dst_size] = o.astype(np.int64) * bi >> 3
        keep.append(b)
        e.in_offset_x = b.ctypes.data_as(C.POINTER(C.c_int32))
        for i in range(4):
            e.block_size_in[i] = 0
            e.in_bump[i] = e.in_stride[i]
    e._keep = keep
    return e


class CompiledUOps:
    """ffhip_sws_uops_compile / _run_dev / _free"""

    def __init__(self, uops, n=None):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        n = len(uops) if n is None else n
        _lib.check(self.L.ffhip_sws_uops_compile(C.cast(uops, C.c_void_p), n, C.byref(self.h)), "ffhip_sws_uops_compile")
        self.block_size = self.L.ffhip_sws_uops_block_size(self.h)

    def run_dev(self, exec_, width, height, nframes=1, in_pitch=None, out_pitch=None, stream=None):
        bs = self.block_size
        ip = (C.c_ssize_t * 4)(*in_pitch) if in_pitch is not None else None
        op = (C.c_ssize_t * 4)(*out_pitch) if out_pitch is not None else None
        _lib.check(self.L.ffhip_sws_uops_run_dev(self.h, C.byref(exec_), 0, 0, (width + bs - 1) // bs, height, nframes, ip, op, stream),
                   "ffhip_sws_uops_run_dev")

    def close(self):
        if self.h:
            self.L.ffhip_sws_uops_free(C.byref(self.h))
            self.h = C.c_void_p()
