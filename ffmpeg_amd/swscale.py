"""Host-side mirror of the reference's libswscale entry points for the hip path.

  SwsContext(...)            ~ sws_getContext()          libswscale/utils.c:2043
  SwsContext.scale(...)      ~ sws_scale()               libswscale/swscale.c:1626   (host numpy planes)
  SwsContext.scale_batch(..) ~ (batched, HBM-resident)   no reference equivalent

Device memory, streams and process groups come from torch; every pixel is produced by the HIP
kernels in libffhip.so (there is no CPU path here).
"""
import ctypes as C

import numpy as np

from . import _lib

PIX_FMT = {"yuv420p": 0, "yuv422p": 4, "yuv444p": 5, "yuva420p": 33, "yuva422p": 78, "yuva444p": 79, "yuvj420p": 12, "yuvj422p": 13, "yuvj444p": 14, "rgb24": 2, "bgr24": 3, "nv12": 23, "nv21": 24, "argb": 25, "rgba": 26, "abgr": 27, "bgra": 28}
SWS_BILINEAR, SWS_BICUBIC, SWS_POINT, SWS_AREA, SWS_BICUBLIN = 2, 4, 0x10, 0x20, 0x40
SWS_GAUSS, SWS_SINC, SWS_LANCZOS = 0x80, 0x100, 0x200
SWS_ACCURATE_RND, SWS_BITEXACT = 0x40000, 0x80000


# AVPixelFormat -> (semi-planar, hsub, vsub) of the formats above 8 bits the scaler takes
HBD_FMT = {45: (0, 1, 1), 47: (0, 1, 0), 49: (0, 0, 0), 60: (0, 1, 1), 62: (0, 1, 1), 64: (0, 1, 0), 66: (0, 0, 0), 68: (0, 0, 0), 70: (0, 1, 0),
           123: (0, 1, 1), 125: (0, 1, 1), 127: (0, 1, 0), 129: (0, 1, 0), 131: (0, 0, 0), 133: (0, 0, 0), 158: (1, 1, 1), 169: (1, 1, 1),
           209: (1, 1, 1)}
YUVA_FMT = {33: 0, 78: 4, 79: 5}   # yuva420p / 422p / 444p -> the base formats (libavutil/pixfmt.h)


def plane_shapes(fmt, w, h):
    """[(rows, bytes_per_row)] of the planes of one frame."""
    fmt = {12: 0, 13: 4, 14: 5}.get(fmt, fmt)          # the full-range twins share their base formats' layout
    if fmt in HBD_FMT:                                    # above 8 bits: two bytes per sample (include/ffhip.h FFHIP_PIX_FMT_*LE)
        semi, hs, vs = HBD_FMT[fmt]
        cw, ch = -((-w) >> hs), -((-h) >> vs)
        return [(h, 2 * w), (ch, 4 * cw)] if semi else [(h, 2 * w), (ch, 2 * cw), (ch, 2 * cw)]
    if fmt in YUVA_FMT:                                   # the base format's planes plus a full-size alpha plane
        return plane_shapes(YUVA_FMT[fmt], w, h) + [(h, w)]
    hs, vs = (0, 0) if fmt == PIX_FMT["yuv444p"] else (1, 0) if fmt == PIX_FMT["yuv422p"] else (1, 1)
    cw, ch = -((-w) >> hs), -((-h) >> vs)
    if fmt in (PIX_FMT["yuv420p"], PIX_FMT["yuv422p"], PIX_FMT["yuv444p"]):
        return [(h, w), (ch, cw), (ch, cw)]
    if fmt in (PIX_FMT["nv12"], PIX_FMT["nv21"]):
        return [(h, w), (ch, 2 * cw)]
    return [(h, (3 if fmt in (PIX_FMT["rgb24"], PIX_FMT["bgr24"]) else 4) * w)]


def frame_bytes(fmt, w, h):
    return sum(r * c for r, c in plane_shapes(fmt, w, h))


def alloc_batch(fmt, w, h, n, device, align=256, fill=None):
    """One uint8 tensor [n, rows, pitch] per plane, pitch rounded up to `align` bytes."""
    import torch
    out = []
    for rows, wb in plane_shapes(fmt, w, h):
        pitch = (wb + align - 1) // align * align
        t = torch.empty((n, rows, pitch), dtype=torch.uint8, device=device)
        if fill is not None:
            t.fill_(fill)
        out.append(t)
    return out


class HostTables:
    """ffhip_sws_tables_*: filter banks / coefficients without touching a device (host logic)."""

    def __init__(self, srcW, srcH, srcFormat, dstW, dstH, dstFormat, flags, ranges=None):
        L = _lib.lib()
        self._h = L.ffhip_sws_tables_create(srcW, srcH, srcFormat, dstW, dstH, dstFormat, flags)
        if not self._h:
            raise ValueError(L.ffhip_last_error().decode())
        if ranges is not None:   # sws_setColorspaceDetails()'s srcRange / dstRange
            _lib.check(L.ffhip_sws_tables_set_ranges(self._h, int(ranges[0]), int(ranges[1])), "ffhip_sws_tables_set_ranges")
        self.t = _lib.SwsTables()
        _lib.check(L.ffhip_sws_tables_get(self._h, C.byref(self.t)))
        self.unscaled_yuv2rgb = bool(L.ffhip_sws_tables_is_unscaled_yuv2rgb(self._h))

    def bank(self, name):
        f = getattr(self.t, name)
        return (np.ctypeslib.as_array(f.filter, (f.n * f.size,)).copy(), np.ctypeslib.as_array(f.pos, (f.n,)).copy(),
                f.size, f.n)

    def banks(self):
        return {k: self.bank(k) for k in ("hLum", "hChr", "vLum", "vChr")}

    def coeffs(self):
        t = self.t
        return dict(cy=t.yuv2rgb_cy, oy=t.yuv2rgb_oy, crv=t.yuv2rgb_crv, cbu=t.yuv2rgb_cbu, cgu=t.yuv2rgb_cgu,
                    cgv=t.yuv2rgb_cgv, yoffs=t.yuv2rgb_yoffs)

    def full(self):
        """the six coefficients of the full-chroma RGB writers when SWS_FULL_CHR_H_INT is in effect for this conversion, else None"""
        return [int(v) for v in self.t.yuv2rgb_full] if self.t.full_chr_h_int else None

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.lib().ffhip_sws_tables_free(self._h)
        self._h = None


class SwsContext:
    def __init__(self, srcW, srcH, srcFormat, dstW, dstH, dstFormat, flags=SWS_BICUBIC, tables=None):
        L = _lib.lib()
        self.srcW, self.srcH, self.srcFormat = srcW, srcH, srcFormat
        self.dstW, self.dstH, self.dstFormat = dstW, dstH, dstFormat
        self.flags = flags
        if L.ffhip_device_count() <= 0:
            raise RuntimeError("ffhip: no HIP device - the hip swscale path cannot run (no CPU fallback)")
        if tables is not None:   # drop-in construction from the reference's own tables
            self._c = L.ffhip_sws_from_tables(C.byref(tables))
        else:
            self._c = L.ffhip_sws_getContext(srcW, srcH, srcFormat, dstW, dstH, dstFormat, flags)
        if not self._c:
            raise ValueError(L.ffhip_last_error().decode())

    @property
    def fast_path(self):
        """True when the banks run on the column-walking kernel (ffhip_sws_fast_path)."""
        return bool(_lib.lib().ffhip_sws_fast_path(self._c) & 1)

    @property
    def mfma_path(self):
        """True when the matrix-core horizontal pass (k_sws_mfma) is available for the banks."""
        return bool(_lib.lib().ffhip_sws_fast_path(self._c) & 2)

    @property
    def wide_path(self):
        """True when the wide-bank walker (k_sws_lwalk: 5..16 taps, down-scaling) is available for the banks."""
        return bool(_lib.lib().ffhip_sws_fast_path(self._c) & 4)

    @property
    def paths(self):
        """the bit set of ffhip_sws_fast_path: 1 column walker, 2 mfma, 4 wide walker, 8 exact 2x, 16 exact 2:1, 32 16-bit walker,
        64 exact 2x of 4:2:0 into packed RGB, 128 the same sources at their own size, 256 planar 4:4:4 into packed RGB at its own size,
        512 4:2:0 between planar and semi-planar layouts at the same size, 1024 / 2048 yuv444p -> yuv420p / yuv420p -> yuv444p at the same size
        (the luma copied, the chroma planes on the exact-2:1 / exact-2x kernel), 4096 exact 3:2 down, 8192 exact 3:2 up above 8 bits"""
        return int(_lib.lib().ffhip_sws_fast_path(self._c))

    @property
    def walk16_path(self):
        """True when the 16-bit column walker (k_sws_walk16) serves the banks."""
        return bool(_lib.lib().ffhip_sws_fast_path(self._c) & 32)

    @property
    def up2_path(self):
        """True when the static-schedule exact-2x kernel (k_sws_up2) serves the banks."""
        return bool(_lib.lib().ffhip_sws_fast_path(self._c) & 8)

    @property
    def tuned_numbering(self):
        """-1 / 0 / 1: the workgroup numbering the launch tuner kept for large launches of the table converter (ffhip_sws_tuned_numbering)"""
        return int(_lib.lib().ffhip_sws_tuned_numbering(self._c))

    @property
    def up2rgb_path(self):
        """True when the static-schedule exact-2x kernel with the packed-RGB writer (k_sws_up2_rgb) serves the banks."""
        return bool(_lib.lib().ffhip_sws_fast_path(self._c) & 64)

    @property
    def down2_path(self):
        """True when the static-schedule exact-2:1 kernel (k_sws_down2) serves the banks."""
        return bool(_lib.lib().ffhip_sws_fast_path(self._c) & 16)

    def close(self):
        if getattr(self, "_c", None) and _lib is not None:
            _lib.lib().ffhip_sws_freeContext(self._c)
        self._c = None

    __del__ = close

    # -- sws_scale(): host planes (2-D uint8 numpy arrays, arbitrary strides) -------------------
    def scale(self, src, dst, srcSliceY=0, srcSliceH=None):
        L = _lib.lib()
        if srcSliceH is None:
            srcSliceH = self.srcH
        sp = (_lib.u8p * 4)()
        ss = (C.c_int * 4)()
        dp = (_lib.u8p * 4)()
        ds = (C.c_int * 4)()
        for i, a in enumerate(src):
            sp[i] = a.ctypes.data_as(_lib.u8p)
            ss[i] = a.strides[0]
        for i, a in enumerate(dst):
            dp[i] = a.ctypes.data_as(_lib.u8p)
            ds[i] = a.strides[0]
        return _lib.check(L.ffhip_sws_scale(self._c, sp, ss, srcSliceY, srcSliceH, dp, ds), "ffhip_sws_scale")

    # -- batched device face: lists of [n, rows, pitch] uint8 cuda tensors ----------------------
    def scale_batch(self, src, dst, stream=None):
        import torch
        L = _lib.lib()
        n = src[0].shape[0]
        sp = (_lib.vp * 4)()
        ss = (C.c_int * 4)()
        sf = (C.c_size_t * 4)()
        dp = (_lib.vp * 4)()
        ds = (C.c_int * 4)()
        df = (C.c_size_t * 4)()
        for i, t in enumerate(src):
            assert t.is_cuda and t.dtype == torch.uint8 and t.stride(2) == 1
            sp[i], ss[i], sf[i] = t.data_ptr(), t.stride(1), t.stride(0)
        for i, t in enumerate(dst):
            assert t.is_cuda and t.dtype == torch.uint8 and t.stride(2) == 1 and t.shape[0] == n
            dp[i], ds[i], df[i] = t.data_ptr(), t.stride(1), t.stride(0)
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        return _lib.check(L.ffhip_sws_scale_batch_dev(self._c, n, sp, ss, sf, dp, ds, df, stream),
                          "ffhip_sws_scale_batch_dev")
